#!/usr/bin/env python3
"""Host-side (Python) time of one training step of a bench workload: cProfile over 10 steps, top functions.

    python tools/host_profile.py [workload] [batch] [--dp]
--dp: a one-rank RCCL group with the sharded loss and the hook-driven gradient reducer forced on (the host work of a data-parallel rank).
"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DP = "--dp" in sys.argv
if DP:
    sys.argv.remove("--dp")
    os.environ.update(MSN_DIST_FORCE_INIT="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29655")
import torch
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "resnet18_cnn1d"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
if name == "vit_s8_lc":
    model, batch = bench.build_model(dev), bench.synthetic_batch(b, 0, dev)
else:
    model, batch = bench.build_workload(name, b, 0, dev)
opt = model.configure_optimizers()["optimizer"]
reducer = None
if DP:
    from multimodal_supernovae_amd import distributed as D
    D.init_from_env(backend="nccl")
    model.global_negatives = "always"
    reducer = D.GradientReducer(model.parameters(), force=True)


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    loss.backward()
    if reducer is not None:
        reducer.finish()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
t_issue = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
t_total = (time.perf_counter() - t0) / 10
print(f"{name} B={b}: host issue time {t_issue * 1e3:.2f} ms / step, with GPU drain {t_total * 1e3:.2f} ms / step")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
