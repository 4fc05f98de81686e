#!/usr/bin/env python3
"""Which Python lines of a headline training step still launch ATen / copy kernels?  (torch.profiler with stacks.)

    python tools/aten_sites.py [--batch 1024]
"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
args = ap.parse_args()
dev = torch.device("cuda:0")
model = bench.build_model(dev)
batch = bench.synthetic_batch(args.batch, 0, dev)
opt = model.configure_optimizers()["optimizer"]


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step()
    torch.cuda.synchronize()
names = collections.Counter(ev.name for ev in prof.events() if ev.name.startswith("aten::"))
for name, n in names.most_common(40):
    print(f"{n:4d}  {name}")
print("--- with Python frames")
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::cat", "aten::_to_copy", "aten::clone", "aten::sum", "aten::add", "aten::add_", "aten::mul"):
        frames = [f for f in (ev.stack or []) if "multimodal_supernovae_amd" in f or "bench.py" in f]
        sites[(ev.name, " <- ".join(frames[:2]) if frames else "<no python frame>")] += 1
for (name, site), n in sites.most_common(60):
    print(f"{n:4d}  {name:14s} {site}")
