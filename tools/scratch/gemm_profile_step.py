"""Per-shape msn_sgemm time inside one real training step of a workload (event pairs around every call: ops.GEMM_PROFILE),
twice -- the spread between the two passes is the noise of the method.    python tools/scratch/gemm_profile_ab.py maven_lc_sp"""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from multimodal_supernovae_amd import ops, _lib
name = sys.argv[1] if len(sys.argv) > 1 else "maven_lc_sp"
dev = torch.device("cuda", 0)
model, batch = bench.build_workload(name, 1024, 0, dev)
opt = model.configure_optimizers() if hasattr(model, "configure_optimizers") else None
def step():
    loss = model.training_step(batch, 0) if hasattr(model, "training_step") else None
    loss.backward()
    for p in model.parameters(): p.grad = None
res = {}
for on in (0, 1):
    step(); step(); torch.cuda.synchronize()
    ops.GEMM_PROFILE = []
    step(); torch.cuda.synchronize()
    prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
    agg = collections.OrderedDict()
    for rec in prof:
        ev0, ev1, fl, key = rec[0], rec[1], rec[2], rec[3]
        k = (key, rec[4])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += ev0.elapsed_time(ev1) * 1e3
    res[on] = agg
    print("== pass", on, "total us", round(sum(v[1] for v in agg.values())))
for k in res[0]:
    a, b = res[0][k], res[1].get(k, [0, 0.0])
    print(f"{str(k):60s} calls {a[0]:3d}  pass 0 {a[1] / a[0]:7.1f} us  pass 1 {b[1] / max(b[0], 1):7.1f} us")
