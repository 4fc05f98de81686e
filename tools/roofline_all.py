#!/usr/bin/env python3
"""profiles/roofline_all.json: one object per BASELINE.json configuration + the reference-native Maven workload --
{workload, rows per GPU, ms per step, dominant kernel, bound, achieved, peak, frac, algorithmic bytes per launch,
traffic (PMC, when a matching profiles/pmc_*.json exists)} -- produced by running bench.py once per workload on the GPU box
(each run measures its dominant GEMM family with HIP events on the launch stream, as the default line does) and, for the
HBM-bound kernels of the reference towers, by timing them against the bytes they must move at 8 TB/s.

    python tools/roofline_all.py [out.json]        (on the GPU box; about two minutes)
"""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RUNS = [("vit_s8_lc", 1024, "BASELINE cfg3 towers = the headline"), ("vit_s8_lc", 256, "cfg3 at its stated 256 rows per GPU"),
        ("vit_s8_lc", 128, "global batch 1024 on 8 GPUs"), ("resnet18_cnn1d", 256, "BASELINE cfg2"),
        ("vit_s8_lc_cnn1d_sp", 256, "BASELINE cfg4 (three towers)"), ("vit_b16_bf16_lc", 512, "BASELINE cfg5"),
        ("maven_lc_sp", 1024, "reference-native Maven pretraining towers"), ("convmixer_lc_sp", 1024, "reference-native three towers")]


def bench(workload, rows):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--per-gpu-batch", str(rows), "--steps", "10",
           "--no-cpu-baseline", "--no-alt", "--no-weak", "--no-three-tower"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"bench.py {workload} failed:\n{r.stderr[-2000:]}")
    return json.loads(lines[-1])


def hbm_kernels():
    """The HBM-bound row kernels of the reference towers on the Maven light-curve token matrix (204 800 x 64): GB/s of the bytes
    they must move against 8 TB/s."""
    from multimodal_supernovae_amd import ops
    out = []
    M, e = 204800, 64
    x = torch.randn(M, e, device="cuda")
    g, b = torch.ones(e, device="cuda"), torch.zeros(e, device="cuda")

    def timeit(fn, iters=50):
        for _ in range(5):
            fn()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        t.record()
        torch.cuda.synchronize()
        return s.elapsed_time(t) / iters * 1e-3

    y, mean, rstd = ops.layernorm_fwd(x, g, b)
    t = timeit(lambda: ops.layernorm_fwd(x, g, b))
    out.append(("msn::ln_fwd_kernel (LayerNorm forward, 8 B / element + statistics)", 8.0 * M * e, t))
    dy = torch.randn_like(x)
    t = timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g))
    out.append(("msn::ln_bwd_kernel (LayerNorm backward, 12 B / element)", 12.0 * M * e, t))
    return [{"kernel": k, "bound": "hbm", "algorithmic_bytes_per_launch": nb, "us_per_launch": t * 1e6,
             "achieved": nb / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": nb / t / 8e12} for k, nb, t in out]


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "roofline_all.json")
    rows_out = []
    for workload, rows, note in RUNS:
        d = bench(workload, rows)
        r = d["roofline"]
        rows_out.append({"workload": workload, "note": note, "rows_per_gpu": rows, "ms_per_step": d["ms_per_step"],
                         "pairs_per_s": d["value"], "dominant_kernel": r["kernel"], "bound": r["bound"], "achieved": r["achieved"],
                         "peak": r["peak"], "unit": r["unit"], "frac": r["frac"], "launches_per_step": r["launches_per_step"],
                         "ms_per_step_in_kernel": r["ms_per_step_in_kernel"],
                         "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"], "traffic": r["traffic"],
                         "traffic_source": r["traffic_source"],
                         "attention": r.get("attention")})      # msn_attention_fwd / _bwd launches of the same step, by shape
        print(f"{workload:22s} B={rows:5d} {d['ms_per_step']:8.2f} ms  {r['achieved']:7.1f} / {r['peak']} {r['unit']} = {r['frac']:.3f}", flush=True)
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    if not commit and os.path.exists(os.path.join(ROOT, ".msn_commit")):      # a gpurun box has no .git: tools/gpu.sh stamps the snapshot
        commit = open(os.path.join(ROOT, ".msn_commit")).read().strip()
    commit = commit or "unknown"
    json.dump({"commit": commit, "device": torch.cuda.get_device_name(0), "workloads": rows_out, "hbm_bound_kernels": hbm_kernels()},
              open(out_path, "w"), indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
