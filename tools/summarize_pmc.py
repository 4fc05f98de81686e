#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES ...) of bench.py into a
per-kernel table + profiles/pmc_sgemm.json (HBM-side traffic per GEMM launch, gfx950 corrections applied:
FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> doubled; WRITE_SIZE exact; both in KB).

    python tools/summarize_pmc.py gpurun_out profiles/r01_pmc_summary.txt
"""
import collections
import csv
import glob
import json
import os
import sys


def load(root, name):
    files = glob.glob(os.path.join(root, f"pmc_{name}", "*", "*counter_collection.csv"))
    d = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0]))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            k = k.replace("void ", "")
            keys = [k]
            if "sgemm_kernel" in k or "sgemm_dma_kernel" in k or "sgemm_list_kernel" in k:      # per template instance (layout = the two bools) and the aggregate row
                keys.append("msn::sgemm_*_kernel<*>")
            if "bgemm_nt_kernel" in k or "bgemm_tn_kernel" in k:    # bf16-resident GEMMs (cfg5)
                keys.append("msn::bgemm_{nt,tn}_kernel<*>")
            if "pgemm_nt_kernel" in k or "pgemm_tn_kernel" in k:    # plane GEMMs (fp32-grade from resident bf16 planes)
                keys.append("msn::pgemm_{nt,tn}_kernel<*>")
            for k in keys:
                e = d[k][r["Counter_Name"]]
                e[0] += 1
                e[1] += float(r["Counter_Value"])
                e[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return d


def main():
    root, out = sys.argv[1], sys.argv[2]
    fs, ws, mf = load(root, "FETCH_SIZE"), load(root, "WRITE_SIZE"), load(root, "SQ_VALU_MFMA_BUSY_CYCLES")
    lines = ["kernel | launches | fetch GB (2 x FETCH_SIZE) | write GB | MB / launch | MFMA busy / (SIMD-cycles) | eff. clock GHz"]
    names = sorted(fs, key=lambda k: -fs[k]["FETCH_SIZE"][1])
    summary = {}
    for k in names[:32]:
        n, f, _ = fs[k]["FETCH_SIZE"]
        w = ws[k]["WRITE_SIZE"][1] if k in ws else 0.0
        fetch_b, write_b = 2 * f * 1024, w * 1024
        util = clock = float("nan")
        if k in mf and mf[k]["GRBM_GUI_ACTIVE"][2]:
            t = mf[k]["GRBM_GUI_ACTIVE"][2] * 1e-9
            clock = mf[k]["GRBM_GUI_ACTIVE"][1] / 8 / t
            util = mf[k]["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (t * clock * 1024)
            clock /= 1e9
        lines.append(f"{k[:60]:60s} | {n:5d} | {fetch_b / 1e9:9.2f} | {write_b / 1e9:8.2f} | {(fetch_b + write_b) / n / 1e6:9.1f} | "
                     f"{util:6.3f} | {clock:5.2f}")
        summary[k] = {"launches": n, "traffic_bytes_per_launch": (fetch_b + write_b) / n, "mfma_util": util,
                      "clock_ghz": clock}
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))
    import datetime
    import subprocess
    # the commit the measured binary was built from: `git -C <repo>` where the checkout has its history, else the stamp
    # tools/gpu.sh writes into the snapshot before it travels (a gpurun box has no .git)
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    commit = ""
    try:
        commit = subprocess.run(["git", "-C", repo, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        pass
    if not commit and os.path.exists(os.path.join(repo, ".msn_commit")):
        commit = open(os.path.join(repo, ".msn_commit")).read().strip()
    commit = commit or "unknown"
    extra = " ".join(sys.argv[3:])
    args = sys.argv[3:]
    rows = int(args[args.index("--per-gpu-batch") + 1]) if "--per-gpu-batch" in args else 1024
    workload = args[args.index("--workload") + 1] if "--workload" in args else "vit_s8_lc"
    g = summary.get("msn::sgemm_*_kernel<*>")
    if g:
        # one file per (workload, rows per GPU); the headline at 1024 rows keeps the historical name
        name = "pmc_sgemm.json" if (workload == "vit_s8_lc" and rows == 1024) else f"pmc_sgemm_{workload}_b{rows}.json"
        json.dump({"kernel": "msn::sgemm_dma_kernel + msn::sgemm_list_kernel + msn::sgemm_kernel (all fp32 GEMM launches)",
                   "traffic_bytes_per_launch": g["traffic_bytes_per_launch"],
                   "mfma_busy_fraction": g["mfma_util"], "effective_clock_ghz": g["clock_ghz"], "launches": g["launches"],
                   "workload": f"bench.py --workload {workload} --per-gpu-batch {rows}",
                   "collected": datetime.date.today().isoformat(), "commit": commit,
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES passes, tools/summarize_pmc.py"},
                  open(os.path.join(os.path.dirname(out), name), "w"), indent=1)
    pg = summary.get("msn::pgemm_{nt,tn}_kernel<*>")
    if pg:
        name = "pmc_pgemm.json" if (workload == "vit_s8_lc" and rows == 1024) else f"pmc_pgemm_{workload}_b{rows}.json"
        json.dump({"kernel": "msn::pgemm_nt_kernel + msn::pgemm_tn_kernel (plane GEMM launches)",
                   "traffic_bytes_per_launch": pg["traffic_bytes_per_launch"],
                   "mfma_busy_fraction": pg["mfma_util"], "effective_clock_ghz": pg["clock_ghz"], "launches": pg["launches"],
                   "workload": f"bench.py --workload {workload} --per-gpu-batch {rows} " + extra,
                   "collected": datetime.date.today().isoformat(), "commit": commit,
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES passes, tools/summarize_pmc.py"},
                  open(os.path.join(os.path.dirname(out), name), "w"), indent=1)
    bg = summary.get("msn::bgemm_{nt,tn}_kernel<*>")
    if bg:
        json.dump({"kernel": "msn::bgemm_nt_kernel + msn::bgemm_tn_kernel", "traffic_bytes_per_launch": bg["traffic_bytes_per_launch"],
                   "mfma_busy_fraction": bg["mfma_util"], "effective_clock_ghz": bg["clock_ghz"], "launches": bg["launches"],
                   "workload": "bench.py " + extra, "collected": datetime.date.today().isoformat(), "commit": commit,
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES passes, tools/summarize_pmc.py"},
                  open(os.path.join(os.path.dirname(out), "pmc_bgemm.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
