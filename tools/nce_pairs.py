#!/usr/bin/env python3
"""Per-configuration (main, finish) kernel durations of the fused InfoNCE from a rocprofv3 kernel-trace CSV of
tools/bench_infonce.py:  python tools/nce_pairs.py <dir with *_kernel_trace.csv>"""
import collections
import csv
import glob
import statistics
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(path)):
    if "nce_" in r["Kernel_Name"]:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"],
                     int(r["Grid_Size_X"]) // 256, int(r["Grid_Size_Y"])))
rows.sort()
agg = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    if "finish" in b[2] and "finish" not in a[2]:
        agg[("fwd" if "fwd" in a[2] else "bwd", a[3], a[4])].append((a[1], b[1]))
for k, v in sorted(agg.items()):
    print(f"{k[0]} qtiles={k[1]:4d} ksplit={k[2]:3d} n={len(v):4d}  main {statistics.median(x[0] for x in v) / 1e3:6.1f} us  "
          f"finish {statistics.median(x[1] for x in v) / 1e3:6.1f} us")
