#!/usr/bin/env python3
"""Per-kernel table (calls, total, average, share) from a rocprofv3 --kernel-trace CSV directory:
    python tools/kernel_stats.py <dir> [top_n] [--csv out.csv]"""
import collections
import csv
import glob
import re
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 40
agg = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(path)):
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")     # (kernels of an unnamed namespace: keep their own name)
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    if "clock_probe_kernel" in name:      # bench.py's one-wave clock probe idles on a side stream for a whole step: not part of the step
        continue
    a = agg[name]
    a[0] += 1
    a[1] += d
total = sum(v[1] for v in agg.values())
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")]
for k, (n, t) in rows:
    out.append((k, n, t, t / n, 100.0 * t / total))
if "--csv" in sys.argv:
    with open(sys.argv[sys.argv.index("--csv") + 1], "w", newline="") as f:
        csv.writer(f).writerows(out)
print(f"total kernel time {total / 1e6:.2f} ms over {sum(v[0] for v in agg.values())} launches")
for k, n, t, avg, pct in out[1:top + 1]:
    print(f"{pct:6.2f}%  {t / 1e6:9.3f} ms  {n:6d} x {avg / 1e3:9.1f} us  {k[:110]}")
