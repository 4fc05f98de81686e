#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the plane NT kernel under given tile walks (separate rocprofv3 --pmc passes, kernel-trace only).
# usage (GPU box, repo root): bash tools/pmc_walk.sh <tag> "<shape> <col_group> <super_rows>" ...     (MSN_HIP_LIB selects the build)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
    set -- $cfg
    for ctr in FETCH_SIZE WRITE_SIZE; do
        d="$ROOT/gpurun_out/pmcw_${tag}_$1_$2_$3_$ctr"
        rm -rf "$d"
        rocprofv3 --kernel-trace --pmc $ctr -d "$d" --output-format csv -- python3 "$ROOT/tools/bench_pgemm_walk.py" --one $1 $2 $3 20 > "$d.log" 2>&1 || { tail -3 "$d.log"; exit 1; }
    done
done
cd "$ROOT" && python3 - "$tag" <<'PY'
import collections, csv, glob, re, sys
tag = sys.argv[1]
rows = collections.defaultdict(dict)
for f in sorted(glob.glob(f"gpurun_out/pmcw_{tag}_*/*/*counter_collection.csv")):
    m = re.search(rf"pmcw_{tag}_(\w+?)_(\d+)_(\d+)_(FETCH_SIZE|WRITE_SIZE)", f)
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if "pgemm_nt_kernel" in r["Kernel_Name"] and r["Counter_Name"] == m.group(4):
            tot += float(r["Counter_Value"]); n += 1
    if n:
        rows[(m.group(1), m.group(2), m.group(3))][m.group(4)] = tot / n * 1024 * (2 if m.group(4) == "FETCH_SIZE" else 1)
out = open(f"gpurun_out/pmcw_{tag}.txt", "w")
for k, v in rows.items():
    line = f"{tag} {k[0]:6s} cg {k[1]} sr {k[2]}: fetch (2 x FETCH_SIZE) {v.get('FETCH_SIZE', 0) / 1e6:7.1f} MB  write {v.get('WRITE_SIZE', 0) / 1e6:7.1f} MB per launch"
    print(line); out.write(line + "\n")
PY
