#!/bin/bash
# SQ counter sets (one rocprofv3 --pmc pass each, kernel-trace only) for a command, then per-kernel sums.
# usage (GPU box, repo root): bash tools/pmc_kernels.sh <tag> <python script and args>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    rm -rf "$ROOT/gpurun_out/pmck_${tag}_$i"
    rocprofv3 --kernel-trace --pmc $set -d "$ROOT/gpurun_out/pmck_${tag}_$i" --output-format csv -- python3 "$ROOT/$1" "${@:2}" > "$ROOT/gpurun_out/pmck_${tag}_$i.log" 2>&1 || { tail -3 "$ROOT/gpurun_out/pmck_${tag}_$i.log"; exit 1; }
    i=$((i + 1))
done
cd "$ROOT" && python3 - "$tag" <<'PY'
import collections, csv, glob, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(f"gpurun_out/pmck_{tag}_*/*/*counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen and r["Counter_Name"] == "SQ_WAVE_CYCLES":
            seen.add(key); calls[k] += 1
names = sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CYCLES", 0))[:14]
cols = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_VALU_MFMA_BUSY_CYCLES",
        "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]
out = open(f"gpurun_out/pmck_{tag}.txt", "w")
for k in names:
    a = agg[k]
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    line = (f"{k[:58]:58s} calls {calls[k]:4d} | per wave-cycle: wait_any {a.get('SQ_WAIT_INST_ANY', 0) / wc:5.2f} wait_lds {a.get('SQ_WAIT_INST_LDS', 0) / wc:5.2f} "
            f"valu {a.get('SQ_ACTIVE_INST_VALU', 0) / wc:5.2f} lds {a.get('SQ_ACTIVE_INST_LDS', 0) / wc:5.2f} vmem {a.get('SQ_ACTIVE_INST_VMEM', 0) / wc:5.2f} | "
            f"insts valu {a.get('SQ_INSTS_VALU', 0):.3g} mfma {a.get('SQ_INSTS_MFMA', 0):.3g} lds {a.get('SQ_INSTS_LDS', 0):.3g} | "
            f"bank conflict / lds active {a.get('SQ_LDS_BANK_CONFLICT', 0) / (a.get('SQ_LDS_IDX_ACTIVE', 0) or 1):5.2f} | mfma busy/busy {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (a.get('SQ_BUSY_CYCLES', 0) or 1):5.2f}")
    print(line); out.write(line + "\n")
PY
