#!/bin/bash
# PMC passes of the default bench on the GPU box (one counter set per run, kernel-trace only), then the per-kernel
# table.  Usage (from the repo root, on the GPU box):  bash tools/run_pmc.sh [extra bench.py flags]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for set in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    name=${set%% *}
    rm -rf "$ROOT/gpurun_out/pmc_$name"
    rocprofv3 --kernel-trace --pmc $set -d "$ROOT/gpurun_out/pmc_$name" --output-format csv -- \
        python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-alt --no-cpu-baseline --no-weak --no-three-tower --serial-towers "$@" > "$ROOT/gpurun_out/pmc_$name.log" 2>&1
    tail -1 "$ROOT/gpurun_out/pmc_$name.log" | cut -c1-200
done
cd "$ROOT" && python3 tools/summarize_pmc.py gpurun_out gpurun_out/pmc_summary.txt "$@"
