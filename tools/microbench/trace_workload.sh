#!/bin/bash
# rocprofv3 kernel table of one bench.py workload (GPU box):  bash tools/microbench/trace_workload.sh <workload> <rows per GPU> [top n]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/trace_w
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace_w --output-format csv -- python3 $R/bench.py --workload $1 --per-gpu-batch $2 --steps 10 --warmup 3 --no-alt --no-cpu-baseline --no-weak --no-three-tower --serial-towers > $R/gpurun_out/trace_w.log 2>&1
python3 $R/tools/kernel_stats.py $R/gpurun_out/trace_w ${3:-30}
rm -rf $R/gpurun_out/trace_w
