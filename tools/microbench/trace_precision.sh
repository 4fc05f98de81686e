#!/bin/bash
# rocprofv3 kernel table of the headline workload under a GEMM precision:  bash tools/microbench/trace_precision.sh <precision> [rows] [top n]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/trace_p
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace_p --output-format csv -- python3 $R/bench.py --gemm-precision $1 --per-gpu-batch ${2:-1024} --steps 10 --warmup 3 --no-alt --no-cpu-baseline --no-weak --no-three-tower --serial-towers > $R/gpurun_out/trace_p.log 2>&1
python3 $R/tools/kernel_stats.py $R/gpurun_out/trace_p ${3:-40} --csv $R/gpurun_out/trace_$1_b${2:-1024}.csv
rm -rf $R/gpurun_out/trace_p
