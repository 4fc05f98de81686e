cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/trace_tt --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload vit_s8_lc_cnn1d_sp --per-gpu-batch 1024 --steps 5 --warmup 2 --no-alt --no-cpu-baseline --no-weak --no-three-tower --serial-towers > $GRAFT_REPO_ROOT/gpurun_out/trace_tt.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kernel_stats.py $GRAFT_REPO_ROOT/gpurun_out/trace_tt 40 | head -45
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_tt
