#!/bin/bash
# The measurement pass behind profiles/ (GPU box, repo root):  bash tools/final_measurements.sh pmc | stats
#   pmc   : tools/run_pmc.sh for the headline, 128 / 256 rows per GPU, cfg5 and Maven  -> gpurun_out/final/pmc_summary_*.txt, pmc_*.json
#   stats : rocprofv3 --kernel-trace per-kernel tables of the same runs, tools/roofline_all.py, the default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/gpurun_out/final"
if [ "$1" = pmc ]; then
  for cfg in "headline:" "b128:--per-gpu-batch 128" "b256:--per-gpu-batch 256" "cfg5:--workload vit_b16_bf16_lc --per-gpu-batch 512" "maven:--workload maven_lc_sp"; do
    tag=${cfg%%:*}; flags=${cfg#*:}
    echo "== pmc $tag"
    bash "$ROOT/tools/run_pmc.sh" $flags || exit 1
    cp "$ROOT/gpurun_out/pmc_summary.txt" "$ROOT/gpurun_out/final/pmc_summary_$tag.txt"
  done
  cp "$ROOT"/gpurun_out/pmc_*.json "$ROOT/gpurun_out/final/"
else
  cd /tmp && export TMPDIR=/tmp
  for cfg in "vit-s8_b1024:" "vit-s8_b128:--per-gpu-batch 128" "vit-s8_b256:--per-gpu-batch 256" "vit_b16_bf16_b512:--workload vit_b16_bf16_lc --per-gpu-batch 512" "maven_lc_sp_b1024:--workload maven_lc_sp" "convmixer_lc_sp_b1024:--workload convmixer_lc_sp" "vit_s8_lc_cnn1d_sp_b256:--workload vit_s8_lc_cnn1d_sp --per-gpu-batch 256" "resnet18_cnn1d_b256:--workload resnet18_cnn1d --per-gpu-batch 256"; do
    tag=${cfg%%:*}; flags=${cfg#*:}
    echo "== kernel trace $tag"
    rm -rf "$ROOT/gpurun_out/trace_$tag"
    rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/trace_$tag" --output-format csv -- \
        python3 "$ROOT/bench.py" --steps 10 --no-alt --no-cpu-baseline --no-weak --no-three-tower --serial-towers $flags > "$ROOT/gpurun_out/trace_$tag.log" 2>&1 || { tail -3 "$ROOT/gpurun_out/trace_$tag.log"; exit 1; }
    python3 "$ROOT/tools/kernel_stats.py" "$ROOT/gpurun_out/trace_$tag" 12 --csv "$ROOT/gpurun_out/final/bench_${tag}_kernel_stats.csv" | head -14
    if [ "$tag" = vit-s8_b1024 ]; then cp "$(ls "$ROOT"/gpurun_out/trace_$tag/*/*kernel_stats.csv | head -1)" "$ROOT/gpurun_out/final/rocprofv3_stats_headline.csv"; fi
    rm -rf "$ROOT/gpurun_out/trace_$tag"
  done
  cd "$ROOT"
  echo "== roofline_all"; python3 tools/roofline_all.py gpurun_out/final/roofline_all.json | tail -12
  echo "== default bench"; python3 bench.py > gpurun_out/final/bench_default_run.json 2> gpurun_out/final/bench_default_run.err; tail -c 600 gpurun_out/final/bench_default_run.json
fi
