for cfg in "maven_lc_sp 1024" "maven_lc_sp 1024 --graphed" "convmixer_lc_sp 1024" "resnet18_cnn1d 1024" "resnet18_cnn1d 1024 --graphed" "vit_s8_lc_cnn1d_sp 1024" "vit_b16_bf16_lc 512" "vit_s8_lc 512" "vit_s8_lc 256" "vit_s8_lc 128" "vit_s8_lc 128 --graphed"; do
  set -- $cfg; w=$1; b=$2; shift; shift
  timeout -k 10 300 python bench.py --workload $w --per-gpu-batch $b --no-cpu-baseline --no-alt --no-weak --no-three-tower "$@" > gpurun_out/all_$w.log 2>&1 || { echo "FAILED $cfg"; tail -3 gpurun_out/all_$w.log; continue; }
  python - "$cfg" <<PY
import json,sys
d=json.loads([x for x in open("gpurun_out/all_$w.log") if x.startswith("{")][-1])
r=d["roofline"]
print(f"{sys.argv[1]:40s} ms/step {d['ms_per_step']:8.2f}  pairs/s {d['value']:9.0f}  gemm {r.get('achieved')}")
PY
done
