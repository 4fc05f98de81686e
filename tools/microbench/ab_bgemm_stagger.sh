run() { python bench.py --workload vit_b16_bf16_lc --per-gpu-batch 512 --no-cpu-baseline --no-alt --no-weak --no-three-tower --steps 20 --warmup 5 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), round(d['roofline']['achieved'],1))"; }
run stagger
MSN_HIP_LIB=$PWD/tools/microbench/ablate/libmsn_nostagger.so run nostagger
run stagger
MSN_HIP_LIB=$PWD/tools/microbench/ablate/libmsn_nostagger.so run nostagger
