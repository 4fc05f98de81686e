for sk in "1024,1024" "1024,384" "1024,768"; do
  for b in 128 256; do
    MSN_GEMM_STREAMK=$sk python bench.py --per-gpu-batch $b --no-cpu-baseline --no-alt --no-weak --no-three-tower --steps 30 --warmup 5 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$sk', $b, round(d['ms_per_step'],3), d['roofline'].get('achieved'))"
  done
done
