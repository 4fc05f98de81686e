#!/bin/bash
# A/B of two configurations on the training step, interleaved on one box (boxes differ by up to 7 %):
#   bash tools/ab_step.sh <A> <B> [bench flags]      A, B: "" (as shipped), a library path (MSN_HIP_LIB) or VAR=value (environment)
# prints ms / step of each run
set -u
A=$1; B=$2; shift 2
one() {
  local cfg=$1; shift
  local pre=()
  if [[ "$cfg" == *=* ]]; then pre=(env "$cfg"); elif [ -n "$cfg" ]; then pre=(env "MSN_HIP_LIB=$cfg"); fi
  "${pre[@]}" python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --no-weak --no-three-tower "$@" 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step' % d['ms_per_step'])"
}
for rep in 1 2 3; do
  echo "A ${A:-shipped}: $(one "$A" "$@")"
  echo "B ${B:-shipped}: $(one "$B" "$@")"
done
