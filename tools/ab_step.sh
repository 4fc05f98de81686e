#!/bin/bash
# A/B of two builds on the training step, interleaved on one box:  bash tools/ab_step.sh <libA or ""> <libB or ""> [bench flags]
# prints ms / step of each run ("" = the shipped library)
set -u
A=$1; B=$2; shift 2
one() { if [ -n "$1" ]; then MSN_HIP_LIB=$1 python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --no-weak --no-three-tower "${@:2}" 2>/dev/null; else python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --no-weak --no-three-tower "${@:2}" 2>/dev/null; fi | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step' % d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "A ${A:-shipped}: $(one "$A" "$@")"
  echo "B ${B:-shipped}: $(one "$B" "$@")"
done
