#!/bin/bash
# A/B of two bench.py flag sets on one box, interleaved:  bash tools/ab_flags.sh "<flags A>" "<flags B>" [common flags]
A=$1; B=$2; shift 2
one() { python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --no-weak --no-three-tower $1 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step' % d['ms_per_step'])"; }
for rep in 1 2 3 4; do
  echo "A [$A]: $(one "$A" "$@")"
  echo "B [$B]: $(one "$B" "$@")"
done
