#!/bin/bash
# A/B of the fused feed-forward of the narrow towers on the training step (one box, interleaved):  bash tools/ab_fused_ff.sh [bench flags]
one() { python -c "
import sys, runpy
import multimodal_supernovae_amd.functional as F
F.FUSED_FF = bool($1)
sys.argv = ['bench.py', '--steps', '20', '--warmup', '5', '--no-alt', '--no-cpu-baseline', '--no-weak', '--no-three-tower'] + sys.argv[1:]
runpy.run_path('bench.py', run_name='__main__')" "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step' % d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "fused FF:   $(one 1 "$@")"
  echo "unfused FF: $(one 0 "$@")"
done
