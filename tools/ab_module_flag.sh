#!/bin/bash
# A/B of a boolean module attribute of multimodal_supernovae_amd.functional on the training step, one box, interleaved:
#   bash tools/ab_module_flag.sh FUSED_FF --workload maven_lc_sp        (True = as shipped, False = the other path)
FLAG=$1; shift
one() { python -c "
import sys, runpy
import multimodal_supernovae_amd.functional as F
assert hasattr(F, '$FLAG')
setattr(F, '$FLAG', bool($1))
sys.argv = ['bench.py', '--steps', '20', '--warmup', '5', '--no-alt', '--no-cpu-baseline', '--no-weak', '--no-three-tower'] + sys.argv[1:]
runpy.run_path('bench.py', run_name='__main__')" "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step' % d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "$FLAG = True:  $(one 1 "$@")"
  echo "$FLAG = False: $(one 0 "$@")"
done
