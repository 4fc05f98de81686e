for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 300 python tools/rccl_smoke.py > gpurun_out/rccl_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(grep -c 'RCCL SMOKE OK' gpurun_out/rccl_$i.log)"
  if [ $rc -ne 0 ]; then grep -v "^\[W\|amdgpu" gpurun_out/rccl_$i.log | tail -15; break; fi
done
