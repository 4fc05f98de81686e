#!/bin/bash
# Plane NT GEMM: what the K loop pays for -- tools/bench_pgemm.py on the diagnostic builds of build_ablate.sh (timing only).
for v in "" _PG_NODMA _PG_NOFRAG _PG_NODMA_PG_NOFRAG _PG_NOBAR; do
  if [ -z "$v" ]; then lib=multimodal_supernovae_amd/lib/libmsn_hip.so; else lib=tools/microbench/ablate/libmsn$v.so; fi
  echo "== ${v:-as shipped}"
  MSN_HIP_LIB=$PWD/$lib PLANES=${PLANES:-3} NT_ONLY=1 timeout -k 10 200 python tools/bench_pgemm.py | grep "^NT"
done
