#!/bin/bash
# ISA lint of every translation unit with hand-counted LDS waits: hipcc -S (device only, in parallel) + tools/check_fragment_waits.py.
# No GPU needed; about 3 minutes on 8 cores.   bash tools/lint_kernels.sh
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/msn_lint
mkdir -p "$OUT"
FILES="gemm gemm_list gemm_pw gemm_bf16 gemm_bf16res attention_bf16 pgemm pgemm_alt1 pgemm_alt2"
for f in $FILES; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -S --cuda-device-only "$ROOT/multimodal_supernovae_amd/csrc/$f.hip" -o "$OUT/$f.s" 2> "$OUT/$f.err" ) &
done
wait
rc=0
for f in $FILES; do python3 "$ROOT/tools/check_fragment_waits.py" "$OUT/$f.s" | tail -n 8 || rc=1; done
exit $rc
