#!/bin/bash
# ISA lint of every translation unit with hand-counted LDS waits: hipcc -S (device only, in parallel) + tools/check_fragment_waits.py,
# and the scratch check (tools/check_scratch.py on the resource-usage remarks of the same compilations): no register spills to memory
# in any plane GEMM kernel, in ANY attention kernel (vector-ALU attn_*, fp32 matrix-core mattn_*, plane pattn_*, bf16 battn_*, class-token
# cls_attn_*) and in the bf16-resident GEMM kernels.
# No GPU needed; about 4 minutes on 8 cores.   bash tools/lint_kernels.sh
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/msn_lint
mkdir -p "$OUT"
FILES="gemm gemm_list gemm_pw gemm_bf16 gemm_bf16res attention_bf16 attention_planes attention_mfma attention cls_attention pgemm pgemm_alt2"
pids=()
for f in $FILES; do
  rm -f "$OUT/$f.s" "$OUT/$f.err" "$OUT/$f.rc"          # never lint a stale listing
  [ -f "$ROOT/multimodal_supernovae_amd/csrc/$f.hip" ] || continue
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast -Rpass-analysis=kernel-resource-usage -S --cuda-device-only "$ROOT/multimodal_supernovae_amd/csrc/$f.hip" -o "$OUT/$f.s" 2> "$OUT/$f.err"; echo $? > "$OUT/$f.rc" ) &
done
wait
rc=0
for f in $FILES; do
  [ -f "$ROOT/multimodal_supernovae_amd/csrc/$f.hip" ] || continue
  if [ "$(cat "$OUT/$f.rc" 2>/dev/null)" != "0" ] || [ ! -s "$OUT/$f.s" ]; then
    echo "hipcc -S failed for $f.hip:"; tail -n 20 "$OUT/$f.err"; rc=1; continue
  fi
  python3 "$ROOT/tools/check_fragment_waits.py" "$OUT/$f.s" | tail -n 8 || rc=1
done
# no register spills to memory in any plane GEMM kernel or attention kernel (remarks of the same compilations)
python3 "$ROOT/tools/check_scratch.py" "$OUT"/pgemm.err "$OUT"/pgemm_alt2.err -- pgemm_nt_kernel pgemm_tn_kernel | tail -n 1 || rc=1
python3 "$ROOT/tools/check_scratch.py" "$OUT"/attention_planes.err "$OUT"/attention_mfma.err "$OUT"/attention.err "$OUT"/attention_bf16.err "$OUT"/cls_attention.err -- attn_ | tail -n 1 || rc=1
# ... nor in the bf16-resident GEMM kernels (254 - 256 registers each since round 6: the first spill would cost the whole K loop)
python3 "$ROOT/tools/check_scratch.py" "$OUT"/gemm_bf16res.err -- bgemm_nt_kernel bgemm_tn_kernel | tail -n 1 || rc=1
exit $rc
