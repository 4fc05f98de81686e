#!/bin/bash
# Where a workgroup of the one-pass attention backward (mattn_bwd_fused_kernel) spends its time: attention_mfma.hip compiled
# with -DMSN_ATTN_TIMELINE (shader-clock stamps of wave 0 of every workgroup), linked with the other objects of the default
# build into tools/microbench/ablate/libmsn_attn_timeline.so, then tools/microbench/attn_fused_timeline.py on the GPU.
# usage (GPU box): bash tools/microbench/attn_fused_timeline.sh [B T heads hd]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
PKG=$ROOT/multimodal_supernovae_amd
mkdir -p "$ROOT/tools/microbench/ablate"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DMSN_ATTN_TIMELINE -c "$PKG/csrc/attention_mfma.hip" -o /tmp/attention_mfma_tl.o
OBJS=$(ls "$PKG"/build/*.o | grep -v "build/attention_mfma.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/microbench/ablate/libmsn_attn_timeline.so" /tmp/attention_mfma_tl.o $OBJS
MSN_HIP_LIB=$ROOT/tools/microbench/ablate/libmsn_attn_timeline.so python3 "$ROOT/tools/microbench/attn_fused_timeline.py" "$@"
