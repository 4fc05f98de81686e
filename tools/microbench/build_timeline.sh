#!/bin/bash
# Diagnostic build of libmsn_hip.so for tools/microbench/gemm_timeline.py: gemm.hip compiled with -DMSN_TIMELINE=<level>
# (1 = per-workgroup start / first-data / loop-end / stores-acknowledged timestamps + core-clock probe,
#  2 = additionally the cycles wave 0 waits for fragments, LDS-DMA pieces and the barrier inside the K loop).
# usage: bash tools/microbench/build_timeline.sh [level]   ->  tools/microbench/ablate/libmsn_timeline.so
set -e
LEVEL=${1:-1}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
PKG=$ROOT/multimodal_supernovae_amd
python3 -m multimodal_supernovae_amd.build > /dev/null
mkdir -p "$ROOT/tools/microbench/ablate"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DMSN_TIMELINE=$LEVEL"
# every translation unit that sees GemmArgs (it grows a member under MSN_TIMELINE)
for f in gemm gemm_bf16 gemm_reg gemm_pw; do /opt/rocm/bin/hipcc $FLAGS -c "$PKG/csrc/$f.hip" -o "/tmp/${f}_tl.o" & done; wait
OBJS=$(ls "$PKG"/build/*.o | grep -v "build/gemm.o\|build/gemm_bf16.o\|build/gemm_reg.o\|build/gemm_pw.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/microbench/ablate/libmsn_timeline.so" /tmp/gemm_tl.o /tmp/gemm_bf16_tl.o /tmp/gemm_reg_tl.o /tmp/gemm_pw_tl.o $OBJS
echo "built $ROOT/tools/microbench/ablate/libmsn_timeline.so (MSN_TIMELINE=$LEVEL)"
