#!/bin/bash
# Diagnostic builds of libmsn_hip.so with pieces of the LDS-DMA GEMM's K loop removed (results are garbage; timing only):
#   NODMA  no global -> LDS operand traffic     NOFRAG  no LDS fragment reads     NOBAR  no workgroup barrier per K-step
# usage: bash tools/microbench/build_ablate.sh NODMA [NOFRAG ...]  ->  tools/microbench/ablate/libmsn_<names>.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
PKG=$ROOT/multimodal_supernovae_amd
python3 -m multimodal_supernovae_amd.build > /dev/null
mkdir -p "$ROOT/tools/microbench/ablate"
DEFS=""; NAME=""
#   BF_NODMA / BF_NOFRAG: the same for the bf16-resident NT kernel (gemm_bf16res.hip)
#   PG_NODMA / PG_NOFRAG / PG_NOBAR: the same for the plane NT kernel (pgemm.hip); PG_SAMEK: the ring fetches K-step 0 of the tile every
#   step (cache hits: memory latency out, DMA instructions and LDS writes in); PG_NODMA_A / PG_NOFRAG_A: the A operand alone (the
#   upper bound of an A operand that does not pass through LDS)
SRC=gemm
#   PG_ANT: the A operand's LDS-DMA pieces with the nt cache policy (aux = 2);  PG_NOGELU / PG_NOAUXST / PG_NOSPLIT / PG_NOPSTORE: the
#   plane-output epilogue without its GELU arithmetic / the saved GELU' matrix / the split arithmetic / the plane stores
#   PG_STNT: the NT epilogue's final stores with the nt cache policy
#   PATTN: the plane attention kernels with their diagnostic switches compiled in (msn_set_attention_planes bits 4 - 5)
for a in "$@"; do case $a in PATTN) DEFS="$DEFS -DMSN_ABL_PATTN"; SRC=attention_planes;; PATTN_NO_NARROW) DEFS="$DEFS -DMSN_PATTN_NO_NARROW"; SRC=attention_planes;; PG_ANT) DEFS="$DEFS -DMSN_PG_A_AUX=2"; SRC=pgemm;; PG_STNT) DEFS="$DEFS -DMSN_PG_ST_AUX=2"; SRC=pgemm;; ACC_AGPR) DEFS="$DEFS -DMSN_ACC_AGPR";;  BF_*) DEFS="$DEFS -DMSN_ABL_$a"; SRC=gemm_bf16res;; PG_*) DEFS="$DEFS -DMSN_ABL_$a"; SRC=pgemm;; *) DEFS="$DEFS -DMSN_ABL_$a";; esac; NAME="${NAME}_$a"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -I"$ROOT/include" $DEFS -c "$PKG/csrc/$SRC.hip" -o "/tmp/gemm_abl$NAME.o"
OBJS=$(ls "$PKG"/build/*.o | grep -v "build/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/microbench/ablate/libmsn$NAME.so" "/tmp/gemm_abl$NAME.o" $OBJS
echo "built tools/microbench/ablate/libmsn$NAME.so"
