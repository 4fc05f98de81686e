#!/bin/bash
# Reports behind DESIGN.md section 4's plane-GEMM rows (GPU box, repo root; diagnostic builds from build_ablate.sh present):
#   gpurun_out/final/pgemm_accuracy.txt   error against fp64 next to the native kernel's, by reduction length
#   gpurun_out/final/pgemm_variants.txt   plane counts / output forms / tail split, TFLOP/s (fp32-equivalent)
#   gpurun_out/final/pgemm_kloop_ablation.txt   what the K loop pays for (LDS-DMA, fragment reads, barrier removed)
mkdir -p gpurun_out/final
python tools/pgemm_error.py 2>&1 | grep -v amdgpu.ids > gpurun_out/final/pgemm_accuracy.txt
{ echo "# tools/bench_pgemm.py, M = 66560 (1024 cutouts x 65 tokens), random operands"
  GELU=1 PLANES=3,2 python tools/bench_pgemm.py 2>&1 | grep "^NT\|^TN"
  echo "# tail split off (TAIL=0: ops.set_pgemm_tail_split(False))"
  TAIL=0 PLANES=3 NT_ONLY=1 python tools/bench_pgemm.py 2>&1 | grep "^NT"
  echo "# M = 8320 (128 cutouts): every tile cut into K-segments vs not"
  PLANES=3 NT_ONLY=1 python tools/bench_pgemm.py 8320 2>&1 | grep "^NT"
  TAIL=0 PLANES=3 NT_ONLY=1 python tools/bench_pgemm.py 8320 2>&1 | grep "^NT"
} > gpurun_out/final/pgemm_variants.txt
bash tools/microbench/pgemm_ablate.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/final/pgemm_kloop_ablation.txt
tail -3 gpurun_out/final/pgemm_accuracy.txt; tail -3 gpurun_out/final/pgemm_kloop_ablation.txt
