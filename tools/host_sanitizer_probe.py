#!/usr/bin/env python3
"""Entry points of the C-ABI that return before any launch (validation errors, launch plans, workspace sizes, work-list
descriptors), run against the host-sanitized build (tools/build_host_sanitized.sh: ASan + UBSan on the host side of every
translation unit) with the ASan runtime preloaded:

    bash tools/build_host_sanitized.sh
    LD_PRELOAD=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 \\
        python tools/host_sanitizer_probe.py
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MSN_HIP_LIB"] = os.path.join(ROOT, "multimodal_supernovae_amd", "build_asan", "libmsn_hip_asan.so")
from multimodal_supernovae_amd import _lib
L = _lib.lib()
print("version", L.msn_version())
fake = ctypes.c_void_p(4096)
rc = L.msn_sgemm(0, 1, 4, 4, 0, None, 4, None, 4, None, 4, None, 0, None, 0, 0, None, 0, None)
print(rc, L.msn_last_error())
print(L.msn_sgemm_workspace_bytes(1, 0, 384, 1536, 66560), L.msn_sgemm_workspace_bytes(0, 1, 66560, 1536, 384))
d = (_lib.GemmDesc * 2)()
for i, (oa, ob, M, N, K) in enumerate([(0, 0, 8320, 384, 1536), (1, 0, 1536, 384, 8320)]):
    d[i].opA, d[i].opB, d[i].M, d[i].N, d[i].K = oa, ob, M, N, K
    d[i].A = d[i].B = d[i].C = 4096
    d[i].lda = K if oa == 0 else M
    d[i].ldb = N
    d[i].ldc = N
print("list ws", L.msn_sgemm_list_workspace_bytes(2, ctypes.cast(d, ctypes.c_void_p)))
print(L.msn_conv2d_workspace_bytes(8, 64, 64, 64, 128, 3, 3, 1, 1, 1, 1), L.msn_conv2d_implicit_ok(8, 64, 64, 64, 128, 3, 3, 1, 1, 1, 1))
print(L.msn_infonce_workspace_bytes(128, 128, 1024, 1024, 128), L.msn_wgrad_bias_workspace_bytes(64, 256, 204800))
for bad in (lambda: L.msn_set_gemm_streamk(-1, 0), lambda: L.msn_set_gemm_lds_pad(1 << 30), lambda: L.msn_set_attention_path(7),
            lambda: L.msn_sgemm_list(0, None, 0, None, 0, None), lambda: L.msn_sgemm_list(4, ctypes.cast(d, ctypes.c_void_p), 0, None, 0, None)):
    assert bad() == 1, L.msn_last_error()
print("HOST SANITIZER PROBE OK")
