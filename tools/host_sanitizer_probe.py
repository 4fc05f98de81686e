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
for bad in (lambda: L.msn_set_gemm_list(7), lambda: L.msn_set_gemm_tile_n(96), lambda: L.msn_set_attention_path(7),
            lambda: L.msn_sgemm_list(0, None, 0, None, 0, None), lambda: L.msn_sgemm_list(4, ctypes.cast(d, ctypes.c_void_p), 0, None, 0, None)):
    assert bad() == 1, L.msn_last_error()
# plane GEMMs (pgemm.hip): sizes, launch plans and the argument checks that return before a launch
print("plane bytes", L.msn_plane_bytes(66560, 384, 3), L.msn_plane_bytes(33, 17, 2), L.msn_plane_bytes(0, 4, 3))
print("pgemm ws", L.msn_pgemm_tn_workspace_bytes(66560, 1152, 384, 3), L.msn_pgemm_tn_workspace_bytes(66560, 384, 1536, 2),
      L.msn_pgemm_tn_workspace_bytes(1 << 22, 8192, 8192, 3),
      L.msn_pgemm_nt_workspace_bytes(66560, 384, 1536, 3, 0, 0, 0), L.msn_pgemm_nt_workspace_bytes(66560, 1536, 384, 3, 1, 4, 1),
      L.msn_plane_split_colsum_workspace_bytes(66560, 1152))
for bad in (lambda: L.msn_plane_split(None, 4, 4, 4, 3, 0, None, None, None, 0, None),
            lambda: L.msn_plane_split(fake, 4, 4, 4, 5, 0, fake, None, None, 0, None),
            lambda: L.msn_pgemm_nt(256, 130, 64, 3, fake, fake, fake, 132, 0, None, 0, None, 0, None, None, 0, None),
            lambda: L.msn_pgemm_nt(256, 128, 64, 3, fake, fake, fake, 128, 0, None, 5, None, 0, None, None, 0, None),
            lambda: L.msn_pgemm_nt(256, 128, 64, 4, fake, fake, fake, 128, 0, None, 0, None, 0, None, None, 0, None),
            lambda: L.msn_pgemm_tn(256, 128, 66, 3, fake, fake, fake, 66, None, 0, None),
            lambda: L.msn_pgemm_tn(256, 1 << 19, 64, 3, fake, fake, fake, 64, None, 0, None),
            lambda: L.msn_attention_bwd_planes(fake, 1152, None, 4, 6, 200, 64, 0.125, fake, 384, fake, fake, 384, 3, fake, None, None, 0, None),
            lambda: L.msn_attention_fwd_planes(fake, 1152, None, 4, 6, 200, 64, 0.125, fake, 384, fake, 3, fake, None),
            lambda: L.msn_attention_fwd_planes(fake, 1152, None, 4, 6, 65, 24, 0.125, fake, 384, fake, 3, fake, None),
            lambda: L.msn_attention_fwd_planes(fake, 1152, None, 4, 6, 65, 64, 0.125, fake, 384, fake, 4, fake, None),
            lambda: L.msn_attention_bwd_planes(fake, 1152, None, 4, 6, 65, 8, 0.125, fake, 384, fake, fake, 384, 3, fake, None, None, 0, None),
            lambda: L.msn_attention_bwd_planes(fake, 1152, None, 4, 6, 65, 64, 0.125, fake, 384, fake, fake, 384, 5, fake, None, None, 0, None),
            lambda: L.msn_attention_bwd_planes(fake, 1152, None, 4, 6, 65, 64, 0.125, fake, 384, fake, fake, 384, 3, fake, fake, fake, 16, None),
            lambda: L.msn_plane_split_f16(fake, 4, 4, 4, 0, fake, None, 0, None, None, 0, None),
            lambda: L.msn_pgemm_nt_f16(256, 128, 64, fake, None, fake, fake, fake, 128, None, 0, None, 0, None, None, 0, None),
            lambda: L.msn_pgemm_tn_f16(256, 128, 64, fake, fake, fake, None, fake, 64, None, 0, None),
            lambda: L.msn_plane_split_list(0, None, 3, None),
            lambda: L.msn_set_attention_planes(2), lambda: L.msn_set_attention_planes(64),
            lambda: L.msn_pgemm_nt(256, 128, 1 << 19, 3, fake, fake, fake, 128, 0, None, 0, None, 0, None, None, 0, None),
            lambda: L.msn_pgemm_nt(256, 128, 64, 3, fake, fake, fake, 1 << 20, 0, None, 0, None, 0, None, None, 0, None),
            lambda: L.msn_layernorm_fwd_planes(fake, 384, 8, 384, fake, fake, 1e-6, 5, fake, None, 0, fake, fake, None)):
    assert bad() == 1, L.msn_last_error()
# fused feed-forward of the narrow towers (ffn_planes.hip) and the clock probe: sizes and the checks that return before a launch
print("ffn", L.msn_ffn_supported(1000, 32, 128), L.msn_ffn_supported(1000, 64, 256), L.msn_ffn_bwd_workspace_bytes(225280, 32, 128),
      L.msn_ffn_bwd_workspace_bytes(10, 32, 128), L.msn_ffn_bwd_workspace_bytes(10, 48, 128))
for bad in (lambda: L.msn_ffn_fwd(fake, 32, 100, 64, 256, fake, fake, fake, fake, fake, 32, None),
            lambda: L.msn_ffn_fwd(fake, 30, 100, 32, 128, fake, fake, fake, fake, fake, 32, None),
            lambda: L.msn_ffn_fwd(None, 32, 100, 32, 128, fake, fake, fake, fake, fake, 32, None),
            lambda: L.msn_ffn_bwd(fake, 32, fake, 32, 100, 32, 128, fake, fake, fake, fake, 32, fake, fake, fake, fake, None, 0, None),
            lambda: L.msn_ffn_bwd(fake, 32, fake, 32, 100, 32, 96, fake, fake, fake, fake, 32, fake, fake, fake, fake, fake, 1 << 30, None),
            lambda: L.msn_clock_probe(None, 100, None), lambda: L.msn_clock_probe(fake, 0, None)):
    assert bad() == 1, L.msn_last_error()
# class-token attention (cls_attention.hip): the shapes it takes and the checks that return before a launch
print("cls attention", L.msn_cls_attention_supported(197, 64), L.msn_cls_attention_supported(257, 64), L.msn_cls_attention_supported(65, 32))
for bad in (lambda: L.msn_cls_attention_fwd(fake, 768, fake, 1536, 0, 2, 12, 300, 64, 0.125, fake, 768, fake, None),
            lambda: L.msn_cls_attention_fwd(fake, 768, fake, 1536, 0, 2, 12, 197, 32, 0.125, fake, 768, fake, None),
            lambda: L.msn_cls_attention_fwd(fake, 768, fake, 1530, 1, 2, 12, 197, 64, 0.125, fake, 768, fake, None),
            lambda: L.msn_cls_attention_fwd(None, 768, fake, 1536, 0, 2, 12, 197, 64, 0.125, fake, 768, fake, None),
            lambda: L.msn_cls_attention_bwd(fake, 768, fake, 1536, 1, 2, 12, 197, 64, 0.125, fake, 768, fake, fake, 768, fake, 760, fake, 1536, None),
            lambda: L.msn_cls_attention_bwd(fake, 768, fake, 1536, 0, 2, 12, 197, 64, 0.125, fake, 768, fake, fake, 768, fake, 768, None, 1536, None)):
    assert bad() == 1, L.msn_last_error()
print("HOST SANITIZER PROBE OK")
