"""ctypes binding of libmsn_hip.so (the C-ABI declared in include/msn_hip.h).

There is NO CPU fallback: `lib()` raises if the shared library has not been built, and
`require_gpu()` raises if no MI355X is visible.  Everything in ops.py goes through here.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSN_HIP_LIB") or os.path.join(_HERE, "lib", "libmsn_hip.so")

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_f32 = ctypes.c_float
c_f64 = ctypes.c_double
c_ptr = ctypes.c_void_p
c_size = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/msn_hip.h one to one
SIGNATURES = {
    "msn_version": (c_int, []),
    "msn_last_error": (ctypes.c_char_p, []),
    "msn_clock_probe": (c_int, [c_ptr, c_int, c_ptr]),
    "msn_ffn_supported": (c_int, [c_i64, c_int, c_int]),
    "msn_ffn_fwd": (c_int, [c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    "msn_ffn_bwd_workspace_bytes": (c_size, [c_i64, c_int, c_int]),
    "msn_ffn_bwd": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr,
                            c_ptr, c_ptr, c_size, c_ptr]),
    "msn_cls_attention_supported": (c_int, [c_int, c_int]),
    "msn_cls_attention_fwd": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_int, c_f32, c_ptr, c_i64, c_ptr, c_ptr]),
    "msn_cls_attention_bwd": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_int, c_f32, c_ptr, c_i64, c_ptr, c_ptr,
                                      c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    "msn_cast_bf16_list": (c_int, [c_int, c_ptr, c_ptr]),
    "msn_device_count": (c_int, []),
    "msn_sgemm_workspace_bytes": (c_size, [c_int, c_int, c_i64, c_i64, c_i64]),
    "msn_sgemm": (c_int, [c_int, c_int, c_i64, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr,
                          c_int, c_ptr, c_i64, c_int, c_ptr, c_size, c_ptr]),
    "msn_sgemm_list_workspace_bytes": (c_size, [c_int, c_ptr]),
    "msn_sgemm_list": (c_int, [c_int, c_ptr, c_int, c_ptr, c_size, c_ptr]),
    "msn_set_gemm_list": (c_int, [c_int]),
    "msn_reset_gemm_counters": (c_int, [c_ptr]),
    "msn_set_gemm_variant": (c_int, [c_int]),
    "msn_set_gemm_tail_split": (c_int, [c_int]),
    "msn_set_gemm_tile_n": (c_int, [c_int]),
    "msn_wgrad_bias_workspace_bytes": (c_size, [c_i64, c_i64, c_i64]),
    "msn_wgrad_bias": (c_int, [c_i64, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_int, c_ptr, c_size, c_ptr]),
    "msn_colsum_workspace_bytes": (c_size, [c_i64, c_i64]),
    "msn_colsum": (c_int, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_layernorm_fwd": (c_int, [c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_f32, c_ptr, c_i64, c_ptr, c_ptr, c_ptr]),
    "msn_layernorm_bwd_workspace_bytes": (c_size, [c_i64, c_int]),
    "msn_layernorm_bwd": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_i64,
                                  c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_vit_tokens_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr]),
    "msn_vit_tokens_bwd": (c_int, [c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr]),
    "msn_l2norm_fwd": (c_int, [c_ptr, c_i64, c_i64, c_int, c_ptr, c_i64, c_ptr, c_ptr]),
    "msn_l2norm_bwd": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_i64, c_ptr]),
    "msn_time_embed_fwd": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr,
                                   c_ptr]),
    "msn_time_embed_bwd_workspace_bytes": (c_size, [c_i64, c_int, c_int]),
    "msn_time_embed_bwd": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_size,
                                   c_ptr]),
    "msn_masked_pool_fwd": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr]),
    "msn_masked_pool_bwd": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr]),
    "msn_mask_tokens": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr]),
    "msn_add_rows": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_ptr]),
    "msn_plane_bytes": (c_size, [c_i64, c_i64, c_int]),
    "msn_plane_split_colsum_workspace_bytes": (c_size, [c_i64, c_i64]),
    "msn_plane_split": (c_int, [c_ptr, c_i64, c_i64, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_plane_split_list": (c_int, [c_int, c_ptr, c_int, c_ptr]),
    "msn_plane_split_f16": (c_int, [c_ptr, c_i64, c_i64, c_i64, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_pgemm_nt_f16": (c_int, [c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_int, c_ptr, c_i64, c_ptr,
                                 c_ptr, c_size, c_ptr]),
    "msn_pgemm_tn_f16": (c_int, [c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_size, c_ptr]),
    "msn_plane_merge": (c_int, [c_ptr, c_int, c_i64, c_i64, c_ptr, c_i64, c_ptr]),
    "msn_pgemm_nt_colsum_workspace_bytes": (c_size, [c_i64, c_int]),
    "msn_pgemm_nt_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_int, c_int, c_int, c_int]),
    "msn_set_pgemm_tail_split": (c_int, [c_int]),
    "msn_pgemm_nt": (c_int, [c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_int, c_ptr, c_i64, c_ptr,
                             c_ptr, c_size, c_ptr]),
    "msn_pgemm_tn_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_int]),
    "msn_pgemm_tn": (c_int, [c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_size, c_ptr]),
    "msn_layernorm_fwd_planes": (c_int, [c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_f32, c_int, c_ptr, c_ptr, c_i64, c_ptr, c_ptr,
                                         c_ptr]),
    "msn_layernorm_bwd_planes": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr,
                                         c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_bgemm_nt": (c_int, [c_i64, c_int, c_int, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_int, c_ptr, c_i64,
                             c_ptr, c_ptr, c_size, c_ptr]),
    "msn_bgemm_nt_colsum_workspace_bytes": (c_size, [c_i64, c_int]),
    "msn_bgemm_tn_workspace_bytes": (c_size, [c_i64, c_int, c_int]),
    "msn_bgemm_tn": (c_int, [c_i64, c_int, c_int, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_size, c_ptr]),
    "msn_cast_bf16": (c_int, [c_ptr, c_i64, c_ptr, c_ptr]),
    "msn_attention_bf16_fwd": (c_int, [c_ptr, c_i64, c_int, c_int, c_int, c_f32, c_ptr, c_i64, c_ptr, c_ptr]),
    "msn_attention_bf16_bwd": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_int, c_int, c_int, c_f32, c_ptr,
                                       c_ptr, c_ptr, c_ptr, c_ptr]),
    "msn_cast_bf16_transposed": (c_int, [c_ptr, c_int, c_int, c_ptr, c_ptr]),
    "msn_bcolsum_workspace_bytes": (c_size, [c_i64, c_int]),
    "msn_bcolsum": (c_int, [c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_layernorm_fwd_bf16": (c_int, [c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_f32, c_ptr, c_i64, c_ptr, c_ptr, c_ptr]),
    "msn_layernorm_bwd_bf16": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr,
                                       c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_size, c_ptr]),
    "msn_attention_fwd": (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, c_ptr,
                                  c_int, c_int, c_int, c_int, c_int, c_f32, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    "msn_attention_bwd": (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, c_ptr,
                                  c_int, c_int, c_int, c_int, c_int, c_f32, c_ptr, c_i64, c_i64, c_ptr,
                                  c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64,
                                  c_ptr, c_i64, c_i64, c_ptr]),
    "msn_patchify": (c_int, [c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "msn_unpatchify": (c_int, [c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "msn_bn_workspace_bytes": (c_size, [c_i64, c_int]),
    "msn_batchnorm_fwd": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr, c_f32, c_int, c_f32, c_ptr, c_ptr, c_ptr, c_int,
                                  c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_dropout": (c_int, [c_ptr, c_i64, c_f32, ctypes.c_uint64, c_ptr, c_ptr, c_ptr]),
    "msn_dropout_dev": (c_int, [c_ptr, c_i64, c_f32, c_ptr, ctypes.c_uint64, c_ptr, c_ptr, c_ptr]),
    "msn_seed_advance": (c_int, [c_ptr, c_ptr]),
    "msn_augment_workspace_bytes": (c_size, []),
    "msn_augment_images": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_f32, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_augment_series": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_ptr, c_ptr]),
    "msn_masked_mse_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    "msn_masked_mse_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "msn_series_features": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_ptr, c_ptr]),
    "msn_relu_mask": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    "msn_im2col": (c_int, [c_ptr] + [c_int] * 10 + [c_ptr, c_ptr]),
    "msn_col2im": (c_int, [c_ptr] + [c_int] * 10 + [c_ptr, c_ptr]),
    "msn_im2col_tap": (c_int, [c_ptr] + [c_int] * 10 + [c_ptr, c_ptr]),
    "msn_col2im_tap": (c_int, [c_ptr] + [c_int] * 10 + [c_ptr, c_ptr]),
    "msn_conv_weight_relayout": (c_int, [c_ptr, c_i64, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "msn_conv2d_implicit_ok": (c_int, [c_int] * 11),
    "msn_conv2d_workspace_bytes": (c_size, [c_int] * 11),
    "msn_conv2d_fwd": (c_int, [c_ptr] + [c_int] * 4 + [c_ptr] + [c_int] * 7 + [c_ptr, c_int, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_conv2d_dgrad": (c_int, [c_ptr] + [c_int] * 4 + [c_ptr] + [c_int] * 5 + [c_ptr, c_ptr, c_size, c_ptr]),
    "msn_conv2d_wgrad": (c_int, [c_ptr, c_ptr] + [c_int] * 11 + [c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_pad_channels": (c_int, [c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr]),
    "msn_maxpool2d_fwd": (c_int, [c_ptr] + [c_int] * 7 + [c_ptr, c_ptr, c_ptr]),
    "msn_maxpool2d_bwd": (c_int, [c_ptr, c_ptr] + [c_int] * 7 + [c_ptr, c_ptr]),
    "msn_batchnorm_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr,
                                  c_ptr, c_size, c_ptr]),
    "msn_batchnorm_relu_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr,
                                       c_ptr, c_size, c_ptr]),
    "msn_bn_colsum": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_bn_mean_from_sum": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr]),
    "msn_bn_rstd_from_sqdev": (c_int, [c_ptr, c_i64, c_int, c_f32, c_f32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "msn_batchnorm_apply": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr]),
    "msn_bn_bwd_sums": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_bn_bwd_apply": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "msn_dwconv_gelu_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr]),
    "msn_dwconv_bwd_workspace_bytes": (c_size, [c_int, c_int, c_int]),
    "msn_dwconv_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr,
                               c_ptr, c_size, c_ptr]),
    "msn_radam_step": (c_int, [c_ptr, c_int, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i64, c_ptr]),
    "msn_radam_step_dev": (c_int, [c_ptr, c_int, c_i64, c_ptr, c_ptr, c_ptr]),
    "msn_set_attention_path": (c_int, [c_int]),
    "msn_set_attention_fused": (c_int, [c_int]),
    "msn_set_attention_planes": (c_int, [c_int]),
    "msn_attention_bwd_planes_workspace_bytes": (c_size, [c_int, c_int, c_int]),
    "msn_attention_fwd_planes": (c_int, [c_ptr, c_i64, c_ptr, c_int, c_int, c_int, c_int, c_f32, c_ptr, c_i64, c_ptr, c_int, c_ptr, c_ptr]),
    "msn_attention_bwd_planes": (c_int, [c_ptr, c_i64, c_ptr, c_int, c_int, c_int, c_int, c_f32, c_ptr, c_i64, c_ptr, c_ptr, c_i64,
                                         c_int, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_retrieval_rank": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_sigmoid_loss_fwd": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_int,
                                     c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_sigmoid_loss_bwd": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_int,
                                     c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_infonce_workspace_bytes": (c_size, [c_int] * 5),
    "msn_infonce_fwd": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr, c_i64, c_int,
                                c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "msn_infonce_bwd": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr, c_i64, c_int,
                                c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr,
                                c_ptr, c_size, c_ptr]),
}



class GemmDesc(ctypes.Structure):
    """msn_gemm_desc of include/msn_hip.h (one product of a work-list launch)."""
    _fields_ = [("opA", c_int), ("opB", c_int), ("M", c_i64), ("N", c_i64), ("K", c_i64), ("A", c_ptr), ("lda", c_i64),
                ("B", c_ptr), ("ldb", c_i64), ("C", c_ptr), ("ldc", c_i64), ("bias", c_ptr), ("epilogue", c_int),
                ("aux", c_ptr), ("ldaux", c_i64), ("colsum", c_ptr)]


_lib = None


class MsnHipError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (loudly) when it is missing -- the product has no other path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MsnHipError(
                f"{LIB_PATH} not found: build it with `python -m multimodal_supernovae_amd.build` "
                "(there is no CPU fallback for the contrastive hot path)")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def require_gpu():
    if not torch.cuda.is_available():
        raise MsnHipError("multimodal_supernovae_amd needs an MI355X (gfx950): torch.cuda.is_available() is "
                          "False and the hot path has no CPU implementation")


def check(rc, what=""):
    if rc != 0:
        msg = lib().msn_last_error().decode("utf-8", "replace")
        if rc == 2 and torch.cuda.is_available() and what.startswith(("msn_sgemm", "msn_wgrad", "msn_conv2d")):
            # a GEMM launch failed: tiles finished inside a launch count their contributors in module memory and rely on the
            # counters being zero between launches -- put THIS stream's slice back to zero so that a caller that catches the
            # error and carries on does not inherit half-counted tiles (msn_reset_gemm_counters: only the failing stream's
            # slice, never while the stream is capturing)
            try:
                lib().msn_reset_gemm_counters(stream_ptr())
            except Exception:      # noqa: BLE001 -- the original error is the one to report
                pass
        raise MsnHipError(f"{what or 'libmsn_hip'} failed (code {rc}): {msg}")


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
