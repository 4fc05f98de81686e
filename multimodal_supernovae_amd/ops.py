"""Thin Python wrappers over the C-ABI (raw kernels, no autograd).  PyTorch is used only as the
owner of device memory and of the current HIP stream; every FLOP happens in libmsn_hip.so."""
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

OP_N, OP_T = 0, 1
EPI_NONE, EPI_RELU, EPI_GELU, EPI_RELU_BWD, EPI_GELU_BWD, EPI_ADD = range(6)


def _f32c(t, name):
    if t.device.type != "cuda":
        _lib.require_gpu()
        raise _lib.MsnHipError(f"{name} must live on the GPU (got {t.device})")
    if t.dtype != torch.float32:
        raise _lib.MsnHipError(f"{name} must be float32 (got {t.dtype})")
    return t


def _workspace(nbytes, device):
    return torch.empty((max(int(nbytes), 16) + 3) // 4, dtype=torch.float32, device=device)


def sgemm(a, b, op_a=OP_N, op_b=OP_T, bias=None, epilogue=EPI_NONE, aux=None, out=None):
    """C = epilogue(opA(a) @ opB(b) + bias).  `a`, `b` are 2-D, last-dim contiguous (row stride free)."""
    _f32c(a, "a"), _f32c(b, "b")
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = (a.shape if op_a == OP_N else (a.shape[1], a.shape[0]))
    Kb, N = (b.shape if op_b == OP_N else (b.shape[1], b.shape[0]))
    if K != Kb:
        raise _lib.MsnHipError(f"sgemm: inner dimensions differ ({K} vs {Kb})")
    c = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=a.device)
    assert c.stride(1) == 1 or c.numel() == 0
    L = lib()
    ws_bytes = L.msn_sgemm_workspace_bytes(op_a, op_b, M, N, K)
    ws = _workspace(ws_bytes, a.device) if ws_bytes else None
    if aux is not None:
        _f32c(aux, "aux")
        assert aux.stride(1) == 1
    check(L.msn_sgemm(op_a, op_b, M, N, K, ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(c),
                      c.stride(0) if c.numel() else max(N, 1), ptr(bias), epilogue, ptr(aux),
                      aux.stride(0) if aux is not None else 0, ptr(ws), ws_bytes, stream_ptr()), "msn_sgemm")
    return c


def colsum(x):
    """Sum over rows of a 2-D tensor -> (N,)."""
    _f32c(x, "x")
    assert x.dim() == 2 and x.stride(1) == 1
    M, N = x.shape
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    L = lib()
    nb = L.msn_colsum_workspace_bytes(M, N)
    ws = _workspace(nb, x.device)
    check(L.msn_colsum(ptr(x), x.stride(0), M, N, ptr(out), ptr(ws), nb, stream_ptr()), "msn_colsum")
    return out
