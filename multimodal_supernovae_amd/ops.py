"""Thin Python wrappers over the C-ABI (raw kernels, no autograd).  PyTorch is used only as the
owner of device memory and of the current HIP stream; every FLOP happens in libmsn_hip.so."""
import ctypes

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

# bench.py sets this to a list to time every GEMM launch of one step with HIP events recorded on the
# launch stream: entries are (start_event, end_event, algorithmic_flops, (opA, opB, M, N, K, epilogue), has_aux).  None = no
# instrumentation.
GEMM_PROFILE = None
ATTN_PROFILE = None      # bench.py: list of (event, event, algorithmic flops, (B, H, Tq, Tk, head_dim), 'fwd' | 'bwd') per attention call

# Inner-product precision used by sgemm() when the caller passes none (include/msn_hip.h):
# PREC_F32 exact fp32 MFMA (default), PREC_BF16X3 split-bf16 (fp32-grade, ~1e-5), PREC_BF16 plain bf16.
PREC_F32, PREC_BF16X3, PREC_BF16 = 0, 1, 2
# Python-level precisions of the plane path (csrc/pgemm.hip; the wide ViT products run on pgemm_nt / pgemm_tn with
# operands resident as bf16 planes): "bf16x6" = 3 planes, 6 MFMA products, fp32 grade; "bf16x3p" = 2 planes, 3 products.
# Products the plane kernels do not take (narrow towers, small batches) fall back to msn_sgemm in f32 / bf16x3.
PREC_PLANES3, PREC_PLANES2 = 3, 4
_PREC_NAMES = {"f32": PREC_F32, "bf16x3": PREC_BF16X3, "bf16": PREC_BF16, "bf16x6": PREC_PLANES3, "bf16x3p": PREC_PLANES2}
_C_PRECISION = {PREC_F32: PREC_F32, PREC_BF16X3: PREC_BF16X3, PREC_BF16: PREC_BF16, PREC_PLANES3: PREC_F32,
                PREC_PLANES2: PREC_BF16X3}      # what msn_sgemm is asked for


def plane_count(precision=None):
    """Planes per operand of the plane path under `precision` (default: the one in force), or 0 outside it."""
    p = GEMM_PRECISION if precision is None else precision
    return 3 if p == PREC_PLANES3 else 2 if p == PREC_PLANES2 else 0
# Default "bf16x6": the wide products of the ViT towers run fp32-grade on the bf16 matrix cores from resident planes (the gate
# for that default: tests/test_pgemm_gpu.py::test_fp32_grade_gate_* + every golden / oracle test at unchanged tolerances;
# DESIGN.md section 4); every other product is the native fp32 MFMA kernel, exactly as under "f32".
GEMM_PRECISION = _PREC_NAMES[__import__("os").environ.get("MSN_GEMM_PRECISION", "bf16x6").lower()]


class gemm_precision:
    """Context manager: GEMMs issued inside use the given precision; autograd nodes created inside remember
    it for their backward (functional._remember_precision)."""

    def __init__(self, name):
        self.value = _PREC_NAMES[name] if isinstance(name, str) else int(name)

    def __enter__(self):
        global GEMM_PRECISION
        self.old, GEMM_PRECISION = GEMM_PRECISION, self.value

    def __exit__(self, *exc):
        global GEMM_PRECISION
        GEMM_PRECISION = self.old


def set_gemm_precision(name):
    """Default inner-product precision of every GEMM issued from this process: "f32" | "bf16x3" | "bf16"."""
    global GEMM_PRECISION
    GEMM_PRECISION = _PREC_NAMES[name]

def set_gemm_variant(mode):
    """fp32 GEMM kernel family (msn_set_gemm_variant): 0 register-staged, 1 / 2 / 3 LDS-DMA rings (3 = default),
    4 persistent workgroups with loader waves."""
    check(lib().msn_set_gemm_variant(int(mode)))


def set_gemm_tile_n(bn):
    """Tile width of products with N > 64 (msn_set_gemm_tile_n): 0 = planned (default), 64, 128 -- measurements / tests."""
    check(lib().msn_set_gemm_tile_n(int(bn)))


def set_gemm_tail_split(enabled):
    """Cut the partly filled last round of GEMM tiles into K-slabs (msn_set_gemm_tail_split); default on (True / 1:
    the last workgroup to arrive sums a tile's slabs inside the launch; 2: a finishing launch does; False / 0: off)."""
    check(lib().msn_set_gemm_tail_split(int(enabled)))


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


def sgemm(a, b, op_a=OP_N, op_b=OP_T, bias=None, epilogue=EPI_NONE, aux=None, out=None, precision=None):
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
    prof = GEMM_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(L.msn_sgemm(op_a, op_b, M, N, K, ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(c),
                      c.stride(0) if c.numel() else max(N, 1), ptr(bias), epilogue, ptr(aux),
                      aux.stride(0) if aux is not None else 0, _C_PRECISION[GEMM_PRECISION if precision is None else precision],
                      ptr(ws), ws_bytes, stream_ptr()), "msn_sgemm")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 2.0 * M * N * K, (op_a, op_b, M, N, K, epilogue), aux is not None))
    return c


def wgrad_bias(dy, x, precision=None, out=None):
    """Weight and bias gradient of y = x W^T + b in one call: (dy^T x, column sums of dy) (msn_wgrad_bias).
    `out` = (dw, db) contiguous destinations, e.g. row blocks of a stacked gradient."""
    _f32c(dy, "dy"), _f32c(x, "x")
    assert dy.dim() == 2 and x.dim() == 2 and dy.stride(1) == 1 and x.stride(1) == 1
    K, M = dy.shape
    if x.shape[0] != K:
        raise _lib.MsnHipError(f"wgrad_bias: row counts differ ({K} vs {x.shape[0]})")
    N = x.shape[1]
    if out is not None:
        dw, db = out
        assert dw.shape == (M, N) and dw.is_contiguous() and db.shape == (M,) and db.is_contiguous()
    else:
        dw = torch.empty((M, N), dtype=torch.float32, device=dy.device)
        db = torch.empty(M, dtype=torch.float32, device=dy.device)
    L = lib()
    nb = L.msn_wgrad_bias_workspace_bytes(M, N, K)
    ws = _workspace(nb, dy.device)
    prof = GEMM_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(L.msn_wgrad_bias(M, N, K, ptr(dy), dy.stride(0), ptr(x), x.stride(0), ptr(dw), max(N, 1), ptr(db),
                           _C_PRECISION[GEMM_PRECISION if precision is None else precision], ptr(ws), nb, stream_ptr()), "msn_wgrad_bias")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 2.0 * M * N * K, (OP_T, OP_N, M, N, K, EPI_NONE), False))
    return dw, db


def set_gemm_list(mode=1):
    """msn_set_gemm_list: 1 / True (default) = work-list launches for msn_sgemm_list and, by the planner's rule, for long-K under-filled
    single products; 0 / False = none at all (lists one by one: measurements, bit comparisons); 2 = lists only, single products never
    (tests that pin the flat kernels); 3 = lists + every single product the kernel can take (tests)."""
    check(lib().msn_set_gemm_list(int(mode)))


def _gemm_desc(a, b, op_a, op_b, c, bias=None, epilogue=EPI_NONE, aux=None, colsum_out=None):
    M, K = (a.shape if op_a == OP_N else (a.shape[1], a.shape[0]))
    Kb, N = (b.shape if op_b == OP_N else (b.shape[1], b.shape[0]))
    if K != Kb:
        raise _lib.MsnHipError(f"sgemm_list: inner dimensions differ ({K} vs {Kb})")
    assert a.stride(1) == 1 and b.stride(1) == 1 and c.stride(1) == 1 and tuple(c.shape) == (M, N)
    d = _lib.GemmDesc()
    d.opA, d.opB, d.M, d.N, d.K = op_a, op_b, M, N, K
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), c.stride(0)
    d.bias = bias.data_ptr() if bias is not None else None
    d.epilogue = epilogue
    d.aux, d.ldaux = (aux.data_ptr(), aux.stride(0)) if aux is not None else (None, 0)
    d.colsum = colsum_out.data_ptr() if colsum_out is not None else None
    return d


def sgemm_list(descs, precision=None, profile_key=None):
    """Up to three independent products (built by _gemm_desc) in one work-list launch (msn_sgemm_list)."""
    import ctypes
    n = len(descs)
    arr = (_lib.GemmDesc * n)(*descs)
    L = lib()
    nb = L.msn_sgemm_list_workspace_bytes(n, ctypes.cast(arr, ctypes.c_void_p))
    ws = _workspace(nb, torch.device("cuda", torch.cuda.current_device()))
    prof = GEMM_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(L.msn_sgemm_list(n, ctypes.cast(arr, ctypes.c_void_p), _C_PRECISION[GEMM_PRECISION if precision is None else precision],
                           ptr(ws), nb, stream_ptr()), "msn_sgemm_list")
    if prof is not None:
        ev1.record()
        flops = sum(2.0 * d.M * d.N * d.K for d in descs)
        nbytes = sum(4.0 * (d.M * d.K + d.K * d.N + d.M * d.N * (2 if d.aux else 1)) for d in descs)
        d0 = descs[0]
        prof.append((ev0, ev1, flops, profile_key or (8, n, d0.M, d0.N, d0.K, d0.epilogue), False, nbytes))


def dgrad_wgrad(dy, w, x, epilogue=EPI_NONE, aux=None, want_bias=True, precision=None, want_dx=True):
    """Both backward products of y = x W^T (+ b) in ONE launch: dx = epilogue(dy . W), dW = dy^T . x and (want_bias)
    db = column sums of dy (ref src/transformer_utils.py:45-47, 89, 102-106 backward).  dy (rows, n_out), w (n_out, n_in),
    x (rows, n_in), all with contiguous columns.  Returns (dx, dW, db | None)."""
    _f32c(dy, "dy"), _f32c(w, "w"), _f32c(x, "x")
    rows, n_out = dy.shape
    n_in = w.shape[1]
    dev = dy.device
    dw = torch.empty((n_out, n_in), dtype=torch.float32, device=dev)
    db = torch.empty(n_out, dtype=torch.float32, device=dev) if want_bias else None
    descs = []
    dx = None
    if want_dx:
        dx = torch.empty((rows, n_in), dtype=torch.float32, device=dev)
        descs.append(_gemm_desc(dy, w, OP_N, OP_N, dx, epilogue=epilogue, aux=aux))
    descs.append(_gemm_desc(dy, x, OP_T, OP_N, dw, colsum_out=db))
    sgemm_list(descs, precision, profile_key=(9, len(descs), rows, n_in, n_out, epilogue))
    return dx, dw, db


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


# ------------------------------------------------------------------------------------------ row ops
def _rows2d(t):
    """View (..., C) as (rows, C) without copying (last dim contiguous, uniform row stride)."""
    if t.dim() == 2:
        assert t.stride(1) == 1
        return t
    assert t.is_contiguous()
    return t.view(-1, t.shape[-1])


def layernorm_fwd(x, gamma, beta, eps=1e-5):
    x2 = _rows2d(_f32c(x, "x"))
    rows, cols = x2.shape
    y = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib().msn_layernorm_fwd(ptr(x2), x2.stride(0), rows, cols, ptr(gamma), ptr(beta), eps, ptr(y), cols,
                                  ptr(mean), ptr(rstd), stream_ptr()), "msn_layernorm_fwd")
    return y.view(x.shape), mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, add=None):
    dy2, x2 = _rows2d(_f32c(dy, "dy")), _rows2d(x)
    add2 = _rows2d(add) if add is not None else None
    rows, cols = x2.shape
    dx = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
    dg = torch.empty(cols, dtype=torch.float32, device=x.device)
    db = torch.empty(cols, dtype=torch.float32, device=x.device)
    L = lib()
    nb = L.msn_layernorm_bwd_workspace_bytes(rows, cols)
    ws = _workspace(nb, x.device)
    check(L.msn_layernorm_bwd(ptr(dy2), dy2.stride(0), ptr(x2), x2.stride(0), rows, cols, ptr(mean), ptr(rstd),
                              ptr(gamma), ptr(add2), add2.stride(0) if add2 is not None else 0, ptr(dx), cols, ptr(dg),
                              ptr(db), ptr(ws), nb, stream_ptr()), "msn_layernorm_bwd")
    return dx.view(x.shape), dg, db


def l2norm_fwd(x):
    x2 = _rows2d(_f32c(x, "x"))
    rows, cols = x2.shape
    y = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
    inv = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib().msn_l2norm_fwd(ptr(x2), x2.stride(0), rows, cols, ptr(y), cols, ptr(inv), stream_ptr()),
          "msn_l2norm_fwd")
    return y, inv


def l2norm_bwd(dy, y, inv):
    dy2, y2 = _rows2d(_f32c(dy, "dy")), _rows2d(y)
    rows, cols = y2.shape
    dx = torch.empty((rows, cols), dtype=torch.float32, device=y.device)
    check(lib().msn_l2norm_bwd(ptr(dy2), dy2.stride(0), ptr(y2), y2.stride(0), rows, cols, ptr(inv), ptr(dx), cols,
                               stream_ptr()), "msn_l2norm_bwd")
    return dx


def time_embed_fwd(x, t, w, bw, omega, band=None):
    """x, t: (B, T) contiguous -> (B, T, e)."""
    B, T = x.shape
    e = w.numel()
    nband = band.shape[0] if band is not None else 1
    out = torch.empty((B, T, e), dtype=torch.float32, device=x.device)
    check(lib().msn_time_embed_fwd(ptr(_f32c(x, "x")), ptr(_f32c(t, "t")), B, T, e, ptr(w), ptr(bw), ptr(omega),
                                   ptr(band), nband, ptr(out), stream_ptr()), "msn_time_embed_fwd")
    return out


def time_embed_bwd(dy, x, nband):
    B, T, e = dy.shape
    dev = dy.device
    dw = torch.empty(e, dtype=torch.float32, device=dev)
    dbw = torch.empty(e, dtype=torch.float32, device=dev)
    dband = torch.empty((nband, e), dtype=torch.float32, device=dev) if nband > 1 else None
    L = lib()
    nb = L.msn_time_embed_bwd_workspace_bytes(B, e, nband)
    ws = _workspace(nb, dev)
    check(L.msn_time_embed_bwd(ptr(_f32c(dy, "dy")), ptr(x), B, T, e, nband, ptr(dw), ptr(dbw), ptr(dband), ptr(ws),
                               nb, stream_ptr()), "msn_time_embed_bwd")
    return dw, dbw, dband


POOL_MEAN, POOL_MAX = 0, 1


def _mask_u8(mask):
    if mask is None:
        return None
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)
    if mask.dtype == torch.uint8:          # already the byte mask the kernels read (0 / 1): no conversion launch
        return mask.contiguous()
    return (mask != 0).contiguous().view(torch.uint8)


def masked_pool_fwd(x, mask_u8, mode):
    B, T, e = x.shape
    out = torch.empty((B, e), dtype=torch.float32, device=x.device)
    arg = torch.empty((B, e), dtype=torch.int32, device=x.device) if mode == POOL_MAX else None
    cnt = torch.empty(B, dtype=torch.float32, device=x.device) if mode == POOL_MEAN else None
    check(lib().msn_masked_pool_fwd(ptr(_f32c(x, "x")), ptr(mask_u8), B, T, e, mode, ptr(out), ptr(arg), ptr(cnt),
                                    stream_ptr()), "msn_masked_pool_fwd")
    return out, arg, cnt


def masked_pool_bwd(dout, mask_u8, T, mode, arg, cnt):
    B, e = dout.shape
    dx = torch.empty((B, T, e), dtype=torch.float32, device=dout.device)
    check(lib().msn_masked_pool_bwd(ptr(_f32c(dout, "dout")), ptr(mask_u8), B, T, e, mode, ptr(arg), ptr(cnt), ptr(dx),
                                    stream_ptr()), "msn_masked_pool_bwd")
    return dx


def add_rows(dst, src):
    """dst += src for two (rows, cols) row-strided views (in place)."""
    assert dst.dim() == 2 and src.shape == dst.shape and dst.stride(1) == 1 and src.stride(1) == 1
    check(lib().msn_add_rows(ptr(_f32c(dst, "dst")), dst.stride(0), ptr(_f32c(src, "src")), src.stride(0), dst.shape[0],
                             dst.shape[1], stream_ptr()), "msn_add_rows")
    return dst


def mask_tokens(x, mask_u8):
    y = torch.empty_like(x)
    e = x.shape[-1]
    check(lib().msn_mask_tokens(ptr(_f32c(x, "x")), ptr(mask_u8), x.numel() // e, e, ptr(y), stream_ptr()),
          "msn_mask_tokens")
    return y


# ----------------------------------------------------------------------------------------- attention
def _bt(t):
    """(ld, batch stride) of a (B, T, C) view whose last dim is contiguous."""
    assert t.dim() == 3 and t.stride(2) == 1
    return t.stride(1), t.stride(0)


def _needs_head_padding(hd, *tensors):
    """Heads wider than 32 run on the matrix cores only, which take widths that are a multiple of 4 on 16-byte aligned rows;
    anything else is padded with zero columns here (zero columns change no score and no output column)."""
    if hd <= 32:
        return False
    return hd % 4 != 0 or any(t.data_ptr() % 16 or t.stride(1) % 4 or t.stride(0) % 4 for t in tensors)


def _pad_heads(t, heads, hd, hp):
    B, T, _ = t.shape
    out = torch.zeros((B, T, heads, hp), dtype=torch.float32, device=t.device)
    out[..., :hd] = t.reshape(B, T, heads, hd)
    return out.view(B, T, heads * hp)


def attention_fwd(q, k, v, mask_u8, heads, scale, q_shared=False):
    """q: (B, Tq, E) (or (1, Tq, E) with q_shared), k, v: (B, Tk, E) views (column slices allowed)."""
    B, Tk, E = k.shape
    Tq = q.shape[1]
    hd = E // heads
    if _needs_head_padding(hd, q, k, v):
        hp = (hd + 3) // 4 * 4
        qp, kp, vp = (_pad_heads(t, heads, hd, hp) for t in (q, k, v))
        op, lse = attention_fwd(qp, kp, vp, mask_u8, heads, scale, q_shared)
        return op.view(B, Tq, heads, hp)[..., :hd].reshape(B, Tq, E), lse
    out = torch.empty((B, Tq, E), dtype=torch.float32, device=k.device)
    lse = torch.empty((B, heads, Tq, 2), dtype=torch.float32, device=k.device)
    ldq, qbs = _bt(q)
    if q_shared:
        qbs = 0
    ldk, kbs = _bt(k)
    ldv, vbs = _bt(v)
    prof = ATTN_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().msn_attention_fwd(ptr(_f32c(q, "q")), ldq, qbs, ptr(k), ldk, kbs, ptr(v), ldv, vbs, ptr(mask_u8),
                                  B, heads, Tq, Tk, hd, scale, ptr(out), E, Tq * E, ptr(lse), stream_ptr()),
          "msn_attention_fwd")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 4.0 * B * heads * Tq * Tk * hd, (B, heads, Tq, Tk, hd), "fwd"))
    return out, lse


def attention_bwd(q, k, v, mask_u8, heads, scale, out, lse, dout, dq, dk, dv, q_shared=False):
    """Writes into the provided dq (B, Tq, E), dk, dv (B, Tk, E) views."""
    B, Tk, E = k.shape
    Tq = out.shape[1]
    hd = E // heads
    if _needs_head_padding(hd, q, k, v, out, dout, dq, dk, dv):
        hp = (hd + 3) // 4 * 4
        qp, kp, vp, op, dop = (_pad_heads(t, heads, hd, hp) for t in (q, k, v, out, dout))
        gq, gk, gv = torch.empty_like(op), torch.empty_like(kp), torch.empty_like(vp)     # (dq per sample, also for a shared query)
        attention_bwd(qp, kp, vp, mask_u8, heads, scale, op, lse, dop, gq, gk, gv, q_shared)
        for dst, src in ((dq, gq), (dk, gk), (dv, gv)):
            dst.copy_(src.view(src.shape[0], src.shape[1], heads, hp)[..., :hd].reshape(src.shape[0], src.shape[1], E))
        return dq, dk, dv
    delta = torch.empty((B, heads, Tq), dtype=torch.float32, device=k.device)
    ldq, qbs = _bt(q)
    if q_shared:
        qbs = 0
    ldk, kbs = _bt(k)
    ldv, vbs = _bt(v)
    ldd, dbs = _bt(dout)
    lq, bq = _bt(dq)
    lk, bk = _bt(dk)
    lv, bv = _bt(dv)
    prof = ATTN_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().msn_attention_bwd(ptr(q), ldq, qbs, ptr(k), ldk, kbs, ptr(v), ldv, vbs, ptr(mask_u8), B, heads, Tq,
                                  Tk, hd, scale, ptr(out), E, Tq * E, ptr(lse), ptr(_f32c(dout, "dout")), ldd, dbs,
                                  ptr(delta), ptr(dq), lq, bq, ptr(dk), lk, bk, ptr(dv), lv, bv, stream_ptr()),
          "msn_attention_bwd")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 10.0 * B * heads * Tq * Tk * hd, (B, heads, Tq, Tk, hd), "bwd"))
    return dq, dk, dv


# ----------------------------------------------------------------------------------------- ConvMixer
def patchify(img, p):
    B, C, H, W = img.shape
    gh, gw = H // p, W // p
    out = torch.empty((B * gh * gw, C * p * p), dtype=torch.float32, device=img.device)
    check(lib().msn_patchify(ptr(_f32c(img, "img")), B, C, H, W, p, ptr(out), stream_ptr()), "msn_patchify")
    return out


def unpatchify(dpatches, shape, p):
    B, C, H, W = shape
    dimg = torch.empty(shape, dtype=torch.float32, device=dpatches.device)
    check(lib().msn_unpatchify(ptr(_f32c(dpatches, "dpatches")), B, C, H, W, p, ptr(dimg), stream_ptr()),
          "msn_unpatchify")
    return dimg


# Synchronised BatchNorm (data parallel): `distributed.enable_sync_batchnorm(group)` sets this to the process group
# whose ranks share batch statistics; None = per-replica statistics (what Lightning's DDP does by default).
BN_SYNC_GROUP = None


def _bn_sync_world():
    import torch.distributed as dist
    if BN_SYNC_GROUP is None or not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size(None if BN_SYNC_GROUP is True else BN_SYNC_GROUP)


def _bn_allreduce(t):
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=None if BN_SYNC_GROUP is True else BN_SYNC_GROUP)
    return t


def batchnorm_fwd(x, gamma, beta, running_mean, running_var, training, residual=None, momentum=0.1, eps=1e-5,
                  relu=False):
    rows, C = x.shape
    dev = x.device
    y = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=dev)
    rstd = torch.empty(C, dtype=torch.float32, device=dev)
    L = lib()
    nb = L.msn_bn_workspace_bytes(rows, C)
    ws = _workspace(nb, dev)
    world = _bn_sync_world() if training else 1
    if world > 1:
        # batch statistics over the rows of every rank (equal rows per rank): the two-pass scheme of the
        # single-process kernel with an all-reduce of C floats after each local column sum
        count = rows * world
        acc = torch.empty(C, dtype=torch.float32, device=dev)
        st = stream_ptr()
        check(L.msn_bn_colsum(ptr(_f32c(x, "x")), rows, C, None, ptr(acc), ptr(ws), nb, st), "msn_bn_colsum")
        _bn_allreduce(acc)
        check(L.msn_bn_mean_from_sum(ptr(acc), count, C, ptr(mean), st), "msn_bn_mean_from_sum")
        check(L.msn_bn_colsum(ptr(x), rows, C, ptr(mean), ptr(acc), ptr(ws), nb, st), "msn_bn_colsum")
        _bn_allreduce(acc)
        check(L.msn_bn_rstd_from_sqdev(ptr(acc), count, C, eps, momentum, ptr(mean), ptr(running_mean), ptr(running_var),
                                       ptr(rstd), st), "msn_bn_rstd_from_sqdev")
        check(L.msn_batchnorm_apply(ptr(x), rows, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(residual),
                                    1 if relu else 0, ptr(y), st), "msn_batchnorm_apply")
        return y, mean, rstd
    check(L.msn_batchnorm_fwd(ptr(_f32c(x, "x")), rows, C, ptr(gamma), ptr(beta), eps, 1 if training else 0, momentum,
                              ptr(running_mean), ptr(running_var), ptr(residual), 1 if relu else 0, ptr(y), ptr(mean),
                              ptr(rstd), ptr(ws), nb, stream_ptr()), "msn_batchnorm_fwd")
    return y, mean, rstd


def batchnorm_bwd(dy, x, pre, mean, rstd, gamma, training, relu_out=None):
    """relu_out: the saved output of a BatchNorm + ReLU -- dy is masked by relu_out > 0 inside the backward passes
    (single-process statistics; under synchronised BatchNorm the caller masks dy first)."""
    rows, C = x.shape
    dev = x.device
    dx = torch.empty_like(x)
    L = lib()
    nb = L.msn_bn_workspace_bytes(rows, C)
    ws = _workspace(nb, dev)
    world = _bn_sync_world() if training else 1
    if world > 1:
        # d beta / d gamma stay LOCAL sums (the gradient all-reduce adds the other ranks' share); dx needs the sums
        # over every rank's rows
        st = stream_ptr()
        sums = torch.empty(2 * C, dtype=torch.float32, device=dev)
        check(L.msn_bn_bwd_sums(ptr(_f32c(dy, "dy")), ptr(x), rows, C, ptr(mean), ptr(rstd), ptr(sums), ptr(ws), nb, st),
              "msn_bn_bwd_sums")
        total = _bn_allreduce(sums.clone())
        check(L.msn_bn_bwd_apply(ptr(dy), ptr(x), ptr(pre), rows, rows * world, C, ptr(mean), ptr(rstd), ptr(gamma),
                                 ptr(total), ptr(dx), st), "msn_bn_bwd_apply")
        return dx, sums[C:], sums[:C]
    dg = torch.empty(C, dtype=torch.float32, device=dev)
    db = torch.empty(C, dtype=torch.float32, device=dev)
    if relu_out is not None:
        assert pre is None
        check(L.msn_batchnorm_relu_bwd(ptr(_f32c(dy, "dy")), ptr(_f32c(relu_out, "relu_out")), ptr(x), rows, C, ptr(mean), ptr(rstd),
                                       ptr(gamma), 1 if training else 0, ptr(dx), ptr(dg), ptr(db), ptr(ws), nb, stream_ptr()),
              "msn_batchnorm_relu_bwd")
        return dx, dg, db
    check(L.msn_batchnorm_bwd(ptr(_f32c(dy, "dy")), ptr(x), ptr(pre), rows, C, ptr(mean), ptr(rstd), ptr(gamma),
                              1 if training else 0, ptr(dx), ptr(dg), ptr(db), ptr(ws), nb, stream_ptr()),
          "msn_batchnorm_bwd")
    return dx, dg, db


def dwconv_gelu_fwd(x, w, bias, B, gh, gw):
    C = x.shape[-1]
    k = w.shape[-1]
    pre, act = torch.empty_like(x), torch.empty_like(x)
    check(lib().msn_dwconv_gelu_fwd(ptr(_f32c(x, "x")), ptr(w), ptr(bias), B, gh, gw, C, k, ptr(pre), ptr(act),
                                    stream_ptr()), "msn_dwconv_gelu_fwd")
    return pre, act


def dwconv_bwd(dpre, x, w, B, gh, gw, add=None, want_bias=True):
    C = x.shape[-1]
    k = w.shape[-1]
    dev = x.device
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    dbias = torch.empty(C, dtype=torch.float32, device=dev) if want_bias else None
    L = lib()
    nb = L.msn_dwconv_bwd_workspace_bytes(B, C, k)
    ws = _workspace(nb, dev)
    check(L.msn_dwconv_bwd(ptr(_f32c(dpre, "dpre")), ptr(x), ptr(w), B, gh, gw, C, k, ptr(add), ptr(dx), ptr(dw),
                           ptr(dbias), ptr(ws), nb, stream_ptr()), "msn_dwconv_bwd")
    return dx, dw, dbias


# ------------------------------------------------------------------------------ ViT token assembly
def vit_tokens_fwd(patch_emb, cls, pos, B, T):
    e = patch_emb.shape[-1]
    tok = torch.empty((B, T, e), dtype=torch.float32, device=patch_emb.device)
    check(lib().msn_vit_tokens_fwd(ptr(_f32c(patch_emb, "patch_emb")), ptr(cls), ptr(pos), B, T, e, ptr(tok),
                                   stream_ptr()), "msn_vit_tokens_fwd")
    return tok


def vit_tokens_bwd(dtok):
    B, T, e = dtok.shape
    dpatch = torch.empty((B * (T - 1), e), dtype=torch.float32, device=dtok.device)
    check(lib().msn_vit_tokens_bwd(ptr(_f32c(dtok, "dtok")), B, T, e, ptr(dpatch), stream_ptr()), "msn_vit_tokens_bwd")
    return dpatch


# ------------------------------------------------------------------ conv plumbing (build-defined encoders)
def conv_out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


def im2col(x, kh, kw, sh, sw, ph, pw):
    """x: (B, H, W, C) channels-last -> (B*OH*OW, C*kh*kw)."""
    B, H, W, C = x.shape
    oh, ow = conv_out(H, kh, sh, ph), conv_out(W, kw, sw, pw)
    cols = torch.empty((B * oh * ow, C * kh * kw), dtype=torch.float32, device=x.device)
    check(lib().msn_im2col(ptr(_f32c(x, "x")), B, H, W, C, kh, kw, sh, sw, ph, pw, ptr(cols), stream_ptr()), "msn_im2col")
    return cols


def col2im(dcols, shape, kh, kw, sh, sw, ph, pw):
    B, H, W, C = shape
    dx = torch.empty(shape, dtype=torch.float32, device=dcols.device)
    check(lib().msn_col2im(ptr(_f32c(dcols, "dcols")), B, H, W, C, kh, kw, sh, sw, ph, pw, ptr(dx), stream_ptr()),
          "msn_col2im")
    return dx


def im2col_tap(x, kh, kw, sh, sw, ph, pw):
    """x: (B, H, W, C) channels-last, C % 4 == 0 -> (B*OH*OW, kh*kw*C), tap-major columns (channels fastest)."""
    B, H, W, C = x.shape
    oh, ow = conv_out(H, kh, sh, ph), conv_out(W, kw, sw, pw)
    cols = torch.empty((B * oh * ow, C * kh * kw), dtype=torch.float32, device=x.device)
    check(lib().msn_im2col_tap(ptr(_f32c(x, "x")), B, H, W, C, kh, kw, sh, sw, ph, pw, ptr(cols), stream_ptr()),
          "msn_im2col_tap")
    return cols


def col2im_tap(dcols, shape, kh, kw, sh, sw, ph, pw):
    B, H, W, C = shape
    dx = torch.empty(shape, dtype=torch.float32, device=dcols.device)
    check(lib().msn_col2im_tap(ptr(_f32c(dcols, "dcols")), B, H, W, C, kh, kw, sh, sw, ph, pw, ptr(dx), stream_ptr()),
          "msn_col2im_tap")
    return dx


def conv_weight_relayout(w, co, ci, taps, to_tap, ci_pad=None):
    """(co, ci, taps) -> (co, taps, ci_pad) when to_tap (zeros in the channels ci .. ci_pad-1), the inverse otherwise;
    returns a flat 2-D matrix of `co` rows."""
    ci_pad = ci if ci_pad is None else ci_pad
    if to_tap == 2:      # (tap, co, ci): the K-major weight operand of the implicit-GEMM dgrad
        out = torch.empty((taps * co, ci), dtype=torch.float32, device=w.device)
    else:
        out = torch.empty((co, (ci_pad if to_tap else ci) * taps), dtype=torch.float32, device=w.device)
    check(lib().msn_conv_weight_relayout(ptr(_f32c(w, "w")), co, ci, ci_pad, taps, int(to_tap), ptr(out),
                                         stream_ptr()), "msn_conv_weight_relayout")
    return out


def conv2d_implicit_ok(B, H, W, C, co, kh, kw, sh, sw, ph, pw):
    """Does the implicit-GEMM convolution (no column matrix) take this shape?  (msn_conv2d_implicit_ok)"""
    return bool(lib().msn_conv2d_implicit_ok(B, H, W, C, co, kh, kw, sh, sw, ph, pw))


class _conv_profile:
    """GEMM_PROFILE entry for an implicit-GEMM convolution launch (bench.py's roofline step): M x N x K of the product the kernel
    multiplies, and the bytes it must move -- image, weights, result; NOT the column matrix, which is never written."""

    def __init__(self, code, M, N, K, nbytes, epilogue=0):
        self.key, self.flops, self.nbytes = (code, 0, int(M), int(N), int(K), int(epilogue)), 2.0 * M * N * K, float(nbytes)
        self.prof = GEMM_PROFILE

    def __enter__(self):
        if self.prof is not None:
            self.ev0, self.ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.ev0.record()
        return self

    def __exit__(self, *exc):
        if self.prof is not None and exc[0] is None:
            self.ev1.record()
            self.prof.append((self.ev0, self.ev1, self.flops, self.key, False, self.nbytes))
        return False


def _conv_ws(geom, device):
    nb = lib().msn_conv2d_workspace_bytes(*geom)
    return (_workspace(nb, device), nb) if nb else (None, 0)


def conv2d_fwd(x, w_tap, kh, kw, sh, sw, ph, pw, bias=None, relu=False):
    """x: (B, H, W, C) channels-last, w_tap: (C_out, kh*kw*C) tap-major -> (B*OH*OW, C_out)."""
    B, H, W, C = x.shape
    co = w_tap.shape[0]
    y = torch.empty((B * conv_out(H, kh, sh, ph) * conv_out(W, kw, sw, pw), co), dtype=torch.float32, device=x.device)
    ws, nb = _conv_ws((B, H, W, C, co, kh, kw, sh, sw, ph, pw), x.device)
    Mo, K = y.shape[0], kh * kw * C
    with _conv_profile(20, Mo, co, K, 4.0 * (x.numel() + co * K + Mo * co), EPI_RELU if relu else EPI_NONE):
        check(lib().msn_conv2d_fwd(ptr(_f32c(x, "x")), B, H, W, C, ptr(_f32c(w_tap, "w_tap")), co, kh, kw, sh, sw, ph, pw,
                                   ptr(bias) if bias is not None else None, EPI_RELU if relu else EPI_NONE, ptr(y),
                                   ptr(ws) if ws is not None else None, nb, stream_ptr()), "msn_conv2d_fwd")
    return y


def conv2d_dgrad(dy, w_tco, shape, kh, kw, ph, pw):
    """dX of a stride-1 convolution: dy (B*OH*OW, C_out), w_tco (kh*kw*C_out, C) -> (B, H, W, C)."""
    B, H, W, C = shape
    co = dy.shape[1]
    dx = torch.empty(shape, dtype=torch.float32, device=dy.device)
    ws, nb = _conv_ws((B, H, W, C, co, kh, kw, 1, 1, ph, pw), dy.device)
    Mi, K = B * H * W, kh * kw * co
    with _conv_profile(21, Mi, C, K, 4.0 * (dy.numel() + K * C + Mi * C)):
        check(lib().msn_conv2d_dgrad(ptr(_f32c(dy, "dy")), B, H, W, C, ptr(_f32c(w_tco, "w_tco")), co, kh, kw, ph, pw, ptr(dx),
                                     ptr(ws) if ws is not None else None, nb, stream_ptr()), "msn_conv2d_dgrad")
    return dx


def conv2d_wgrad(dy, x, kh, kw, sh, sw, ph, pw, want_bias=False):
    """dW (C_out, kh*kw*C) tap-major [, dbias (C_out,)] from dy (B*OH*OW, C_out) and the channels-last input x."""
    B, H, W, C = x.shape
    co = dy.shape[1]
    dw = torch.empty((co, kh * kw * C), dtype=torch.float32, device=x.device)
    db = torch.empty(co, dtype=torch.float32, device=x.device) if want_bias else None
    ws, nb = _conv_ws((B, H, W, C, co, kh, kw, sh, sw, ph, pw), x.device)
    Mo, K = dy.shape[0], kh * kw * C
    with _conv_profile(22, co, K, Mo, 4.0 * (dy.numel() + x.numel() + co * K)):
        check(lib().msn_conv2d_wgrad(ptr(_f32c(dy, "dy")), ptr(_f32c(x, "x")), B, H, W, C, co, kh, kw, sh, sw, ph, pw, ptr(dw),
                                     ptr(db) if db is not None else None, ptr(ws) if ws is not None else None, nb, stream_ptr()),
              "msn_conv2d_wgrad")
    return dw, db


def pad_channels(x, cp):
    """Channels-last (..., C) -> (..., cp) with zeros in the added channels."""
    C = x.shape[-1]
    out = torch.empty(x.shape[:-1] + (cp,), dtype=torch.float32, device=x.device)
    check(lib().msn_pad_channels(ptr(_f32c(x, "x")), x.numel() // C, C, cp, ptr(out), stream_ptr()), "msn_pad_channels")
    return out


def maxpool2d_fwd(x, k, s, p):
    B, H, W, C = x.shape
    oh, ow = conv_out(H, k, s, p), conv_out(W, k, s, p)
    y = torch.empty((B, oh, ow, C), dtype=torch.float32, device=x.device)
    arg = torch.empty((B, oh, ow, C), dtype=torch.int32, device=x.device)
    check(lib().msn_maxpool2d_fwd(ptr(_f32c(x, "x")), B, H, W, C, k, s, p, ptr(y), ptr(arg), stream_ptr()),
          "msn_maxpool2d_fwd")
    return y, arg


def maxpool2d_bwd(dy, arg, shape, k, s, p):
    B, H, W, C = shape
    dx = torch.empty(shape, dtype=torch.float32, device=dy.device)
    check(lib().msn_maxpool2d_bwd(ptr(_f32c(dy, "dy")), ptr(arg), B, H, W, C, k, s, p, ptr(dx), stream_ptr()),
          "msn_maxpool2d_bwd")
    return dx


def relu_mask(dy, y):
    dm = torch.empty_like(dy)
    check(lib().msn_relu_mask(ptr(_f32c(dy, "dy")), ptr(y), dy.numel(), ptr(dm), stream_ptr()), "msn_relu_mask")
    return dm


def series_features(x, t, mask_u8, inv_norm):
    B, T = x.shape
    feat = torch.empty((B, T, 4), dtype=torch.float32, device=x.device)
    check(lib().msn_series_features(ptr(_f32c(x, "x")), ptr(_f32c(t, "t")), ptr(mask_u8), B * T, inv_norm, ptr(feat),
                                    stream_ptr()), "msn_series_features")
    return feat


# --------------------------------------------------------------------------------------------- dropout
# While a training step is being recorded as a HIP graph (trainer.GraphedTrainStep) seeds are TOKENS: a device-resident
# base that msn_seed_advance moves once per replay + the ordinal of the dropout call inside the step.
GRAPH_SEED = None           # [int64 device tensor of one element, next ordinal] during a capture


class SeedToken:
    __slots__ = ("base", "offset")

    def __init__(self, base, offset):
        self.base, self.offset = base, offset


def new_seed():
    """A fresh 63-bit dropout seed from torch's CPU generator (reproducible under torch.manual_seed); a SeedToken
    while a step is being recorded."""
    if GRAPH_SEED is not None:
        tok = SeedToken(GRAPH_SEED[0], (GRAPH_SEED[1] * 0xD1B54A32D192ED03 + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        GRAPH_SEED[1] += 1
        return tok
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def graph_seed_advance():
    check(lib().msn_seed_advance(ptr(GRAPH_SEED[0]), stream_ptr()), "msn_seed_advance")


def dropout(x, p, seed, residual=None, out=None):
    """y = dropout(x) (+ residual); the same (p, seed) applied to a gradient reproduces the mask."""
    x = _f32c(x if x.is_contiguous() else x.contiguous(), "x")
    y = out if out is not None else torch.empty_like(x)
    if isinstance(seed, SeedToken):
        check(lib().msn_dropout_dev(ptr(x), x.numel(), float(p), ptr(seed.base), seed.offset, ptr(residual), ptr(y),
                                    stream_ptr()), "msn_dropout_dev")
        return y
    check(lib().msn_dropout(ptr(x), x.numel(), float(p), seed, ptr(residual), ptr(y), stream_ptr()), "msn_dropout")
    return y


# ------------------------------------------------------------------- bf16-resident products (BASELINE cfg5 image tower)
BEPI_NONE, BEPI_GELU, BEPI_GELU_BWD, BEPI_ADD = range(4)


def _bf16c(t, name):
    if t.device.type != "cuda":
        _lib.require_gpu()
        raise _lib.MsnHipError(f"{name} must live on the GPU (got {t.device})")
    if t.dtype != torch.bfloat16:
        raise _lib.MsnHipError(f"{name} must be bfloat16 (got {t.dtype})")
    return t


def bgemm_supported(M, N, K):
    """Shapes msn_bgemm_nt / _tn take: reduction a multiple of 64, widths multiples of 8 (ViT-B: 768 / 2304 / 3072)."""
    return K % 64 == 0 and N % 8 == 0 and M > 0


def bgemm_nt(a, w, bias=None, epilogue=BEPI_NONE, aux=None, out_bf16=False, want_colsum=False):
    """C = epi(a @ w.T + bias): a (M, K) bf16, w (N, K) bf16, both with contiguous rows -> (M, N) fp32 or bf16.
    want_colsum: also return the column sums of C (computed in the epilogue) as the last element of the result."""
    _bf16c(a, "a"), _bf16c(w, "w")
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1 and a.shape[1] == w.shape[1]
    M, K = a.shape
    N = w.shape[0]
    c = torch.empty((M, N), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=a.device)
    if epilogue == BEPI_GELU and aux is None:
        aux = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    cs, ws, nb = None, None, 0
    if want_colsum:
        cs = torch.empty(N, dtype=torch.float32, device=a.device)
        nb = lib().msn_bgemm_nt_colsum_workspace_bytes(M, N)
        ws = _workspace(nb, a.device)
    prof = GEMM_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().msn_bgemm_nt(M, N, K, ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(c), N, 1 if out_bf16 else 0, ptr(bias),
                             epilogue, ptr(aux), aux.stride(0) if aux is not None else 0, ptr(cs), ptr(ws), nb, stream_ptr()),
          "msn_bgemm_nt")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 2.0 * M * N * K, (OP_N, OP_T, M, N, K, 100 + epilogue), aux is not None))
    out = (c, aux) if epilogue == BEPI_GELU else (c,)
    if want_colsum:
        out = out + (cs,)
    return out if len(out) > 1 else out[0]


def bgemm_tn(dy, x):
    """dW = dy.T @ x: dy (M, N) bf16, x (M, K) bf16 -> (N, K) fp32 (weight gradient; fixed-order split over M)."""
    _bf16c(dy, "dy"), _bf16c(x, "x")
    assert dy.dim() == 2 and x.dim() == 2 and dy.stride(1) == 1 and x.stride(1) == 1 and dy.shape[0] == x.shape[0]
    M, N = dy.shape
    K = x.shape[1]
    c = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    L = lib()
    nb = L.msn_bgemm_tn_workspace_bytes(M, N, K)
    ws = _workspace(nb, dy.device) if nb else None
    prof = GEMM_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(L.msn_bgemm_tn(M, N, K, ptr(dy), dy.stride(0), ptr(x), x.stride(0), ptr(c), K, ptr(ws), nb, stream_ptr()), "msn_bgemm_tn")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 2.0 * M * N * K, (OP_T, OP_N, N, K, M, 100), False))
    return c


def cast_bf16(x):
    """bf16 copy of a contiguous fp32 tensor (numel % 8 == 0)."""
    _f32c(x, "x")
    assert x.is_contiguous()
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib().msn_cast_bf16(ptr(x), x.numel(), ptr(y), stream_ptr()), "msn_cast_bf16")
    return y


def cast_bf16_t(w):
    """(R, C) fp32 -> (C, R) bf16: the transposed weight copy the input-gradient products multiply by."""
    _f32c(w, "w")
    assert w.dim() == 2 and w.is_contiguous()
    R, C = w.shape
    y = torch.empty((C, R), dtype=torch.bfloat16, device=w.device)
    check(lib().msn_cast_bf16_transposed(ptr(w), R, C, ptr(y), stream_ptr()), "msn_cast_bf16_transposed")
    return y


class _CastItem(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("R", ctypes.c_int64), ("C", ctypes.c_int64), ("transposed", ctypes.c_int), ("y", ctypes.c_void_p)]


def cast_bf16_list(mats, transposed=False):
    """bf16 copies (transposed: (C, R) copies) of a list of contiguous 2-D fp32 matrices from ONE launch (msn_cast_bf16_list)."""
    if not mats:
        return []
    items = (_CastItem * len(mats))()
    outs = []
    for i, w in enumerate(mats):
        _f32c(w, "w")
        assert w.dim() == 2 and w.is_contiguous()
        R, C = w.shape
        y = torch.empty((C, R) if transposed else (R, C), dtype=torch.bfloat16, device=w.device)
        items[i].x, items[i].R, items[i].C, items[i].transposed, items[i].y = ptr(w), R, C, 1 if transposed else 0, ptr(y)
        outs.append(y)
    check(lib().msn_cast_bf16_list(len(mats), ctypes.cast(items, ctypes.c_void_p), stream_ptr()), "msn_cast_bf16_list")
    return outs


def bcolsum(x):
    """Column sums of a bf16 (M, N) matrix -> (N,) fp32."""
    _bf16c(x, "x")
    assert x.dim() == 2 and x.stride(1) == 1
    M, N = x.shape
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    L = lib()
    nb = L.msn_bcolsum_workspace_bytes(M, N)
    ws = _workspace(nb, x.device)
    check(L.msn_bcolsum(ptr(x), x.stride(0), M, N, ptr(out), ptr(ws), nb, stream_ptr()), "msn_bcolsum")
    return out


def layernorm_fwd_bf16(x, gamma, beta, eps=1e-5):
    """LayerNorm whose output is written as bf16 (the next product's operand); returns (y_bf16, mean, rstd)."""
    x2 = _rows2d(_f32c(x, "x"))
    rows, cols = x2.shape
    y = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib().msn_layernorm_fwd_bf16(ptr(x2), x2.stride(0), rows, cols, ptr(gamma), ptr(beta), eps, ptr(y), cols,
                                       ptr(mean), ptr(rstd), stream_ptr()), "msn_layernorm_fwd_bf16")
    return y, mean, rstd


def layernorm_bwd_bf16(dy, x, mean, rstd, gamma, add=None, want_colsum=False):
    """LayerNorm backward returning (dx fp32, dx bf16 copy, dgamma, dbeta[, column sums of dx]).  dy: fp32, or bf16 as the
    input-gradient product of the bf16-resident trunk writes it."""
    dy_b = dy.dtype == torch.bfloat16
    dy2, x2 = _rows2d(dy if dy_b else _f32c(dy, "dy")), _rows2d(x)
    add2 = _rows2d(add) if add is not None else None
    rows, cols = x2.shape
    dx = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
    dxb = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device)
    dg = torch.empty(cols, dtype=torch.float32, device=x.device)
    db = torch.empty(cols, dtype=torch.float32, device=x.device)
    cs = torch.empty(cols, dtype=torch.float32, device=x.device) if want_colsum else None
    L = lib()
    nb = L.msn_layernorm_bwd_workspace_bytes(rows, cols) * 3 // 2
    ws = _workspace(nb, x.device)
    check(L.msn_layernorm_bwd_bf16(ptr(dy2), dy2.stride(0), ptr(x2), x2.stride(0), rows, cols, ptr(mean), ptr(rstd),
                                   ptr(gamma), ptr(add2), add2.stride(0) if add2 is not None else 0, ptr(dx), cols, ptr(dxb),
                                   ptr(dg), ptr(db), ptr(cs), 1 if dy_b else 0, ptr(ws), nb, stream_ptr()), "msn_layernorm_bwd_bf16")
    return (dx, dxb, dg, db, cs) if want_colsum else (dx, dxb, dg, db)


def attention_bf16_supported(T, head_dim):
    return head_dim == 64 and 0 < T <= 256


def attention_bf16_fwd(qkv, B, T, heads, scale):
    """qkv: (B*T, 3*heads*64) bf16 [q | k | v] -> (out (B*T, heads*64) bf16, lse (B, heads, T) fp32)."""
    _bf16c(qkv, "qkv")
    e = heads * 64
    assert qkv.dim() == 2 and qkv.shape == (B * T, 3 * e) and qkv.stride(1) == 1
    out = torch.empty((B * T, e), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B, heads, T), dtype=torch.float32, device=qkv.device)
    check(lib().msn_attention_bf16_fwd(ptr(qkv), qkv.stride(0), B, heads, T, scale, ptr(out), e, ptr(lse), stream_ptr()),
          "msn_attention_bf16_fwd")
    return out, lse


def attention_bf16_bwd(qkv, out, dout, lse, B, T, heads, scale, want_colsum=False):
    """Gradient w.r.t. the packed projection: (B*T, 3*heads*64) bf16 [dq | dk | dv] (+ its column sums, fp32)."""
    _bf16c(qkv, "qkv"), _bf16c(out, "out"), _bf16c(dout, "dout")
    assert dout.shape == out.shape and dout.stride(1) == 1 and out.stride(1) == 1
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, heads, T), dtype=torch.float32, device=qkv.device)
    cs = ws = None
    if want_colsum:
        cs = torch.empty(3 * heads * 64, dtype=torch.float32, device=qkv.device)
        ws = torch.empty((B, 3 * heads * 64), dtype=torch.float32, device=qkv.device)
    check(lib().msn_attention_bf16_bwd(ptr(qkv), qkv.stride(0), ptr(out), out.stride(0), ptr(dout), dout.stride(0), ptr(lse), B,
                                       heads, T, scale, ptr(dqkv), ptr(delta), ptr(cs), ptr(ws), stream_ptr()),
          "msn_attention_bf16_bwd")
    return (dqkv, cs) if want_colsum else dqkv


# ------------------------------------------------ fp32-grade products from resident bf16 planes (csrc/pgemm.hip)
PLANES = 3          # planes per operand when a caller does not say: 3 = fp32 grade (6 products), 2 = 3 products
if __import__("os").environ.get("MSN_ATTN_PLANES"):        # 0: long narrow-head attention on the exact-fp32 matrix-core kernels; 3 / 5: backward forms (A/B runs)
    check(lib().msn_set_attention_planes(int(__import__("os").environ["MSN_ATTN_PLANES"])))


F16_PLANES = 16     # `planes` code of the fp16 form: TWO fp16 planes + a power-of-two scale per matrix (msn_plane_split_f16)


class Planes:
    """An fp32 (R, C) matrix held as `planes` bf16 planes in the blocked layout of include/msn_hip.h (msn_plane_split):
    the operand format of pgemm_nt / pgemm_tn.  `buf` is a flat uint8 device tensor.  fp16 form (F16_PLANES): two planes
    of x 2^e and `scale` = two device floats (2^-e, the bits of the matrix' largest magnitude)."""
    __slots__ = ("buf", "R", "C", "planes", "scale")

    def __init__(self, buf, R, C, planes, scale=None):
        self.buf, self.R, self.C, self.planes, self.scale = buf, int(R), int(C), int(planes), scale

    @staticmethod
    def empty(R, C, planes, device):
        if planes == F16_PLANES:
            nb = lib().msn_plane_bytes(R, C, 2)
            return Planes(torch.empty(nb, dtype=torch.uint8, device=device), R, C, 2, torch.empty(2, dtype=torch.float32, device=device))
        nb = lib().msn_plane_bytes(R, C, planes)
        return Planes(torch.empty(nb, dtype=torch.uint8, device=device), R, C, planes)

    def to_float(self):
        assert self.scale is None, "bf16 planes only"
        y = torch.empty((self.R, self.C), dtype=torch.float32, device=self.buf.device)
        check(lib().msn_plane_merge(ptr(self.buf), self.planes, self.R, self.C, ptr(y), self.C, stream_ptr()), "msn_plane_merge")
        return y


def plane_split(x, planes=None, transposed=False, want_colsum=False, scale_of=None):
    """Planes of a 2-D fp32 matrix (row stride free) or, transposed=True, of its transpose.  want_colsum: also the column
    sums of x (a bias gradient) from the same pass.  planes = F16_PLANES: the fp16 form (a pass for the largest magnitude
    first; scale_of = the Planes of the same matrix in the other orientation: its scale is taken over, no extra pass)."""
    _f32c(x, "x")
    assert x.dim() == 2 and x.stride(1) == 1
    planes = PLANES if planes is None else planes
    R, C = x.shape
    out = Planes.empty(C, R, planes, x.device) if transposed else Planes.empty(R, C, planes, x.device)
    cs = ws = None
    nb = 0
    if want_colsum:
        cs = torch.empty(C, dtype=torch.float32, device=x.device)
        nb = lib().msn_plane_split_colsum_workspace_bytes(R, C)
        ws = _workspace(nb, x.device)
    if planes == F16_PLANES:
        if scale_of is not None:
            out.scale = scale_of.scale
        check(lib().msn_plane_split_f16(ptr(x), x.stride(0), R, C, 1 if transposed else 0, ptr(out.buf), ptr(out.scale),
                                        1 if scale_of is not None else 0, ptr(cs), ptr(ws), nb, stream_ptr()), "msn_plane_split_f16")
        return (out, cs) if want_colsum else out
    check(lib().msn_plane_split(ptr(x), x.stride(0), R, C, planes, 1 if transposed else 0, ptr(out.buf), ptr(cs), ptr(ws), nb,
                                stream_ptr()), "msn_plane_split")
    return (out, cs) if want_colsum else out


# ----------------------------------------------------------------------------------------- fused feed-forward (narrow towers)
def ffn_supported(M, emb, hidden):
    """Shapes msn_ffn_fwd / _bwd take (emb 32, hidden 128: the reference's spectrum transformer)."""
    return bool(lib().msn_ffn_supported(int(M), int(emb), int(hidden)))


def ffn_weight_planes(w1, w2):
    """The two operands msn_ffn_* keep in LDS: planes of ff.0.weight (4e, e) and of ff.2.weight (e, 4e) transposed -- one launch."""
    w1p, w2tp = plane_split_list([w1, w2], 3, transposed=[False, True])
    return w1p, w2tp


FFN_PROFILE = None      # a list while bench.py times the fused feed-forward launches of one step: (event, event, algorithmic flops, kind)


def ffn_fwd(x, w1p, w2tp, c1, c2):
    """z = x + relu(x W1^T + c1) W2^T + c2 without the hidden matrix in memory (msn_ffn_fwd).  x: (M, e) fp32, rows ld apart."""
    _f32c(x, "x")
    M, e = x.shape
    z = torch.empty((M, e), dtype=torch.float32, device=x.device)
    prof = FFN_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().msn_ffn_fwd(ptr(x), x.stride(0), M, e, c1.numel(), ptr(w1p.buf), ptr(w2tp.buf), ptr(_f32c(c1, "c1")), ptr(_f32c(c2, "c2")),
                            ptr(z), e, stream_ptr()), "msn_ffn_fwd")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 4.0 * M * e * c1.numel(), "fwd"))
    return z


def ffn_bwd(x, dz, w1p, w2tp, c1):
    """-> dx = dz + ((dz W2) o relu'(x W1^T + c1)) W1, dw1, dc1, dw2, dc2 (msn_ffn_bwd: the hidden tile is recomputed on chip)."""
    _f32c(x, "x"), _f32c(dz, "dz")
    M, e = x.shape
    hid = c1.numel()
    dx = torch.empty((M, e), dtype=torch.float32, device=x.device)
    dw1 = torch.empty((hid, e), dtype=torch.float32, device=x.device)
    dw2 = torch.empty((e, hid), dtype=torch.float32, device=x.device)
    dc1 = torch.empty(hid, dtype=torch.float32, device=x.device)
    dc2 = torch.empty(e, dtype=torch.float32, device=x.device)
    nb = lib().msn_ffn_bwd_workspace_bytes(M, e, hid)
    ws = _workspace(nb, x.device)
    prof = FFN_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().msn_ffn_bwd(ptr(x), x.stride(0), ptr(dz), dz.stride(0), M, e, hid, ptr(w1p.buf), ptr(w2tp.buf), ptr(_f32c(c1, "c1")),
                            ptr(dx), e, ptr(dw1), ptr(dc1), ptr(dw2), ptr(dc2), ptr(ws), nb, stream_ptr()), "msn_ffn_bwd")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 8.0 * M * e * hid, "bwd"))       # the four products of an unfused backward (the recomputation is not counted)
    return dx, dw1, dc1, dw2, dc2


def attention_bwd_planes(qkv, heads, scale, out, lse, dout, planes=None, want_colsum=True, mask_u8=None):
    """Backward of self-attention on the packed (B, T, 3 E) q | k | v matrix: the gradient dqkv as Planes (B T, 3 E) and, with
    want_colsum, its column sums -- one launch, no fp32 dqkv in memory (msn_attention_bwd_planes)."""
    B, T, E3 = qkv.shape
    E = E3 // 3
    planes = PLANES if planes is None else planes
    _f32c(qkv, "qkv"), _f32c(out, "out"), _f32c(dout, "dout")
    assert qkv.stride(2) == 1 and qkv.stride(0) == T * qkv.stride(1) and out.shape == (B, T, E) and dout.shape == (B, T, E)
    assert out.stride(0) == T * out.stride(1) and dout.stride(0) == T * dout.stride(1)
    dqkv = Planes.empty(B * T, E3, planes, qkv.device)
    cs = ws = None
    nb = 0
    if want_colsum:
        cs = torch.empty(E3, dtype=torch.float32, device=qkv.device)
        nb = lib().msn_attention_bwd_planes_workspace_bytes(B, heads, E // heads)
        ws = _workspace(nb, qkv.device)
    check(lib().msn_attention_bwd_planes(ptr(qkv), qkv.stride(1), ptr(mask_u8), B, heads, T, E // heads, scale, ptr(out),
                                         out.stride(1), ptr(lse), ptr(dout), dout.stride(1), planes, ptr(dqkv.buf), ptr(cs),
                                         ptr(ws), nb, stream_ptr()), "msn_attention_bwd_planes")
    return (dqkv, cs) if want_colsum else dqkv


def attention_fwd_planes(qkv, heads, scale, planes=None, mask_u8=None):
    """Forward of self-attention on the packed (B, T, 3 E) q | k | v matrix: (out (B, T, E) fp32, lse, out as Planes (B T, E)) from
    one launch -- the operand of the output projection without a split pass (msn_attention_fwd_planes)."""
    B, T, E3 = qkv.shape
    E = E3 // 3
    planes = PLANES if planes is None else planes
    _f32c(qkv, "qkv")
    assert qkv.stride(2) == 1 and qkv.stride(0) == T * qkv.stride(1)
    out = torch.empty((B, T, E), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((B, heads, T, 2), dtype=torch.float32, device=qkv.device)
    op = Planes.empty(B * T, E, planes, qkv.device)
    hd = E // heads
    prof = ATTN_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().msn_attention_fwd_planes(ptr(qkv), qkv.stride(1), ptr(mask_u8), B, heads, T, hd, scale, ptr(out), E, ptr(lse), planes,
                                         ptr(op.buf), stream_ptr()), "msn_attention_fwd_planes")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 4.0 * B * heads * T * T * hd, (B, heads, T, T, hd), "fwd"))
    return out, lse, op




def cls_attention_supported(T, head_dim):
    return bool(lib().msn_cls_attention_supported(int(T), int(head_dim)))


def _kv_rows(kv, B, T, e):
    assert kv.dim() == 2 and kv.shape[0] == B * T and kv.shape[1] == 2 * e and kv.stride(1) == 1 and kv.dtype in (torch.float32, torch.bfloat16)
    return 1 if kv.dtype == torch.bfloat16 else 0


def cls_attention_fwd(q, kv, T, heads, scale):
    """One query per (sample, head) over T keys (the class-token row of the ViT's last block): q (B, e) fp32, kv (B T, 2 e) fp32 or
    bf16 rows (keys | values) -> (out (B, e) fp32, probabilities (B, heads, T) fp32 for the backward).  msn_cls_attention_fwd."""
    B, e = q.shape
    _f32c(q, "q")
    bf = _kv_rows(kv, B, T, e)
    out = torch.empty((B, e), dtype=torch.float32, device=q.device)
    probs = torch.empty((B, heads, T), dtype=torch.float32, device=q.device)
    check(lib().msn_cls_attention_fwd(ptr(q), q.stride(0), ptr(kv), kv.stride(0), bf, B, heads, T, e // heads, scale, ptr(out), e,
                                      ptr(probs), stream_ptr()), "msn_cls_attention_fwd")
    return out, probs


def cls_attention_bwd(q, kv, T, heads, scale, out, probs, dout):
    """-> (dq (B, e) fp32, dkv (B T, 2 e) in kv's type).  msn_cls_attention_bwd."""
    B, e = q.shape
    bf = _kv_rows(kv, B, T, e)
    _f32c(dout, "dout")
    dq = torch.empty((B, e), dtype=torch.float32, device=q.device)
    dkv = torch.empty((B * T, 2 * e), dtype=kv.dtype, device=q.device)
    check(lib().msn_cls_attention_bwd(ptr(q), q.stride(0), ptr(kv), kv.stride(0), bf, B, heads, T, e // heads, scale, ptr(out),
                                      out.stride(0), ptr(probs), ptr(dout), dout.stride(0), ptr(dq), e, ptr(dkv), 2 * e, stream_ptr()),
          "msn_cls_attention_bwd")
    return dq, dkv


def attention_fwd_planes_supported(T, head_dim):
    return T <= 128 and head_dim % 16 == 0 and head_dim <= 64


def attention_bwd_planes_supported(T, head_dim):
    return T <= 128 and head_dim % 16 == 0 and head_dim <= 64 and 4 * (T + 3) * (head_dim + 4) * 4 + 4096 <= 160 * 1024


def set_attention_fused(on):
    """True / 1: the one-pass backward (default); 3: its form without the shared recomputation; False / 0: two kernels."""
    check(lib().msn_set_attention_fused(int(on)), "msn_set_attention_fused")


def set_attention_planes(mode):
    """1 (default): long sequences of heads up to 16 wide run fp32-grade on the bf16 matrix cores (csrc/attention_planes.hip);
    0: the exact-fp32 matrix-core kernels; 3 / 5: planes with the two-kernel / the one-pass backward everywhere (tests, A/B runs)."""
    check(lib().msn_set_attention_planes(int(mode)), "msn_set_attention_planes")


class _SplitItem(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("ldx", ctypes.c_int64), ("R", ctypes.c_int64), ("C", ctypes.c_int64),
                ("transposed", ctypes.c_int), ("out", ctypes.c_void_p)]


def plane_split_list(mats, planes=None, transposed=False):
    """Planes of several 2-D fp32 matrices (or of their transposes; `transposed` may be one flag per matrix) from ONE launch
    (msn_plane_split_list): the weights of every block of a tower."""
    planes = PLANES if planes is None else planes
    flags = list(transposed) if isinstance(transposed, (list, tuple)) else [transposed] * len(mats)
    if planes == F16_PLANES:
        return [plane_split(m, planes, transposed=f) for m, f in zip(mats, flags)]
    items = (_SplitItem * len(mats))()
    outs = []
    for it, m, tr in zip(items, mats, flags):
        _f32c(m, "x")
        assert m.dim() == 2 and m.stride(1) == 1
        R, C = m.shape
        out = Planes.empty(C, R, planes, m.device) if tr else Planes.empty(R, C, planes, m.device)
        it.x, it.ldx, it.R, it.C, it.transposed, it.out = m.data_ptr(), m.stride(0), R, C, 1 if tr else 0, out.buf.data_ptr()
        outs.append(out)
    check(lib().msn_plane_split_list(len(mats), ctypes.cast(items, ctypes.c_void_p), planes, stream_ptr()), "msn_plane_split_list")
    return outs


def pgemm_supported(M, N, K):
    """Shapes worth the plane kernels (256-row tiles, 16-deep K-steps): wide layers of the ViT towers."""
    return M >= 256 and N >= 128 and N % 16 == 0 and K >= 128 and K % 4 == 0


def pgemm_nt(a, w, bias=None, epilogue=EPI_NONE, aux=None, out_planes=False, want_colsum=False):
    """C = epilogue(a @ w.T + bias): a = Planes (M, K), w = Planes (N, K) -> fp32 (M, N) or, out_planes, Planes (M, N).
    GELU with aux=True allocates and returns the fp32 gelu' matrix as second result.  want_colsum: column sums of C last."""
    assert isinstance(a, Planes) and isinstance(w, Planes) and a.C == w.C and a.planes == w.planes
    M, K, N = a.R, a.C, w.R
    dev = a.buf.device
    c = Planes.empty(M, N, a.planes, dev) if out_planes else torch.empty((M, N), dtype=torch.float32, device=dev)
    ret_aux = False
    if epilogue == EPI_GELU and aux is True:
        aux = torch.empty((M, N), dtype=torch.float32, device=dev)
        ret_aux = True
    cs = torch.empty(N, dtype=torch.float32, device=dev) if want_colsum else None
    nb = lib().msn_pgemm_nt_workspace_bytes(M, N, K, a.planes, 1 if out_planes else 0, epilogue, 1 if want_colsum else 0)
    ws = _workspace(nb, dev) if nb else None
    prof = GEMM_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if a.scale is not None:
        assert w.scale is not None and not out_planes, "fp16 planes: both operands, fp32 result"
        check(lib().msn_pgemm_nt_f16(M, N, K, ptr(a.buf), ptr(a.scale), ptr(w.buf), ptr(w.scale), ptr(c), N, ptr(bias), epilogue,
                                     ptr(aux), aux.stride(0) if aux is not None else 0, ptr(cs), ptr(ws), nb, stream_ptr()),
              "msn_pgemm_nt_f16")
    else:
        check(lib().msn_pgemm_nt(M, N, K, a.planes, ptr(a.buf), ptr(w.buf), ptr(c.buf if out_planes else c), N,
                                 1 if out_planes else 0, ptr(bias), epilogue, ptr(aux), aux.stride(0) if aux is not None else 0,
                                 ptr(cs), ptr(ws), nb, stream_ptr()), "msn_pgemm_nt")
    if prof is not None:
        ev1.record()
        # bytes this launch MUST move in the formats it is given: plane operands are 2 bytes x planes per element (6 for the
        # fp32-grade form), the result fp32 or planes, the aux matrix (gelu' written / read, residual read) fp32
        pb = 2.0 * a.planes
        nbytes = pb * (M * K + N * K) + (pb if out_planes else 4.0) * M * N + (4.0 * M * N if aux is not None else 0.0)
        prof.append((ev0, ev1, 2.0 * M * N * K, (OP_N, OP_T, M, N, K, 200 + epilogue), aux is not None, nbytes))
    out = (c, aux) if ret_aux else (c,)
    if want_colsum:
        out = out + (cs,)
    return out if len(out) > 1 else out[0]


def pgemm_tn(dy, x):
    """dW = dy.T @ x: dy = Planes (M, N), x = Planes (M, K) -> fp32 (N, K) (weight gradient; fixed-order split over M)."""
    assert isinstance(dy, Planes) and isinstance(x, Planes) and dy.R == x.R and dy.planes == x.planes
    M, N, K = dy.R, dy.C, x.C
    dev = dy.buf.device
    c = torch.empty((N, K), dtype=torch.float32, device=dev)
    L = lib()
    nb = L.msn_pgemm_tn_workspace_bytes(M, N, K, dy.planes)
    ws = _workspace(nb, dev) if nb else None
    prof = GEMM_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if dy.scale is not None:
        assert x.scale is not None
        check(L.msn_pgemm_tn_f16(M, N, K, ptr(dy.buf), ptr(dy.scale), ptr(x.buf), ptr(x.scale), ptr(c), K, ptr(ws), nb, stream_ptr()),
              "msn_pgemm_tn_f16")
    else:
        check(L.msn_pgemm_tn(M, N, K, dy.planes, ptr(dy.buf), ptr(x.buf), ptr(c), K, ptr(ws), nb, stream_ptr()), "msn_pgemm_tn")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 2.0 * M * N * K, (OP_T, OP_N, N, K, M, 200), False, 2.0 * dy.planes * (M * N + M * K) + 4.0 * N * K))
    return c


def layernorm_fwd_planes(x, gamma, beta, eps, planes):
    """LayerNorm whose output goes straight to bf16 planes (the next product's operand); returns (Planes, mean, rstd)."""
    x2 = _rows2d(_f32c(x, "x"))
    rows, cols = x2.shape
    y = Planes.empty(rows, cols, planes, x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib().msn_layernorm_fwd_planes(ptr(x2), x2.stride(0), rows, cols, ptr(gamma), ptr(beta), eps, planes, ptr(y.buf), None, 0,
                                         ptr(mean), ptr(rstd), stream_ptr()), "msn_layernorm_fwd_planes")
    return y, mean, rstd


def layernorm_bwd_planes(dy, x, mean, rstd, gamma, planes, add=None, want_colsum=False):
    """LayerNorm backward returning (dx fp32, dx Planes, dgamma, dbeta[, column sums of dx])."""
    dy2, x2 = _rows2d(_f32c(dy, "dy")), _rows2d(x)
    add2 = _rows2d(add) if add is not None else None
    rows, cols = x2.shape
    dx = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
    dxp = Planes.empty(rows, cols, planes, x.device)
    dg = torch.empty(cols, dtype=torch.float32, device=x.device)
    db = torch.empty(cols, dtype=torch.float32, device=x.device)
    cs = torch.empty(cols, dtype=torch.float32, device=x.device) if want_colsum else None
    L = lib()
    nb = L.msn_layernorm_bwd_workspace_bytes(rows, cols) * 3 // 2
    ws = _workspace(nb, x.device)
    check(L.msn_layernorm_bwd_planes(ptr(dy2), dy2.stride(0), ptr(x2), x2.stride(0), rows, cols, ptr(mean), ptr(rstd), ptr(gamma),
                                     ptr(add2), add2.stride(0) if add2 is not None else 0, ptr(dx), cols, planes, ptr(dxp.buf),
                                     ptr(dg), ptr(db), ptr(cs), ptr(ws), nb, stream_ptr()), "msn_layernorm_bwd_planes")
    return (dx, dxp, dg, db, cs) if want_colsum else (dx, dxp, dg, db)


def set_pgemm_tail_split(enabled):
    """pgemm_nt: cut the tiles that do not fill a round of the persistent workgroups into K-segments (default on)."""
    check(lib().msn_set_pgemm_tail_split(int(bool(enabled))))
