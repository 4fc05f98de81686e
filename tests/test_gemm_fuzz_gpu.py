"""Seeded fuzz of msn_sgemm / msn_wgrad_bias over random shapes, layouts, leading dimensions and epilogues (the launch
plan switches between kernel families, split-K, tail split and fallbacks on shape and alignment alone, so odd sizes are
where a mistake would hide).  Reference: fp64 matmul on the same device."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _flat_launches_only():
    """These tests pin the planned FLAT launches (tile shapes, tail split, kernel families bit for bit); msn_sgemm's own
    detour through the work-list kernel (long-K under-filled products, tests/test_gemm_list_gpu.py) is switched off here."""
    from multimodal_supernovae_amd import ops
    ops.set_gemm_list(2)
    yield
    ops.set_gemm_list(1)


def _dim(rng, hi):
    kind = rng.random()
    if kind < 0.25:
        return rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 31, 32, 33, 63, 64, 65, 127, 128, 129])
    if kind < 0.6:
        return rng.randint(1, min(hi, 600))
    return rng.randint(1, hi)


@pytest.mark.parametrize("seed", range(6))
def test_sgemm_fuzz(seed):
    from multimodal_supernovae_amd import ops
    rng = random.Random(1000 + seed)
    g = torch.Generator(device="cuda").manual_seed(seed)
    for _ in range(40):
        op_a, op_b = rng.randint(0, 1), rng.randint(0, 1)
        M, N, K = _dim(rng, 70000 if rng.random() < 0.15 else 3000), _dim(rng, 1600), _dim(rng, 2100)
        if M * N > 40_000_000 or M * K > 40_000_000 or N * K > 40_000_000:
            continue
        pad_a, pad_b = rng.choice([0, 0, 4, 3]), rng.choice([0, 0, 4, 1])     # row strides beyond the logical width
        a_shape = (M, K) if op_a == 0 else (K, M)
        b_shape = (K, N) if op_b == 0 else (N, K)
        a = torch.randn(a_shape[0], a_shape[1] + pad_a, generator=g, device="cuda")[:, :a_shape[1]]
        b = torch.randn(b_shape[0], b_shape[1] + pad_b, generator=g, device="cuda")[:, :b_shape[1]]
        epi = rng.choice(["none", "none", "bias", "relu", "gelu", "add", "relu_bwd", "gelu_bwd"])
        if op_a == 1 and epi != "none":
            epi = "none"          # split-K products (opA = T) take the plain epilogue only
        bias = torch.randn(N, generator=g, device="cuda")
        aux = torch.randn(M, N, generator=g, device="cuda")
        ref = (a.double() if op_a == 0 else a.double().T) @ (b.double() if op_b == 0 else b.double().T)
        kw = {}
        if epi == "bias":
            kw, ref = dict(bias=bias), ref + bias.double()
        elif epi == "relu":
            kw, ref = dict(bias=bias, epilogue=ops.EPI_RELU), (ref + bias.double()).relu()
        elif epi == "gelu":
            kw, ref = dict(bias=bias, epilogue=ops.EPI_GELU), torch.nn.functional.gelu(ref + bias.double())
        elif epi == "add":
            kw, ref = dict(epilogue=ops.EPI_ADD, aux=aux), ref + aux.double()
        elif epi == "relu_bwd":
            kw, ref = dict(epilogue=ops.EPI_RELU_BWD, aux=aux), ref * (aux.double() > 0)
        elif epi == "gelu_bwd":
            kw, ref = dict(epilogue=ops.EPI_GELU_BWD, aux=aux), ref * aux.double()
        out = ops.sgemm(a, b, op_a, op_b, **kw)
        tol = 2e-5 * K ** 0.5 * (4.0 if epi in ("gelu_bwd", "relu_bwd") else 1.0) + 1e-5
        err = float((out.double() - ref).abs().max())
        assert err <= tol * (1.0 + float(ref.abs().max()) / max(K ** 0.5, 1.0)), (M, N, K, op_a, op_b, epi, pad_a, pad_b, err)


@pytest.mark.parametrize("seed", range(3))
def test_wgrad_bias_fuzz(seed):
    from multimodal_supernovae_amd import ops
    rng = random.Random(2000 + seed)
    g = torch.Generator(device="cuda").manual_seed(seed)
    for _ in range(25):
        K, M, N = _dim(rng, 60000), _dim(rng, 1600), _dim(rng, 800)
        if K * max(M, N) > 40_000_000:
            continue
        pad = rng.choice([0, 0, 4, 2])
        dy = torch.randn(K, M + pad, generator=g, device="cuda")[:, :M]
        x = torch.randn(K, N, generator=g, device="cuda")
        dw, db = ops.wgrad_bias(dy, x)
        tol = 2e-5 * K ** 0.5 + 1e-5
        assert float((dw.double() - dy.double().T @ x.double()).abs().max()) <= tol * 3, (K, M, N, pad)
        assert float((db.double() - dy.double().sum(0)).abs().max()) <= tol * 3, (K, M, N, pad)
