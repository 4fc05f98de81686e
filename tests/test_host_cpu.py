"""CPU-side checks of the boundary and the host mirror (no compute: the product has no CPU path)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, Fixture, golden_names

HEADER = os.path.join(ROOT, "include", "msn_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol_and_bindings_cover_the_header():
    import __graft_entry__ as entry
    entry.build()                                     # hipcc cross-compiles gfx950 without a GPU
    from multimodal_supernovae_amd import _lib
    handle = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/msn_hip.h but not exported"
    assert set(_lib.SIGNATURES) == set(names), set(_lib.SIGNATURES) ^ set(names)
    lib = _lib.lib()
    assert lib.msn_version() >= 100
    assert lib.msn_device_count() in (-1, 0) or torch.cuda.is_available()


def test_shape_errors_are_reported_without_a_gpu():
    """Argument validation happens before any launch, so it can be exercised on a CPU-only box."""
    from multimodal_supernovae_amd import _lib
    lib = _lib.lib()
    rc = lib.msn_sgemm(0, 1, 4, 4, 0, None, 4, None, 4, None, 4, None, 0, None, 0, 0, None, 0, None)
    assert rc == 1 and b"K must be positive" in lib.msn_last_error()
    fake = ctypes.c_void_p(4096)          # never dereferenced: validation rejects the call before any launch
    rc = lib.msn_infonce_fwd(fake, 300, 4, fake, 300, 4, fake, 300, 4, fake, 300, 4, 300, 0, fake, fake, fake, fake, fake,
                             fake, 0, None)
    assert rc == 1 and b"unsupported" in lib.msn_last_error()          # D = 300 > 256
    rc = lib.msn_infonce_fwd(fake, 12, 4, fake, 12, 4, fake, 12, 4, fake, 12, 4, 12, 0, fake, fake, fake, fake, fake,
                             fake, 0, None)
    assert rc == 1 and b"workspace" in lib.msn_last_error()            # D = 12 is fine now; the workspace is not
    assert lib.msn_set_attention_path(7) == 1
    assert lib.msn_sgemm_workspace_bytes(1, 0, 384, 1536, 66560) > 0      # wgrad takes the split-K path
    # 6240 tiles = 12 rounds of 512 + 96: those 96 run as 3 K-slabs each (tail split)
    assert lib.msn_sgemm_workspace_bytes(0, 1, 66560, 1536, 384) == 96 * 3 * 128 * 128 * 4
    assert lib.msn_sgemm_workspace_bytes(0, 1, 4096, 4096, 4096) == 0      # 1024 tiles: two full rounds
    # plane weight gradient: 15 tiles -> 17 reduction splits (one workgroup per CU); a reduction split never spans 2 GB of an operand
    # (the kernel addresses it through a buffer descriptor with 32-bit offsets): 131 072 row blocks of 1.5 MB -> 97 splits, not 1
    assert lib.msn_pgemm_tn_workspace_bytes(66560, 1152, 384, 3) == 17 * 1152 * 384 * 4
    assert lib.msn_pgemm_tn_workspace_bytes(1 << 22, 8192, 8192, 3) == 97 * 8192 * 8192 * 4
    rc = lib.msn_pgemm_tn(256, 1 << 19, 64, 3, fake, fake, fake, 64, None, 0, None)
    assert rc == 1 and b"2^19" in lib.msn_last_error()
    # the NT product's ring uses 32-bit offsets over K as well (8 row blocks x K / 16 column blocks x planes x 1 KB < 2 GB)
    rc = lib.msn_pgemm_nt(256, 128, 1 << 19, 3, fake, fake, fake, 128, 0, None, 0, None, 0, None, None, 0, None)
    assert rc == 1 and b"K must be below 2^19" in lib.msn_last_error()
    rc = lib.msn_pgemm_nt(256, 1 << 20, 64, 3, fake, fake, fake, 1 << 20, 0, None, 0, None, 0, None, None, 0, None)
    assert rc == 1 and b"2^20" in lib.msn_last_error()


def test_no_cpu_fallback():
    from multimodal_supernovae_amd import _lib, ops
    from multimodal_supernovae_amd.loss import clip_loss
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.MsnHipError):
        ops.sgemm(torch.randn(4, 4), torch.randn(4, 4))
    with pytest.raises(_lib.MsnHipError):
        clip_loss(torch.randn(8, 16), torch.randn(8, 16), torch.tensor(0.0), torch.tensor(0.0))


def _build(cfg):
    from multimodal_supernovae_amd.models_multimodal import LightCurveImageCLIP
    return LightCurveImageCLIP(enc_dim=cfg["enc_dim"], logit_scale=10.0, nband=cfg["nband"],
                               transformer_kwargs=cfg["transformer_kwargs"],
                               transformer_spectral_kwargs=cfg["transformer_spectral_kwargs"],
                               conv_kwargs=cfg["conv_kwargs"], meta_kwargs=cfg["meta_kwargs"],
                               combinations=cfg["combinations"], loss=cfg["loss"])


@pytest.mark.parametrize("name", [n for n in golden_names("clip_") if "sigmoid" not in n])
def test_state_dict_is_the_reference_state_dict(name):
    """Keys, shapes and dtypes equal the reference module's, so its checkpoints load strict=True."""
    f = Fixture(name)
    model = _build(f.cfg)
    sd = model.state_dict()
    assert list(sd.keys()) == list(f.P.keys()) or set(sd.keys()) == set(f.P.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(f.P[k].shape) and v.dtype == f.P[k].dtype, k
    model.load_state_dict(f.P, strict=True)
    assert abs(float(model.logit_bias) + 10.0) < 1e-6


def test_constructor_contract():
    from multimodal_supernovae_amd.models_multimodal import LightCurveImageCLIP, MLP
    from multimodal_supernovae_amd.transformer_utils import SelfAttention
    with pytest.raises(NotImplementedError):
        LightCurveImageCLIP(regression=True)
    with pytest.raises(AssertionError):
        SelfAttention(10, heads=3)                       # ref transformer_utils.py:21-23
    m = MLP(input_dim=4, hidden_dim=8, output_dim=2, num_layers=2, dropout=0.0)
    assert [k for k in m.state_dict()] == ["layers.0.weight", "layers.0.bias", "layers.3.weight", "layers.3.bias",
                                           "layers.6.weight", "layers.6.bias"]
    model = LightCurveImageCLIP(combinations=["spectral", "lightcurve"], loss="softmax")
    assert model.combinations == {"spectral", "lightcurve"} and not hasattr(model, "image_encoder")
    opt = model.configure_optimizers()["optimizer"]
    assert opt.param_groups[0]["lr"] == 1e-4 and opt.defaults["betas"] == (0.9, 0.999)


def test_trainer_moves_the_nine_tuple_and_drops_empty_placeholders():
    from multimodal_supernovae_amd.trainer import _to_device
    batch = (None, torch.zeros(2, 3), torch.zeros(2, 3), torch.ones(2, 3, dtype=torch.bool), torch.empty(0),
             torch.empty(0), torch.empty(0), torch.zeros(2), None)
    out = _to_device(batch, "cpu")
    assert len(out) == 9 and out[0] is None and out[4] is None and out[8] is None and out[3].dtype == torch.bool


def test_convolution_plan_queries_without_a_gpu():
    """Which convolutions the implicit-GEMM kernels take, and their workspace, are host-side decisions (no launch)."""
    from multimodal_supernovae_amd import _lib
    L = _lib.lib()
    ok = lambda *g: bool(L.msn_conv2d_implicit_ok(*g))
    # (B, H, W, C, C_out, kh, kw, sh, sw, ph, pw)
    assert ok(1024, 16, 16, 64, 64, 3, 3, 1, 1, 1, 1)          # ResNet-18 layer1
    assert ok(1024, 16, 16, 64, 128, 3, 3, 2, 2, 1, 1)         # strided entry of layer2
    assert ok(1024, 1, 200, 64, 128, 1, 5, 1, 1, 0, 2)         # 1-D CNN layer 2
    assert not ok(1024, 64, 64, 4, 64, 7, 7, 2, 2, 3, 3)       # the stem: 4 channels, several taps per K-step
    assert not ok(1024, 16, 16, 64, 32, 3, 3, 1, 1, 1, 1)      # 32-wide output: narrower than the kernels' tiles
    assert not ok(3, 5, 5, 64, 64, 3, 3, 1, 1, 1, 1)           # 75 output pixels: not whole K-steps for the weight gradient
    assert ok(1024, 1, 1024, 64, 128, 1, 5, 1, 1, 0, 2)        # the 1-D CNN on 1024-bin spectra (cfg4's third tower)
    assert ok(1, 1024, 1024, 64, 64, 3, 3, 1, 1, 1, 1)         # one 1024 x 1024 image: 2^20 pixels x 2^10 < 2^36
    assert not ok(64, 2048, 2048, 64, 64, 3, 3, 1, 1, 1, 1)    # 2^28 pixels: beyond the gather arithmetic
    assert not ok(100, 1024, 1024, 64, 64, 3, 3, 1, 1, 1, 1)   # < 2^27 pixels, but pixels x side >= 2^36 (multiply-shift division)
    assert L.msn_conv2d_workspace_bytes(1024, 16, 16, 64, 64, 3, 3, 1, 1, 1, 1) > 0      # split-K slabs of the weight gradient
    assert L.msn_conv2d_workspace_bytes(3, 5, 5, 64, 64, 3, 3, 1, 1, 1, 1) == 0
    # weight gradients with few rows are planned on short tiles; the workspace query answers for the same plan
    assert L.msn_wgrad_bias_workspace_bytes(32, 128, 225280) > 0 and L.msn_sgemm_workspace_bytes(1, 0, 32, 128, 225280) > 0


def test_pretraining_masks_reproduce_the_reference_draws():
    """get_continous_random_mask with the generator's seed gives the masks the REFERENCE drew (tests/golden/pretraining.npz);
    get_random_mask hides floor(f * n_observed) observed points per sample and consumes exactly one randperm(n_observed)
    per sample, in order (what keeps a seeded run on the reference's stream)."""
    import random
    import numpy as np
    from multimodal_supernovae_amd.models_pretraining import get_continous_random_mask, get_random_mask
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pretraining.npz"), allow_pickle=True)
    pad = torch.from_numpy(g["in/padding_mask"])
    random.seed(5)                                   # tools/gen_golden.py gen_pretraining
    mask_in, mask_pred = get_continous_random_mask(pad, 2, f_mask=0.3)
    assert torch.equal(mask_in, torch.from_numpy(g["in/mask_in"])) and torch.equal(mask_pred, torch.from_numpy(g["in/mask_pred"]))
    # ragged tail: 23 positions in 2 bands of 11 -- the last position is left as it is
    pad23 = torch.ones(3, 23, dtype=torch.bool)
    pad23[1, 7:11] = False
    mi, mp = get_continous_random_mask(pad23, 2, f_mask=0.25)
    assert mi[:, 22].all() and mp[:, 22].all() and not (mi & mp)[:, :22].any() and not (mp & ~pad23).any()
    for i in range(3):
        for k in range(2):
            run = mp[i, 11 * k: 11 * (k + 1)].nonzero().flatten()
            n = int(pad23[i, 11 * k: 11 * (k + 1)].sum())
            assert len(run) == int(n * 0.25) and (len(run) == 0 or int(run[-1] - run[0]) == len(run) - 1)   # one contiguous run

    torch.manual_seed(11)
    m, mpred = get_random_mask(pad, f_mask=0.3)
    after = torch.get_rng_state()
    n_obs = pad.sum(1)
    assert torch.equal(mpred.sum(1), (n_obs.double() * 0.3).floor().long()) and torch.equal(m | mpred, pad) and not (m & mpred).any()
    torch.manual_seed(11)
    first = torch.randperm(int(n_obs[0]))
    for n in n_obs.tolist()[1:]:
        torch.randperm(n)
    assert torch.equal(torch.get_rng_state(), after)                                    # same draws, same order
    hidden0 = pad[0].nonzero().flatten()[first[: int(int(n_obs[0]) * 0.3)]]             # the first sample hides the first-ranked points
    assert torch.equal(mpred[0].nonzero().flatten(), hidden0.sort().values)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_mask_helpers_match_reference_goldens(tag):
    """Both mask helpers against masks the reference drew (tools/gen_golden.py gen_random_masks: 2 / 3 / 1 bands, ragged
    samples, one sample with a single observed point per band), with the generator's seeds."""
    import random
    import numpy as np
    from multimodal_supernovae_amd.models_pretraining import get_continous_random_mask, get_random_mask
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "random_masks.npz"), allow_pickle=True)
    t = lambda k: torch.from_numpy(g[f"{tag}/{k}"])
    pad, nband, f = t("padding_mask"), int(t("nband")), float(t("f_mask"))
    torch.manual_seed(int(t("torch_seed")))
    m, mp = get_random_mask(pad, f_mask=f)
    assert torch.equal(m, t("mask")) and torch.equal(mp, t("mask_pred"))
    random.seed(int(t("python_seed")))
    cm, cmp_ = get_continous_random_mask(pad, nband, f_mask=f)
    assert torch.equal(cm, t("cmask")) and torch.equal(cmp_, t("cmask_pred"))


def test_saved_attention_weights_do_not_share_storage():
    """SelfAttention keeps toqueries / tokeys / tovalues side by side in one buffer (stacked_qkv); what state_dict() hands
    to a writer must be three independent tensors with the reference's keys and values (safetensors refuses shared
    storage; torch.save would write the whole buffer per key)."""
    from multimodal_supernovae_amd.transformer_utils import SelfAttention
    torch.manual_seed(0)
    att = SelfAttention(16, heads=2)
    before = {k: v.clone() for k, v in att.state_dict().items()}
    wcat = att.stacked_qkv()
    assert wcat.shape == (48, 16) and att.tokeys.weight.data_ptr() == att.toqueries.weight.data_ptr() + 16 * 16 * 4
    sd = att.state_dict()
    assert list(sd) == list(before)
    ptrs = set()
    for k, v in sd.items():
        assert torch.equal(v, before[k])
        assert v.untyped_storage().nbytes() == v.numel() * v.element_size(), k
        ptrs.add(v.untyped_storage().data_ptr())
    assert len(ptrs) == len(sd)
    fresh = SelfAttention(16, heads=2)
    fresh.load_state_dict(sd, strict=True)


def test_host_shim_under_address_and_ub_sanitizers():
    """The C-ABI's host side (validation, launch planning, workspace sizing, work-list descriptors) under ASan + UBSan:
    runs when tools/build_host_sanitized.sh has produced the sanitized library (a 90-second build, not part of the default
    CPU suite); the probe exercises every entry point that returns before a launch."""
    import glob
    import subprocess
    import sys
    lib = os.path.join(ROOT, "multimodal_supernovae_amd", "build_asan", "libmsn_hip_asan.so")
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not os.path.exists(lib) or not rt:
        pytest.skip("sanitized library not built (bash tools/build_host_sanitized.sh)")
    srcs = glob.glob(os.path.join(ROOT, "multimodal_supernovae_amd", "csrc", "*")) + [os.path.join(ROOT, "include", "msn_hip.h")]
    if os.path.getmtime(lib) < max(os.path.getmtime(f) for f in srcs):
        pytest.skip("sanitized library older than the sources (bash tools/build_host_sanitized.sh)")
    env = dict(os.environ, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_sanitizer_probe.py")], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0 and "HOST SANITIZER PROBE OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
