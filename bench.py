#!/usr/bin/env python3
"""Headline benchmark: contrastive pairs/s of one full training step (forward, fused InfoNCE with
global negatives, backward, gradient all-reduce, fused RAdam) on synthetic data resident in HBM.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N ranks itself (N child processes, one
per GPU, RCCL rendezvous on 127.0.0.1; the parent makes no GPU call and only relays rank 0's JSON line).

Workload (BASELINE.json configs[2] towers; SURVEY.md section 8(d) "headline cfg3"): build-defined ViT-S/8
image encoder on 64x64x3 cutouts + the reference light-curve TransformerWithTimeEmbeddings (T = 200 =
2 bands x 100, emb 64, 8 heads, depth 5, mean pooling), n_out 32, enc_dim 128, fp32 (the reference
precision).  The metric is quoted at GLOBAL batch 1024, so the headline `value` is strong scaling: every rank
takes 1024 / N rows and all-gathers the embeddings so each rank contrasts its rows against all 1024.  The second
field `weak_scaling_256_per_gpu` keeps 256 rows per GPU (global 256 * N: the north star's >= 6x target at 8 GPUs).
"""
import argparse
import json
import math
import os
import sys
import time

# the pool's host driver only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails with
# "hipIpcGetMemHandle: invalid argument" (must be in the environment before the HIP runtime starts)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LC = dict(n_out=32, emb=64, heads=8, depth=5, dropout=0.0, time_norm=20583.369161312577, agg="mean")
CONV_PLACEHOLDER = dict(dim=8, depth=1, channels=3, kernel_size=5, patch_size=8, n_out=32, dropout_prob=0.0)
IMG, T_LC, NBAND, ENC_DIM, N_OUT = 64, 200, 2, 128, 32
GEMM_KERNEL_NAME = {"f32": "msn::sgemm_dma_kernel + msn::sgemm_kernel (fp32 v_mfma_f32_32x32x2_f32; all msn_sgemm launches and, where a tower has convolutions, the implicit-GEMM msn_conv2d_* launches of the same kernel)",
                    "bf16x3": "msn::bgemm_kernel<planes=2> (3 x v_mfma_f32_32x32x16_bf16 per algorithmic MAC tile)",
                    "bf16": "msn::bgemm_kernel<planes=1> (v_mfma_f32_32x32x16_bf16)",
                    "bf16x6": "msn::pgemm_nt_kernel<3> + msn::pgemm_tn_kernel<3> (operands resident as 3 bf16 planes, 6 x v_mfma_f32_32x32x16_bf16 per algorithmic MAC tile, two accumulator sets; narrow products: msn::sgemm_dma_kernel)",
                    "bf16x3p": "msn::pgemm_nt_kernel<2> + msn::pgemm_tn_kernel<2> (operands resident as 2 bf16 planes, 3 x v_mfma_f32_32x32x16_bf16 per algorithmic MAC tile)"}
GEMM_ARITHMETIC = {
    "bf16x6": "fp32-grade on the bf16 matrix cores: the wide ViT products multiply operands resident as 3 bf16 planes (an fp32 value "
              "splits exactly), 6 v_mfma_f32_32x32x16_bf16 products per multiply-add, fp32 accumulation in two accumulator sets; "
              "error against fp64 <= 1.5x the native fp32 MFMA kernel's on every headline shape (tests/test_pgemm_gpu.py::"
              "test_fp32_grade_gate_*); every other product: native fp32 MFMA",
    "bf16x3p": "two resident bf16 planes, 3 bf16 MFMA products (~1e-5 relative per product; opt-in)",
    "f32": "native fp32 MFMA (v_mfma_f32_32x32x2_f32) for every product",
    "bf16x3": "operands split hi + lo in registers inside the K loop, 3 bf16 MFMA products (opt-in)",
    "bf16": "operands rounded to bf16 (BASELINE cfg5 arithmetic)"}
LR, WD, LOGIT_SCALE = 3.716367614864064e-05, 0.000555522900788888, 19.545966923442453  # maven_pretrain_config.yaml


def _plane_side(prof, products):
    """Roofline figures of one profiled step: the plane launches (epilogue key >= 200) against the bf16 matrix peak in executed
    flops, the native fp32 launches against the fp32 matrix peak."""
    pl = [e for e in prof if e[3][5] >= 200]
    f32 = [e for e in prof if e[3][5] < 200]
    out = {}
    if pl and products:
        ms, fl = sum(e[0].elapsed_time(e[1]) for e in pl), sum(e[2] for e in pl)
        out["plane_launches"] = {"launches_per_step": len(pl), "ms_per_step_in_kernel": ms, "mfma_products_per_multiply_add": products,
                                 "executed_tflops": products * fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, "peak_tflops": 2500.0,
                                 "frac": products * fl / (ms * 1e-3) / 1e12 / 2500.0 if ms > 0 else 0.0}
    if f32:
        ms, fl = sum(e[0].elapsed_time(e[1]) for e in f32), sum(e[2] for e in f32)
        out["fp32_launches"] = {"launches_per_step": len(f32), "ms_per_step_in_kernel": ms,
                                "achieved_tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, "peak_tflops": 157.3,
                                "frac": fl / (ms * 1e-3) / 1e12 / 157.3 if ms > 0 else 0.0}
    return out


def synthetic_batch(b, seed, device):
    """images U[0,1); light curves N(0,1) with per-band sorted U[0,100) day stamps."""
    g = torch.Generator().manual_seed(seed)
    x_img = torch.rand(b, 3, IMG, IMG, generator=g)
    x_lc = torch.randn(b, T_LC, generator=g)
    per = T_LC // NBAND
    t_lc = torch.cat([torch.sort(torch.rand(b, per, generator=g) * 100.0, dim=1)[0] for _ in range(NBAND)], dim=1)
    mask = torch.ones(b, T_LC, dtype=torch.bool)
    return tuple(t.to(device) if t is not None else None
                 for t in (x_img, x_lc, t_lc, mask, None, None, None, None, None))


def build_model(device, seed=0):
    from multimodal_supernovae_amd.encoders import vit_s8
    from multimodal_supernovae_amd.models_multimodal import LightCurveImageCLIP
    torch.manual_seed(seed)
    model = LightCurveImageCLIP(enc_dim=ENC_DIM, logit_scale=LOGIT_SCALE, nband=NBAND, transformer_kwargs=LC,
                                conv_kwargs=CONV_PLACEHOLDER, combinations=["host_galaxy", "lightcurve"],
                                optimizer_kwargs={"weight_decay": WD}, lr=LR, loss="softmax")
    model.image_encoder = vit_s8(img_size=IMG, n_out=N_OUT)     # the encoder slot (ref models_multimodal.py:190)
    return model.to(device).train()


# ---- the other BASELINE.json configurations (parity-test cases; measured on request with --workload) ------------
SP_MAVEN = dict(n_out=32, emb=32, heads=2, depth=13, dropout=0.0, time_norm=17945.142213594805, agg="mean")
WORKLOADS = {
    "vit_s8_lc": "headline: ViT-S/8 + reference LC transformer (BASELINE cfg3 towers)",
    "resnet18_cnn1d": "BASELINE cfg2: ResNet-18 @64x64 + 1-D CNN light-curve encoder, local batch 256",
    "vit_s8_lc_cnn1d_sp": "BASELINE cfg4 towers: ViT-S/8 + LC transformer + 1-D CNN spectrum (1024 bins), 3-way InfoNCE",
    "vit_b16_bf16_lc": "BASELINE cfg5 towers: ViT-B/16 @224x224 with bf16 MFMA GEMMs + LC transformer",
    "maven_lc_sp": "reference-native Maven pretraining: LC (T=200, e64, h8, L5) + spectrum (T=220, e32, h2, L13)",
    "convmixer_lc_sp": "reference-native 3-tower: ConvMixer (64x64, p8, dim 32, depth 2) + LC + spectrum (T=1024)",
}


def build_workload(name, b, seed, device):
    """(model, batch, flops_per_pair or None) for the non-headline workloads."""
    from multimodal_supernovae_amd import encoders as E
    from multimodal_supernovae_amd.models_multimodal import LightCurveImageCLIP
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(0)

    def series(t, nband, lo, hi):
        per = t // nband
        x = torch.randn(b, t, generator=g)
        tt = torch.cat([torch.sort(torch.rand(b, per, generator=g) * (hi - lo) + lo, dim=1)[0] for _ in range(nband)], 1)
        return x, tt, torch.ones(b, t, dtype=torch.bool)

    img = lambda s: torch.rand(b, 3, s, s, generator=g)
    kw = dict(enc_dim=ENC_DIM, logit_scale=LOGIT_SCALE, nband=NBAND, transformer_kwargs=LC, transformer_spectral_kwargs=SP_MAVEN,
              conv_kwargs=CONV_PLACEHOLDER, optimizer_kwargs={"weight_decay": WD}, lr=LR, loss="softmax")
    lc = series(T_LC, NBAND, 0.0, 100.0)
    none3 = (None, None, None)
    if name == "resnet18_cnn1d":
        m = LightCurveImageCLIP(combinations=["host_galaxy", "lightcurve"], **kw)
        m.image_encoder, m.lightcurve_encoder = E.ResNet18(n_out=N_OUT), E.Conv1dEncoder(n_out=N_OUT, time_norm=100.0)
        batch = (img(64), *lc, *none3, None, None)
    elif name == "vit_s8_lc_cnn1d_sp":
        m = LightCurveImageCLIP(combinations=["host_galaxy", "lightcurve", "spectral"], **kw)
        m.image_encoder = E.vit_s8(img_size=64, n_out=N_OUT)
        m.spectral_encoder = E.Conv1dEncoder(n_out=N_OUT, time_norm=9000.0)
        batch = (img(64), *lc, *series(1024, 1, 3000.0, 9000.0), None, None)
    elif name == "vit_b16_bf16_lc":
        m = LightCurveImageCLIP(combinations=["host_galaxy", "lightcurve"], **kw)
        m.image_encoder = E.vit_b16(img_size=224, n_out=N_OUT)
        batch = (img(224), *lc, *none3, None, None)
    elif name == "maven_lc_sp":
        m = LightCurveImageCLIP(combinations=["lightcurve", "spectral"], **kw)
        batch = (None, *lc, *series(220, 1, 3000.0, 9000.0), None, None)
    elif name == "convmixer_lc_sp":
        kw["conv_kwargs"] = dict(dim=32, depth=2, channels=3, kernel_size=5, patch_size=8, n_out=N_OUT, dropout_prob=0.0)
        m = LightCurveImageCLIP(combinations=["host_galaxy", "lightcurve", "spectral"], **kw)
        batch = (img(64), *lc, *series(1024, 1, 3000.0, 9000.0), None, None)
    else:
        raise SystemExit(f"unknown workload {name}")
    return m.to(device).train(), tuple(t.to(device) if t is not None else None for t in batch)


def flops_per_pair(executed=False):
    """Algorithmic work model of BASELINE.md section 4 (2 flop / MAC, GEMM-shaped work only, train = 3 x fwd).
    executed=True: what the step really multiplies -- the ViT's last block is evaluated for the class-token row only
    (the head reads nothing else; output and gradients are identical), which drops the query / output projections, the
    attention rows and the MLP of the other T - 1 tokens of that one block."""
    def f_tr(t, e, l):
        return l * (24 * t * e * e + 4 * t * t * e)
    hw = (IMG // 8) ** 2
    t, e = 1 + hw, 384
    vit = f_tr(t, e, 12) + 2 * hw * (3 * 8 * 8) * e
    if executed:
        vit += -f_tr(t, e, 1) + (4 * t * e * e + 20 * e * e + 4 * t * e)     # k|v of every token; q, proj, MLP of one row
    lc = f_tr(T_LC, 64, 5)
    return 3.0 * (vit + lc)


def usable_cores(cap=32):
    """Threads the CPU leg uses: the scheduler affinity and the cgroup CPU quota bound what this process
    really owns (os.cpu_count() reports the whole host); capped so a shared host is not oversubscribed."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sample_b, steps, budget_s=12.0):
    """The oracle (CPU restatement, parity-pinned against the reference) timed on this host's cores on a
    bounded sample of the same workload: full train step incl. RAdam, `sample_b` pairs per step -- once on every
    core this process owns and once on a single thread."""
    from oracle import clip as oclip
    from oracle.build_defined import vision_transformer
    from oracle import encoders as oenc
    from oracle import loss as oloss
    cores = usable_cores()
    model = build_model("cpu")
    P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    batch = synthetic_batch(sample_b, 4321, "cpu")
    opt = oclip.RAdam([v for v in P.values() if v.requires_grad], lr=LR, weight_decay=WD)

    def step():
        opt.zero_grad()
        h = vision_transformer(P, "image_encoder.", batch[0], patch=8, heads=6, depth=12)
        e_img = oclip.l2_normalise(oenc.linear(P, "image_projection", h))
        h = oenc.transformer_with_time_embeddings(P, "lightcurve_encoder.", batch[1][..., None], batch[2], batch[3],
                                                  emb=LC["emb"], heads=LC["heads"], depth=LC["depth"],
                                                  time_norm=LC["time_norm"], nband=NBAND, agg="mean")
        e_lc = oclip.l2_normalise(oenc.linear(P, "lightcurve_projection", h))
        loss = oloss.clip_loss_multimodal([e_img, e_lc], P["logit_scale"], P["logit_bias"])
        loss.backward()
        opt.step()
        return float(loss.detach())

    def timed(threads, n_steps, budget):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        step()                                  # first step also warms the allocator / thread pool
        first = time.perf_counter() - t0
        if first > budget:                      # keep the whole leg bounded
            return 1, first
        if n_steps <= 0:                        # default: as many steps as fit in the budget, judged by the first one
            n_steps = max(2, min(40, int(budget / max(first, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        return n_steps, (time.perf_counter() - t0) / n_steps

    n_all, dt_all = timed(cores, steps, budget_s)
    n_one, dt_one = timed(1, 0, budget_s)
    torch.set_num_threads(cores)
    return {"value": sample_b / dt_all, "unit": "pairs/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model_name(), "host_cpus": os.cpu_count(),
            "single_thread": {"value": sample_b / dt_one, "unit": "pairs/s", "cores": 1, "steps": n_one,
                              "s_per_step": dt_one},
            "sample": f"{n_all} full train steps (fwd + InfoNCE + bwd + RAdam) of the same two-tower workload at batch "
                      f"{sample_b} on {cores} host threads (torch {torch.__version__} CPU fp32), {dt_all:.2f} s/step; "
                      f"single thread: {n_one} steps, {dt_one:.2f} s/step"}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks as CHILD processes (fresh interpreters: nothing is
    exec'ed over a process that has touched the GPU, and this parent never touches it -- torch.cuda.device_count() does
    not initialise HIP), one per GPU, rendezvous on 127.0.0.1; relay rank 0's JSON line.  With fewer GPUs than ranks
    (a one-GPU box rehearsing the flow) the ranks share devices round-robin over gloo: RCCL refuses two ranks on one
    device."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    n_dev = torch.cuda.device_count()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.setdefault("OMP_NUM_THREADS", "4")             # N ranks share the host: no N x all-cores thread pools
        if n_dev < n:
            env.setdefault("MSN_DIST_BACKEND", "gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's line is read by a thread; the parent watches the children: if one dies the others would wait for it inside
    # a collective for ever, so they are stopped (by the PIDs started here) and the failure is reported
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=30)
    # ONE JSON line on stdout: anything else rank 0 wrote there (gloo announces its connections on stdout) goes to stderr
    for line in b"".join(chunks).decode().splitlines():
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    codes = [p.returncode for p in procs]
    if failed is not None or any(codes):
        raise SystemExit(f"rank {failed} failed; exit codes {codes}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--per-gpu-batch", type=int, default=0,
                    help="weak scaling instead of the default: fix the rows PER GPU (global = N x this)")
    ap.add_argument("--global-batch", type=int, default=1024,
                    help="strong scaling (the default, the metric's definition): fix the GLOBAL batch and give each "
                         "rank global / N rows")
    ap.add_argument("--no-weak", action="store_true", help="skip the second measurement at 256 rows per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra split-bf16 measurement")
    ap.add_argument("--no-three-tower", action="store_true",
                    help="skip the third field (image + light-curve + 1024-bin-spectrum workload at 256 / 1024 rows per GPU)")
    ap.add_argument("--workload", default="vit_s8_lc", choices=list(WORKLOADS),
                    help="vit_s8_lc = the headline (default); the others are the remaining BASELINE.json configurations")
    ap.add_argument("--serial-towers", action="store_true",
                    help="towers one after the other on one stream (profiling: with a tower per stream rocprofv3's "
                         "per-kernel durations include whatever the other stream ran meanwhile)")
    ap.add_argument("--gemm-table", action="store_true", help="per-shape GEMM timing of one step on stderr")
    ap.add_argument("--gemm-variant", type=int, default=3, choices=[0, 1, 2, 3, 4],
                    help="fp32 GEMM kernel family: 0 register-staged, 1 / 2 / 3 LDS-DMA rings (3 = default)")
    ap.add_argument("--gemm-tail", type=int, default=1, choices=[0, 1, 2],
                    help="tail tiles of the fp32 GEMMs: 1 = K-slabs summed by the last workgroup to arrive (default), 2 = by a finishing launch, 0 = unsplit")
    ap.add_argument("--gemm-precision", default="bf16x6", choices=["f32", "bf16x6", "bf16x3p", "bf16x3", "bf16"],
                    help="inner-product arithmetic of the wide GEMMs: bf16x6 (default) = fp32-grade from three resident bf16 planes, 6 bf16 MFMA "
                         "products, error within 1.5x of the native kernel (tests/test_pgemm_gpu.py gate); f32 = native fp32 MFMA; "
                         "bf16x3p = two resident planes, 3 products; bf16x3 = split in registers, 3 products)")
    ap.add_argument("--graphed", action="store_true",
                    help="record the training step as HIP graphs (trainer.GraphedTrainStep; N > 1: segments between the host-driven "
                         "exchanges) and time its replays -- what launch-bound batches (the reference's 32 ... 256, or the "
                         "128 rows per GPU of the global batch on 8 GPUs) gain")
    ap.add_argument("--sync-batchnorm", action="store_true",
                    help="N > 1, BatchNorm towers (resnet18_cnn1d, convmixer_lc_sp): batch statistics over all ranks")
    ap.add_argument("--cpu-sample-batch", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=0, help="timed oracle steps (0 = fill about 12 s)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])

    from multimodal_supernovae_amd import _lib, distributed as D, ops
    _lib.require_gpu()
    ops.set_gemm_precision(args.gemm_precision)
    ops.set_gemm_variant(args.gemm_variant)
    ops.set_gemm_tail_split(args.gemm_tail)
    rank, local, world = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    device = torch.device("cuda", local)
    if args.per_gpu_batch:
        b, scaling = args.per_gpu_batch, "weak"
    else:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not divisible by {world} ranks")
        b, scaling = args.global_batch // world, "strong"
    headline = args.workload == "vit_s8_lc"
    if not headline:
        args.no_alt, args.no_cpu_baseline, args.no_weak, args.no_three_tower = True, True, True, True

    def make(rows):
        if headline:
            return build_model(device), synthetic_batch(rows, 1234 + rank, device)
        return build_workload(args.workload, rows, 1234 + rank, device)

    model, batch = make(b)
    if args.serial_towers:
        model.concurrent_towers = False
    D.broadcast_module(model)
    if args.sync_batchnorm and world > 1:
        D.enable_sync_batchnorm()
    opt = model.configure_optimizers()["optimizer"]
    # N > 1: bucketed SUM all-reduce launched from autograd hooks, under backward (graph replay: every bucket after backward)
    reducer = D.GradientReducer(model.parameters(), overlap=not args.graphed)

    from multimodal_supernovae_amd.trainer import _backward_seed

    from multimodal_supernovae_amd import markers     # roctx ranges (MSN_ROCTX=1): readable rocprofv3 timelines

    def make_step(model, opt, reducer, batch):
        def step():
            opt.zero_grad(set_to_none=True)
            with markers.range("forward + loss"):
                loss = model.training_step(batch, 0)
            with markers.range("backward"):
                loss.backward(_backward_seed(loss))      # the cached seed the Trainer uses (no fill launch per step)
            reducer.finish()
            with markers.range("optimiser (RAdam)"):
                opt.step()
            return loss
        return step

    step = make_step(model, opt, reducer, batch)

    if args.graphed:
        from multimodal_supernovae_amd.trainer import GraphedTrainStep
        graphed = GraphedTrainStep(model, opt, warmup=3, reducer=reducer)
        args.warmup = max(args.warmup, 5)          # 3 eager steps, the capture, one replay
        args.no_alt = True
        eager_step = step

        def step():
            return graphed(batch)

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(fn, warmup, steps):
        """W untimed steps, then EXACTLY `steps` steps between two (barrier + synchronize) fences; max over ranks."""
        for _ in range(warmup):
            loss = fn()
        fence()
        # Short steps (small towers: 15 ms) are not warm after W of them -- caching allocator still growing, core clock
        # still ramping after the idle gap (1.9 -> 2.4 GHz): cfg2 timed 21 ms / step after 3 warm-up steps and 15.2 ms
        # after 10 or 30.  Keep stepping, untimed, until about 0.6 s of steps have run (same count on every rank).
        t0 = time.perf_counter()
        loss = fn()
        fence()
        one = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if world > 1:
            torch.distributed.all_reduce(one, op=torch.distributed.ReduceOp.MAX)
        extra = int(min(200, max(0, 0.6 / max(float(one), 1e-4) - warmup - 1)))
        for _ in range(extra):
            loss = fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = fn()
        fence()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t), float(loss.detach()), extra + 1

    dt, loss_value, untimed_extra = timed(step, args.warmup, args.steps)

    # ---- communication of one step: every collective issued through distributed.py, timed on the compute stream ----
    comm = {"backend": "rccl" if D.backend_name() == "nccl" else D.backend_name(), "ranks": world,
            "rccl_ranks": torch.distributed.get_world_size() if world > 1 and D.backend_name() == "nccl" else None}
    if world > 1 and not args.graphed:
        D.COMM_LOG = []
        step()
        torch.cuda.synchronize()
        log, D.COMM_LOG = D.COMM_LOG, None
        per_kind = {}
        for kind, nbytes, ev0, ev1 in log:
            k = per_kind.setdefault(kind, {"calls": 0, "bytes": 0, "ms": 0.0})
            k["calls"] += 1
            k["bytes"] += nbytes
            k["ms"] += ev0.elapsed_time(ev1) if ev0 is not None else 0.0
        comm["per_step"] = per_kind
        comm["all_gather_ms"] = sum(v["ms"] for k, v in per_kind.items() if "all_gather" in k)
        comm["all_reduce_ms"] = sum(v["ms"] for k, v in per_kind.items() if "all_reduce" in k)
        comm["note"] = ("ms = time the compute stream was held from issue to completion of the collective (HIP events "
                        "on the issuing stream); the gradient all-reduce is issued from autograd hooks under backward")

    # ---- roofline of the dominant kernel (sgemm_kernel: every dense product of both towers) -----------
    # One extra, identical step with HIP events recorded on the launch stream around every msn_sgemm call.
    # (towers one after the other for this step: with the light-curve tower on its own stream, as in the timed
    # steps, an event pair around a GEMM would also span whatever the other stream ran meanwhile)
    if args.graphed:
        step = eager_step                           # the instrumented roofline step below is an eager one
    ops.GEMM_PROFILE = []
    concurrent, model.concurrent_towers = getattr(model, "concurrent_towers", False), False
    try:    # this one step replays the side-stream towers on the caller's stream: autograd's stream-mismatch note is expected
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
    except AttributeError:
        pass
    step()                      # rehearsal of the serial order (allocator blocks, first-use code objects): not measured
    torch.cuda.synchronize()
    ops.GEMM_PROFILE = []
    ops.ATTN_PROFILE = []
    ops.FFN_PROFILE = []
    step()
    torch.cuda.synchronize()
    prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
    aprof, ops.ATTN_PROFILE = ops.ATTN_PROFILE, None
    fprof, ops.FFN_PROFILE = ops.FFN_PROFILE, None
    # the fused feed-forward launches of the narrow towers (msn_ffn_fwd / _bwd: the four Linear products of a block's feed-forward half
    # per direction, the hidden matrix on chip) are not msn_sgemm launches: reported beside the GEMM family, in the flops of the
    # products they replace
    fused_ff = None
    if fprof:
        f_ms, f_fl = sum(e[0].elapsed_time(e[1]) for e in fprof), sum(e[2] for e in fprof)
        fused_ff = {"launches_per_step": len(fprof), "ms_per_step_in_kernel": f_ms, "algorithmic_gflop_per_step": f_fl / 1e9,
                    "achieved_tflops": f_fl / (f_ms * 1e-3) / 1e12 if f_ms > 0 else 0.0,
                    "kernels": "msn::ffn_fwd_kernel / msn::ffn_bwd_kernel (+ finish): fp32-grade on v_mfma_f32_16x16x32_bf16, 6 plane products",
                    "flop_count": "4 M e hidden forward, 8 M e hidden backward (the products of the unfused block; the backward's recomputation is not counted)"}
    # the attention launches of the same step (msn_attention_fwd / _bwd; the ViT towers' one-pass backward with plane output is a
    # different entry point and is not in this list), by (heads, head width, tokens): for the reference-native three-tower
    # workload they, not the GEMMs, are where the step's time goes
    attention = None
    if aprof:
        groups = {}
        for ev0, ev1, fl, shape, kind in aprof:
            gk = f"{shape[1]} heads x {shape[4]}, {shape[2]} x {shape[3]} tokens"
            gentry = groups.setdefault(gk, {"launches_per_step": 0, "ms_per_step_in_kernel": 0.0, "algorithmic_gflop_per_step": 0.0})
            gentry["launches_per_step"] += 1
            gentry["ms_per_step_in_kernel"] += ev0.elapsed_time(ev1)
            gentry["algorithmic_gflop_per_step"] += fl / 1e9
        for gk, gentry in groups.items():
            gentry["achieved_tflops"] = gentry["algorithmic_gflop_per_step"] / max(gentry["ms_per_step_in_kernel"], 1e-9)
            wide = int(gk.split(" heads x ")[1].split(",")[0])
            long = max(int(x) for x in gk.split(", ")[1].split(" tokens")[0].split(" x ")) > 128
            gentry["kernels"] = ("msn::pattn_* (fp32-grade on v_mfma_f32_16x16x32_bf16: 6 plane products, probabilities split in registers)"
                                 if (12 < wide <= 16 and long) else
                                 "forward msn::attn_fwd_kernel (vector ALU); backward msn::pattn_bwd_fused_kernel (one pass on the bf16 planes) from 256 (sample, head) pairs on"
                                 if (wide < 16 and long) else
                                 "msn::attn_* (vector ALU)" if wide < 16 else "msn::mattn_* (v_mfma_f32_16x16x4_f32)")
        a_ms = sum(v["ms_per_step_in_kernel"] for v in groups.values())
        attention = {"ms_per_step_in_kernel": a_ms, "launches_per_step": len(aprof),
                     "flop_count": "4 T^2 head_dim per (sample, head) forward, 10 T^2 head_dim backward", "by_shape": groups}
    fp32_side = None
    if args.workload == "vit_b16_bf16_lc":
        # cfg5: the roofline object is that of the bf16-resident launches (epilogue key >= 100) against the bf16 matrix peak; the
        # fp32 products of the light-curve tower (0.5 % of the flops) are reported beside it, not folded into the fraction
        rest = [e for e in prof if e[3][5] < 100]
        prof = [e for e in prof if e[3][5] >= 100]
        r_ms, r_fl = sum(e[0].elapsed_time(e[1]) for e in rest), sum(e[2] for e in rest)
        fp32_side = {"launches_per_step": len(rest), "ms_per_step_in_kernel": r_ms, "algorithmic_gflop_per_step": r_fl / 1e9,
                     "achieved_tflops": r_fl / (r_ms * 1e-3) / 1e12 if r_ms > 0 else 0.0, "peak_tflops": 157.3}
    products = {"bf16x6": 6, "bf16x3p": 3}.get(args.gemm_precision)
    if products and any(e[3][5] >= 200 for e in prof):
        # plane path: the roofline object is that of the plane launches (epilogue key >= 200) against the bf16 matrix peak, counted
        # in EXECUTED bf16 flops (`products` MFMA products per algorithmic multiply-add); the native fp32 launches that remain
        # (light-curve tower, last ViT block, patch embedding, heads) are reported beside it
        rest = [e for e in prof if e[3][5] < 200]
        prof = [e for e in prof if e[3][5] >= 200]
        r_ms, r_fl = sum(e[0].elapsed_time(e[1]) for e in rest), sum(e[2] for e in rest)
        fp32_side = {"launches_per_step": len(rest), "ms_per_step_in_kernel": r_ms, "algorithmic_gflop_per_step": r_fl / 1e9,
                     "achieved_tflops": r_fl / (r_ms * 1e-3) / 1e12 if r_ms > 0 else 0.0, "peak_tflops": 157.3}
    else:
        products = None
    gemm_ms = sum(e[0].elapsed_time(e[1]) for e in prof)
    gemm_flops = sum(e[2] for e in prof)
    # operands + result, plus the M x N aux matrix an epilogue writes (gelu' saved by the forward) or reads (gelu' / ReLU
    # output / residual in the backward and residual-add epilogues)
    gemm_bytes = sum(e[5] if len(e) > 5 else 4.0 * (e[3][2] * e[3][4] + e[3][4] * e[3][3] + e[3][2] * e[3][3] * (2 if e[4] else 1))
                     for e in prof)
    # the same launches with EVERY element at 4 bytes (SURVEY 8d's algorithmic bytes of an fp32 path: what `traffic` is scored against;
    # the plane-format count above is what the launches must move at the least as built)
    gemm_bytes_fp32 = sum(4.0 * (e[3][2] * e[3][4] + e[3][4] * e[3][3] + e[3][2] * e[3][3] * (2 if e[4] else 1)) for e in prof)
    achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    algorithmic_tflops = achieved
    if products:
        achieved *= products        # executed bf16 MFMA flops
    # dense matrix peak of the instruction the dominant kernel issues (MI355X_MICROARCH.md): fp32 157.3 TFLOP/s;
    # bf16 2500 TFLOP/s, of which the 3-product split can deliver at most a third as algorithmic flops
    # (a plane precision on a workload without products wide enough for the plane kernels -- the reference-native towers, the
    # convolution towers -- runs every product on the native fp32 kernels: its roofline is the fp32 one)
    eff_precision = "f32" if (args.gemm_precision in ("bf16x6", "bf16x3p") and not products) else args.gemm_precision
    peak = {"f32": 157.3, "bf16x6": 2500.0, "bf16x3p": 2500.0, "bf16x3": 2500.0, "bf16": 2500.0}[eff_precision]
    kernel_name = GEMM_KERNEL_NAME[eff_precision]
    if args.workload == "vit_b16_bf16_lc" and eff_precision == "f32":
        peak = 2500.0           # the image tower of cfg5 issues bf16 MFMAs whatever the process default is
        kernel_name = ("msn::bgemm_nt_kernel + msn::bgemm_tn_kernel (bf16-resident operands, 256x256 tiles, LDS-DMA, "
                       "v_mfma_f32_16x16x32_bf16; the image tower's launches -- the light-curve tower's fp32 products: fp32_launches)")

    # ---- the shader clock the chip holds under this step's load (boxes of the pool differ by up to 7 % on the matrix-core-dense
    # kernels and the clock falls as the matrix pipe fills): a one-wave probe on a side stream stamps the shader-cycle and the 100-MHz
    # real-time counters ~one step apart while a step runs on the main stream (msn_clock_probe)
    shader_clock_ghz = None
    if world == 1:
        try:
            probe_out = torch.zeros(2, dtype=torch.int64, device=device)
            side = torch.cuda.Stream(device=device)
            step_us = max(2000, int(dt / args.steps * 1e6 * 0.8))
            torch.cuda.synchronize()
            _lib.check(_lib.lib().msn_clock_probe(probe_out.data_ptr(), step_us, side.cuda_stream), "msn_clock_probe")
            step()
            torch.cuda.synchronize()
            cyc, ticks = (int(v) for v in probe_out.tolist())
            shader_clock_ghz = cyc / ticks * 0.1 if ticks > 0 else None
        except Exception as exc:       # a measurement beside the metric: never fails the line
            print(f"# clock probe failed: {exc}", file=sys.stderr)

    # ---- per-tower split (serial order, HIP events): what each tower costs alone, and the serial step next to the
    # concurrent one, so the gain of running the towers on separate streams can be read off the JSON
    towers = None
    if headline and not args.graphed and world == 1:      # one process: the reducer's hooks are no-ops
        serial_dt, _, _ = timed(step, 1, max(3, args.steps // 4))
        towers = {"concurrent_streams_ms_per_step": dt / args.steps * 1e3,
                  "serial_ms_per_step": serial_dt / max(3, args.steps // 4) * 1e3}
        cot = torch.full((b, ENC_DIM), 1.0 / b, device=device)
        for name, fn in (("image_tower_fwd_bwd_ms", lambda: model.image_embeddings_with_projection(batch[0])),
                         ("lightcurve_tower_fwd_bwd_ms",
                          lambda: model.lightcurve_embeddings_with_projection(batch[1], batch[2], batch[3]))):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for it in range(3):
                model.zero_grad(set_to_none=True)
                if it == 1:
                    ev[0].record()
                fn().backward(cot)
            ev[1].record()
            torch.cuda.synchronize()
            towers[name] = ev[0].elapsed_time(ev[1]) / 2
        model.zero_grad(set_to_none=True)
    model.concurrent_towers = concurrent

    if rank == 0 and args.gemm_table:
        table = {}
        for e in prof:
            t = table.setdefault(e[3], [0, 0.0, 0.0])
            t[0] += 1
            t[1] += e[0].elapsed_time(e[1])
            t[2] += e[2]
        for key, (n, ms_, fl) in sorted(table.items(), key=lambda kv: -kv[1][1]):
            print(f"# gemm opA={key[0]} opB={key[1]} M={key[2]:6d} N={key[3]:5d} K={key[4]:6d} epi={key[5]} calls={n:3d} "
                  f"{ms_:8.3f} ms {fl / ms_ / 1e9:7.1f} TFLOP/s", file=sys.stderr)
    # HBM-side traffic of the same kernel comes from separate rocprofv3 --pmc passes of this command
    # (FETCH_SIZE, WRITE_SIZE; tools/run_pmc.sh -> tools/summarize_pmc.py) -- counters cannot be read from inside the
    # process, so the figure is the committed one and says which commit / date it was collected at.
    traffic, traffic_source = None, None
    if args.workload == "vit_b16_bf16_lc":
        pmc_name = "pmc_bgemm.json"
    elif headline and b == 1024:
        pmc_name = "pmc_pgemm.json" if products else "pmc_sgemm.json"
    else:
        pmc_name = f"pmc_{'pgemm' if products else 'sgemm'}_{args.workload}_b{b}.json"      # tools/run_pmc.sh --workload W --per-gpu-batch B
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
        if eff_precision == "f32" or products:       # per launch on ONE GPU: the same whatever the number of ranks (rows per GPU match)
            traffic = pmc["traffic_bytes_per_launch"]
            traffic_source = (f"profiles/{pmc_name}: rocprofv3 --pmc passes of `{pmc.get('workload', 'python bench.py')}` "
                              f"(tools/run_pmc.sh), collected {pmc.get('collected', '?')} at commit {pmc.get('commit', '?')}; "
                              "not re-measured inside this run")
    except (OSError, ValueError, KeyError):
        pass

    # ---- second measurements: the same step under the other arithmetic routes.  Reported beside the headline, never as it.
    alt = None
    if not args.no_alt and world == 1 and headline:
        alt = {}
        for name, text in (("f32", "native fp32 MFMA (v_mfma_f32_32x32x2_f32) for every product: the default of rounds 1-3"),
                           ("bf16x3p", "two resident bf16 planes, 3 bf16 MFMA products, fp32 accumulate (~1e-5 relative per product: "
                                       "NOT fp32 grade, opt-in)")):
            if name == args.gemm_precision:
                continue
            ops.set_gemm_precision(name)
            n_alt = max(3, args.steps // 2)
            dt_alt, _, _ = timed(step, 2, n_alt)
            alt[name] = {"gemm_arithmetic": text, "value": b * world * n_alt / dt_alt, "unit": "pairs/s",
                         "ms_per_step": dt_alt / n_alt * 1e3, "steps": n_alt}
        ops.set_gemm_precision(args.gemm_precision)

    # ---- second field: weak scaling at 256 rows per GPU (global 256 * N): the SAME model, optimiser and gradient buckets on
    # a batch of 256 rows per rank (throughput does not depend on the optimiser's age; building a second model + a second
    # hook-driven reducer in the same process is what the 2-rank rehearsal on one shared GPU could not do at speed)
    weak = None
    import gc
    if not args.no_weak and not args.graphed:
        batch_w = synthetic_batch(256, 1234 + rank, device)
        step_w = make_step(model, opt, reducer, batch_w)
        dt_w, loss_w, _ = timed(step_w, args.warmup, args.steps)
        weak = {"value": 256 * world * args.steps / dt_w, "unit": "pairs/s", "per_gpu_batch": 256,
                "global_batch": 256 * world, "ms_per_step": dt_w / args.steps * 1e3, "steps": args.steps,
                "scaling": "weak", "loss": loss_w}
        del step_w, batch_w
    want_three = headline and not args.no_three_tower and not args.graphed
    if want_three:
        # the headline's model, optimiser state and gradient buckets are released before the three-tower model is built
        reducer.remove()
        del step, reducer, opt, model, batch
        gc.collect()
        torch.cuda.empty_cache()

    # ---- third field (row N1 of the north star: image + light-curve + 1024-bin spectra): the three-tower workload of
    # BASELINE cfg4 (ViT-S/8 + LC transformer + 1-D CNN spectrum tower, symmetric 3-way InfoNCE) at 256 and 1024 rows per GPU
    three = None
    if want_three:
        three = {"workload": WORKLOADS["vit_s8_lc_cnn1d_sp"], "unit": "pairs/s", "scaling": "weak", "per_gpu": []}
        for rows in (256, 1024):
            model_t, batch_t = build_workload("vit_s8_lc_cnn1d_sp", rows, 1234 + rank, device)
            D.broadcast_module(model_t)
            opt_t = model_t.configure_optimizers()["optimizer"]
            red_t = D.GradientReducer(model_t.parameters())
            step_t = make_step(model_t, opt_t, red_t, batch_t)
            n_t = max(3, args.steps // 2)
            dt_t, loss_t, _ = timed(step_t, 2, n_t)
            conc_t, model_t.concurrent_towers = getattr(model_t, "concurrent_towers", False), False
            step_t()                                   # rehearsal of the serial order, then the instrumented step
            torch.cuda.synchronize()
            ops.GEMM_PROFILE = []
            step_t()
            torch.cuda.synchronize()
            prof_t, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
            model_t.concurrent_towers = conc_t
            ms_t = sum(e[0].elapsed_time(e[1]) for e in prof_t)
            fl_t = sum(e[2] for e in prof_t)
            tf_t = fl_t / (ms_t * 1e-3) / 1e12 if ms_t > 0 else 0.0
            three["per_gpu"].append({"per_gpu_batch": rows, "global_batch": rows * world, "ms_per_step": dt_t / n_t * 1e3,
                                     "value": rows * world * n_t / dt_t, "steps": n_t, "loss": loss_t,
                                     "gemm_tflops": tf_t, "gemm_tflops_is": "algorithmic (fp32-equivalent) over all GEMM launches",
                                     **_plane_side(prof_t, products),
                                     "gemm_launches_per_step": len(prof_t), "gemm_ms_per_step": ms_t})
            red_t.remove()
            del step_t, model_t, batch_t, opt_t, red_t
            gc.collect()
            torch.cuda.empty_cache()

    if rank == 0:
        ms = dt / args.steps * 1e3
        pairs = b * world * args.steps / dt
        out = {
            "metric": "contrastive pairs/sec (image+light-curve) at global batch 1024, 1/2/4/8 GPUs",
            "value": pairs, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("ViT-S/8 image tower (64x64x3, build-defined) + reference light-curve "
                                    "transformer (T=200, emb 64, 8 heads, depth 5, 2 bands) -> enc_dim 128, "
                                    "symmetric InfoNCE with all-gathered global negatives, RAdam; full train step")
                       if headline else WORKLOADS[args.workload] + " (non-headline configuration)",
                       "per_gpu_batch": b, "global_batch": b * world, "parallelism": f"dp{world}",
                       "launch": "HIP graph replay" if args.graphed else "eager",
                       "gemm_arithmetic": GEMM_ARITHMETIC[eff_precision] if args.workload != "vit_b16_bf16_lc" else
                       "image tower: operands rounded to bf16, resident in HBM (BASELINE cfg5 arithmetic); light-curve tower: native fp32 MFMA",
                       "untimed_steps_after_warmup": untimed_extra,
                       "loss": loss_value, "algorithmic_gflop_per_pair": flops_per_pair() / 1e9,
                       "executed_gflop_per_pair": flops_per_pair(executed=True) / 1e9,
                       "model_tflops": pairs * flops_per_pair(executed=True) / 1e12},
            "roofline": {"bound": "mfma", "kernel": kernel_name,
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_unit": "bytes per launch (mean), 2 x FETCH_SIZE + WRITE_SIZE",
                         "traffic_source": traffic_source,
                         "launches_per_step": len(prof), "ms_per_step_in_kernel": gemm_ms,
                         "algorithmic_gflop_per_step": gemm_flops / 1e9,
                         "algorithmic_bytes_per_launch": gemm_bytes_fp32 / max(len(prof), 1),
                         "algorithmic_bytes_is": "operands + result (+ the aux matrix an epilogue reads or writes) of the launches at 4 bytes per element",
                         "traffic_over_algorithmic": (traffic / (gemm_bytes_fp32 / max(len(prof), 1))) if traffic else None,
                         **({"algorithmic_bytes_per_launch_plane_format": gemm_bytes / max(len(prof), 1),
                             "traffic_over_plane_format_bytes": (traffic / (gemm_bytes / max(len(prof), 1))) if traffic else None}
                            if products else {}),
                         "shader_clock_ghz": shader_clock_ghz,
                         "shader_clock_is": "d s_memtime / d s_memrealtime x 100 MHz of a one-wave probe on a side stream beside one training step (nominal 2.4)",
                         **({"mfma_products_per_multiply_add": products, "algorithmic_tflops": algorithmic_tflops,
                             "note": "achieved / peak count EXECUTED bf16 MFMA flops (products x algorithmic); "
                                     "algorithmic_tflops is the fp32-equivalent rate of the same launches; algorithmic_bytes_per_launch_plane_format "
                                     "counts plane operands / plane results at their HBM format (2 bytes x planes per element), fp32 matrices at 4"} if products else {}),
                         **({"fp32_launches": fp32_side} if fp32_side is not None else {}),
                         **({"attention": attention} if attention is not None else {}),
                         **({"fused_feed_forward": fused_ff} if fused_ff is not None else {})},
            "comm": comm,
        }
        if not headline:
            for k in ("algorithmic_gflop_per_pair", "executed_gflop_per_pair", "model_tflops"):
                out["config"].pop(k)
        if weak is not None:
            out["weak_scaling_256_per_gpu"] = weak
        if three is not None:
            out["three_tower"] = three
        if towers is not None:
            out["towers"] = towers
        if alt is not None:
            out["alt_arithmetic"] = alt
            if "f32" in alt:       # box-independent figure of merit: the default arithmetic's step over the native-fp32 step of the SAME run
                out["headline_over_native_fp32"] = ms / alt["f32"]["ms_per_step"]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_batch, args.cpu_steps)
        elif world == 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
