"""Data-parallel training step end to end on the GPU: two ranks (sharing the one GPU of the test box, gloo
transport with CUDA tensors -- the production launch uses RCCL through the same code) run the HIP path with
all-gathered global negatives and SUM-all-reduced gradients; loss and every parameter gradient must equal the
single-process step at the doubled batch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_step_equals_single_process_global_batch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dist_check.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "DIST CHECK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_four_rank_step_equals_single_process_global_batch():
    """The same with FOUR ranks sharing the GPU: the row offsets of ranks 2 and 3 in the sharded InfoNCE kernels and in the packed
    all-gather (the 4- and 8-GPU points of the metric use them; two ranks only ever exercise offsets 0 and b)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dist_check.py"), "--world", "4"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "DIST CHECK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_two_rank_synchronised_batchnorm_equals_single_process_global_batch():
    """ConvMixer tower (three BatchNorms per layer) under data parallel with `enable_sync_batchnorm`: loss, gradients
    and running statistics of two ranks equal the single-process step at the doubled batch."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dist_check.py"), "--batchnorm"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "DIST CHECK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_rccl_carries_the_exchange_on_one_gpu():
    """RCCL ("nccl" backend) itself, through the product's exchange path: a one-rank group with the sharded loss and
    the gradient reducer forced on (tools/rccl_smoke.py) -- three exchange collectives for three modality pairs plus
    the bucketed gradient all-reduces, results equal to the plain single-process step."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_smoke.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "RCCL SMOKE OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the parent starts two ranks (sharing this box's
    one GPU over gloo; RCCL on a multi-GPU node), relays rank 0's JSON line, and the line carries the comm fields:
    one packed embedding all-gather, one LSE all-gather, one loss all-reduce per step."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--global-batch", "64"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["per_gpu_batch"] == 32
    assert out["config"]["global_batch"] == 64 and out["value"] > 0
    comm = out["comm"]
    assert comm["ranks"] == 2 and comm["backend"] in ("gloo", "rccl")
    per = comm["per_step"]
    assert per["embedding_all_gather"]["calls"] == 1 and per["embedding_all_gather"]["bytes"] == 64 * 2 * 128 * 4
    assert per["lse_all_gather"]["calls"] == 1 and per["loss_all_reduce"]["calls"] == 1
    assert per["grad_all_reduce"]["calls"] >= 1
    weak = out["weak_scaling_256_per_gpu"]
    assert weak["global_batch"] == 512 and weak["value"] > 0


def test_trainer_validation_with_uneven_shards_two_ranks():
    """`Trainer.fit(model, train, val)` on two ranks whose VALIDATION shards differ in length and batch sizes: no hang, both
    ranks report the batch-weighted global val_loss; an uneven TRAINING shard is refused with an error instead of a hang."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dist_check.py"), "--trainer"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "DIST CHECK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_graph_replayed_data_parallel_step_equals_eager_steps_bit_for_bit():
    """trainer.GraphedTrainStep with two ranks (segmented capture: graph segments between the host-driven exchanges): seven
    steps incl. two eager warm-up steps, the recording, replays and one short batch == the same steps issued eagerly."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dist_check.py"), "--graphed"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "DIST CHECK OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
