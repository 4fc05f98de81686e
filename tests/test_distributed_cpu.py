"""world_size = 2 on CPU (gloo): the all-gather / row-offset / sum-over-ranks algebra of the
global-negatives loss and of the gradient all-reduce, with the ORACLE injected as the compute
backend (the product kernels need a GPU).  Checks against the single-process oracle at batch R*b:
loss equal on every rank, dE of the local rows equal to the matching rows of the full gradient,
scale gradient and a toy encoder's parameter gradient equal after the SUM all-reduce."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_mod, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from multimodal_supernovae_amd import distributed as D
    from multimodal_supernovae_amd.loss import clip_loss_multimodal
    from oracle.sharded import OraclePairKernels
    from oracle import loss as oloss
    torch.set_num_threads(1)
    r, _, w = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and D.world_size() == world
    b, d_in, d = 6, 5, 8
    g = torch.Generator().manual_seed(7)
    x_all = [torch.randn(world * b, d_in, generator=g) for _ in range(n_mod)]
    w_all = [torch.randn(d, d_in, generator=g) for _ in range(n_mod)]
    ls0, lb0 = torch.tensor(1.7), torch.tensor(-0.4)

    def encode(xs, ws):
        es = [x @ wt.T for x, wt in zip(xs, ws)]
        return [e / e.norm(dim=-1, keepdim=True) for e in es]

    # single-process oracle at the global batch
    wr = [t.clone().requires_grad_() for t in w_all]
    lsr, lbr = ls0.clone().requires_grad_(), lb0.clone().requires_grad_()
    er = encode(x_all, wr)
    for e in er:
        e.retain_grad()
    ref = oloss.clip_loss_multimodal(er, lsr, lbr)
    ref.backward()

    # this rank: local rows only, weights replicated (broadcast from rank 0 after a deliberate perturbation)
    lin = torch.nn.ModuleList([torch.nn.Linear(d_in, d, bias=False) for _ in range(n_mod)])
    with torch.no_grad():
        for m, wt in zip(lin, w_all):
            m.weight.copy_(wt + (0.5 if rank else 0.0))
    D.broadcast_module(lin)
    ls, lb = ls0.clone().requires_grad_(), lb0.clone().requires_grad_()
    sl = slice(rank * b, (rank + 1) * b)
    el = encode([x[sl] for x in x_all], [m.weight for m in lin])
    for e in el:
        e.retain_grad()
    D.COMM_LOG = []
    loss = clip_loss_multimodal(el, ls, lb, kernels=OraclePairKernels, global_negatives=True)
    loss.backward()
    # the fused exchange: ONE packed embedding all-gather, ONE packed LSE all-gather, ONE loss all-reduce per step,
    # whatever the number of modality pairs (1 pair at n_mod = 2, 3 pairs at n_mod = 3); nothing in backward
    kinds = [e[0] for e in D.COMM_LOG]
    sizes = [e[1] for e in D.COMM_LOG]
    D.COMM_LOG = None
    n_pairs = n_mod * (n_mod - 1) // 2
    comm_ok = kinds == ["embedding_all_gather", "lse_all_gather", "loss_all_reduce"] and \
        sizes == [world * b * n_mod * d * 4, world * 2 * n_pairs * b * 4, 4]
    params = [m.weight for m in lin] + [ls, lb]
    D.allreduce_gradients(params)

    ok = comm_ok and torch.allclose(loss.detach(), ref.detach(), rtol=1e-5, atol=1e-6)
    for e_loc, e_ref in zip(el, er):
        ok = ok and torch.allclose(e_loc.grad, e_ref.grad[sl], rtol=1e-4, atol=1e-6)
    for m, wref in zip(lin, wr):
        ok = ok and torch.allclose(m.weight.grad, wref.grad, rtol=1e-4, atol=1e-6)
    ok = ok and torch.allclose(ls.grad, lsr.grad, rtol=1e-4, atol=1e-6)
    ok = ok and abs(float(lb.grad)) < 1e-5
    # local-negatives mode must NOT communicate: equals the oracle on the local rows alone
    loc = clip_loss_multimodal([e.detach() for e in el], ls0, lb0, kernels=OraclePairKernels, global_negatives=False)
    ok = ok and torch.allclose(loc, oloss.clip_loss_multimodal([e.detach() for e in el], ls0, lb0), rtol=1e-5, atol=1e-6)
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_mod,world", [(2, 2), (3, 2), (3, 4), (2, 8)])
def test_global_negatives_two_ranks_gloo(n_mod, world):
    """world 2, and -- the row offsets of ranks beyond the second, the layouts of the 4- and 8-GPU points of the metric -- 4 and 8."""
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_mod, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert dict(out) == {r: True for r in range(world)}


def _reducer_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from multimodal_supernovae_amd import distributed as D
    torch.set_num_threads(1)
    D.init_from_env(backend="gloo")
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(7, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                              torch.nn.Linear(16, 4))
    unused = torch.nn.Parameter(torch.ones(3))                 # never reached by backward: contributes zeros
    params = list(net.parameters()) + [unused]
    D.broadcast_module(net)
    import copy
    ref_net = copy.deepcopy(net)                               # hook-free twin for the single-process gradients
    g = torch.Generator().manual_seed(11)
    x_all = torch.randn(world * 5, 7, generator=g)
    ok = True
    reducer = D.GradientReducer(params, bucket_bytes=300)      # several small buckets
    ok = ok and len(reducer.buckets) >= 3
    for step in range(3):                                      # the flat buffers are reused step after step
        for p in params:
            p.grad = None
        (net(x_all[rank * 5:(rank + 1) * 5] * (step + 1)).square().sum()).backward()
        reducer.finish()
        got = [p.grad.clone() for p in params]
        ref_net.zero_grad(set_to_none=True)
        (ref_net(x_all * (step + 1)).square().sum()).backward()    # single-process gradient over the global batch
        for gp, p in zip(got[:-1], ref_net.parameters()):
            ok = ok and torch.allclose(gp, p.grad, rtol=1e-4, atol=1e-5)
        ok = ok and bool((got[-1] == 0).all())
    # the after-backward form gives the same sums
    for p in params:
        p.grad = None
    reducer.remove()
    (net(x_all[rank * 5:(rank + 1) * 5]).square().sum()).backward()
    D.allreduce_gradients(params, bucket_bytes=300)
    ref = [p.grad.clone() for p in params]
    ref_net.zero_grad(set_to_none=True)
    (ref_net(x_all).square().sum()).backward()
    for gp, p in zip(ref[:-1], ref_net.parameters()):
        ok = ok and torch.allclose(gp, p.grad, rtol=1e-4, atol=1e-5)
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gradient_reducer_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert dict(out) == {0: True, 1: True}


def _shard_worker(rank, world, port, out):
    """Deferred GradientReducer (the form a graph-replayed step uses) == the hook-driven one, and the sharded-loader check:
    a DistributedSampler shard whose last batch is short but EQUAL on every rank is accepted (1000 samples, 2 ranks,
    batches of 300: 300 + 200 rows each), unequal last batches and unequal batch counts are refused up front."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from multimodal_supernovae_amd import distributed as D
    from multimodal_supernovae_amd.trainer import _check_sharded_loader
    from torch.utils.data import DataLoader, TensorDataset
    from torch.utils.data.distributed import DistributedSampler
    torch.set_num_threads(1)
    D.init_from_env(backend="gloo")
    ok = True
    ds = TensorDataset(torch.arange(1000.0))
    loader = DataLoader(ds, batch_size=300, sampler=DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=False))
    _check_sharded_loader(loader, None, "even shards")                       # must not raise
    uneven = TensorDataset(torch.arange(10.0 if rank == 0 else 12.0))       # same batch count (3), last batches 2 / 4 rows
    try:
        _check_sharded_loader(DataLoader(uneven, batch_size=4), None, "uneven last batch")
        ok = False
    except ValueError:
        pass
    counts = TensorDataset(torch.arange(8.0 if rank == 0 else 16.0))
    try:
        _check_sharded_loader(DataLoader(counts, batch_size=4), None, "uneven counts")
        ok = False
    except ValueError:
        pass
    # deferred reducer: same sums as the hook-driven one, .grad re-pointed at the flat buffer, one exchange for all buckets
    torch.manual_seed(5)
    net = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.Tanh(), torch.nn.Linear(9, 3))
    D.broadcast_module(net)
    import copy
    twin = copy.deepcopy(net)
    x = torch.randn(world * 4, 6, generator=torch.Generator().manual_seed(2))
    hooked = D.GradientReducer(net.parameters(), bucket_bytes=100)
    deferred = D.GradientReducer(twin.parameters(), bucket_bytes=100, overlap=False)
    calls = []
    D.SEGMENTED_CAPTURE = type("Spy", (), {"exchange": staticmethod(lambda fn: (calls.append(1), fn()))})()
    try:
        for step in range(2):
            for m, red in ((net, hooked), (twin, deferred)):
                m.zero_grad(set_to_none=True)
                m(x[rank * 4:(rank + 1) * 4] * (step + 1)).square().sum().backward()
                red.finish()
            for p, q in zip(net.parameters(), twin.parameters()):
                ok = ok and torch.equal(p.grad, q.grad)
    finally:
        D.SEGMENTED_CAPTURE = None
    ok = ok and len(deferred.buckets) >= 2 and len(calls) == 2               # ONE exchange per deferred finish()
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_loader_check_and_deferred_reducer_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert dict(out) == {0: True, 1: True}
