"""A minimal stand-in for the `pl.Trainer.fit(model, train_loader, val_loader)` call the reference
makes (script_wandb.py:229-245, pretraining_clip_wandb.py:140-156): same hook order and per-batch
sequence as Lightning's automatic optimisation --

  on_train_epoch_start -> [zero_grad -> training_step -> backward -> optimizer.step]* ->
  on_train_epoch_end -> on_validation_start -> validation_step* -> on_validation_epoch_end

-- with the 9-tuple batch moved to the GPU (None / empty placeholders pass through), epoch means
of train_loss / val_loss, and (multi-process) SUM all-reduce of the gradients before the step.
Checkpointing, early stopping and W&B logging of the reference harness are out of scope.
"""
import torch
import torch.distributed as dist

from . import distributed as D
from . import markers


def _to_device(batch, device):
    out = []
    for t in batch:
        if t is None or not torch.is_tensor(t):
            out.append(t)
        elif t.numel() == 0:
            out.append(None)   # SimulationDataset hands over torch.empty(0) for absent modalities
        else:
            out.append(t.to(device, non_blocking=True))
    return tuple(out)


def _batch_rows(batch):
    """Rows of a 9-tuple batch = leading dimension of its first tensor (Lightning's batch-size inference)."""
    for t in batch:
        if torch.is_tensor(t) and t.dim() > 0:
            return int(t.shape[0])
    return 1


def _weighted_mean(values, rows):
    """Epoch mean the way Lightning's on_epoch=True reduction forms it: every step weighted by its batch size."""
    w = torch.tensor(rows, dtype=torch.float64, device=values[0].device)
    return float((torch.stack([v.double() for v in values]) * w).sum() / w.sum())


def _check_sharded_loader(loader, group, what):
    """Data parallel: every rank must see the same number of equally sized batches per epoch -- the embedding
    all-gather assumes equal local row counts, and a rank with one batch more would wait in a collective the others
    never enter.  Checked up front (an error instead of a hang) for anything that has a length."""
    import torch.distributed as dist
    n = len(loader) if hasattr(loader, "__len__") else -1
    bs, drop_last = getattr(loader, "batch_size", None), getattr(loader, "drop_last", None)
    # rows THIS rank iterates over: its sampler's length (a DistributedSampler holds the shard), else the whole dataset
    shard = getattr(loader, "sampler", None)
    if shard is None or not hasattr(shard, "__len__"):
        shard = getattr(loader, "dataset", None)
    rows = len(shard) if shard is not None and hasattr(shard, "__len__") else -1
    last = (rows % bs or bs) if (bs and rows > 0 and not drop_last) else (bs or -1)      # rows of this rank's last batch
    world = dist.get_world_size(group)
    seen = [None] * world
    dist.all_gather_object(seen, (n, last), group=group)
    if len({c for c, _ in seen}) != 1:
        raise ValueError(f"{what}: ranks disagree on the number of batches per epoch {[c for c, _ in seen]}; shard the "
                         "dataset into equal parts (e.g. DistributedSampler(drop_last=True))")
    if len({l for _, l in seen}) != 1:
        raise ValueError(f"{what}: data-parallel training with global negatives needs equal batches on every rank -- the "
                         f"last batches of the ranks have {[l for _, l in seen]} rows; build the DataLoader with drop_last=True")


def _hook(model, name):
    """A Lightning hook is optional on the module (LightningModule supplies no-op defaults)."""
    fn = getattr(model, name, None)
    if callable(fn):
        fn()


_SEEDS = {}


def _backward_seed(loss):
    """d loss / d loss = 1 as a cached device tensor (autograd would launch a fill for it every step)."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    if key not in _SEEDS:
        _SEEDS[key] = torch.ones(loss.shape, dtype=loss.dtype, device=loss.device)
    return _SEEDS[key]


class Trainer:
    def __init__(self, max_epochs=1, device=None, group=None, log_fn=None, sync_batchnorm=False, graphed_steps=False):
        self.max_epochs = max_epochs
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.group = group
        self.log_fn = log_fn
        self.sync_batchnorm = sync_batchnorm   # pl.Trainer(sync_batchnorm=...): batch statistics over all ranks
        self.graphed_steps = graphed_steps     # replay the training step as HIP graphs (GraphedTrainStep; any world size)
        self.history = {"train_loss": [], "val_loss": []}
        self.step_losses = []
        self.global_step = 0

    def fit(self, model, train_dataloaders, val_dataloaders=None):
        model.to(self.device)
        D.broadcast_module(model, group=self.group)
        D.enable_sync_batchnorm(self.group, enabled=self.sync_batchnorm and D.world_size(self.group) > 1)
        optim_config = model.configure_optimizers()
        optimizer = optim_config["optimizer"]
        self.optimizer = optimizer
        # bucket all-reduces run under backward; a graph-replayed step reduces every bucket after backward (deferred form)
        reducer = D.GradientReducer(model.parameters(), group=self.group, overlap=not self.graphed_steps)
        world = D.world_size(self.group)
        if world > 1:
            _check_sharded_loader(train_dataloaders, self.group, "train_dataloaders")
        scheduler = self._scheduler_of(optim_config)
        graphed = GraphedTrainStep(model.train(), optimizer, reducer=reducer, group=self.group) if self.graphed_steps else None
        for epoch in range(self.max_epochs):
            model.train()
            _hook(model, "on_train_epoch_start")
            losses, rows = [], []
            for batch_idx, batch in enumerate(train_dataloaders):
                batch = _to_device(batch, self.device)
                rows.append(_batch_rows(batch))
                if graphed is not None:
                    losses.append(graphed(batch, batch_idx).detach().clone())   # the graph's loss tensor is overwritten by the next replay
                    self.global_step += 1
                    continue
                optimizer.zero_grad(set_to_none=True)
                with markers.range("forward + loss"):
                    loss = model.training_step(batch, batch_idx)
                with markers.range("backward"):
                    loss.backward(_backward_seed(loss))
                reducer.finish()
                with markers.range("optimiser (RAdam)"):
                    optimizer.step()
                losses.append(loss.detach())
                self.global_step += 1
            _hook(model, "on_train_epoch_end")
            self.step_losses += losses
            if losses:
                self.history["train_loss"].append(_weighted_mean(losses, rows))
            if scheduler is not None:
                scheduler.step()                  # Lightning steps an epoch-interval scheduler after the training epoch
            if val_dataloaders is not None:
                self._validate(model, val_dataloaders, world)
            if self.log_fn:
                self.log_fn(epoch, {k: v[-1] for k, v in self.history.items() if v})
        reducer.remove()
        return self

    @staticmethod
    def _scheduler_of(cfg):
        """`configure_optimizers` may return {"optimizer": ..., "lr_scheduler": scheduler | {"scheduler": ...}} as the
        reference's MaskedLightCurveEncoder does (src/models_pretraining.py:167-189: RAdam + StepLR per epoch)."""
        sch = cfg.get("lr_scheduler")
        return sch.get("scheduler") if isinstance(sch, dict) else sch

    def _validate(self, model, val_dataloaders, world):
        """on_validation_start -> validation_step* -> on_validation_epoch_end (ref src/models_multimodal.py:415-556).
        With several ranks each one validates ITS shard against local negatives (no collective inside the loop: a
        short last batch or an uneven shard cannot hang or mix row counts), then the batch-weighted loss sums are
        all-reduced once; the retrieval AUC a rank logs is that of its own shard."""
        model.eval()
        _hook(model, "on_validation_start")
        vlosses, vrows = [], []
        had = getattr(model, "global_negatives", None)
        if world > 1 and had is not None:
            model.global_negatives = False
        try:
            with torch.no_grad():
                for batch_idx, batch in enumerate(val_dataloaders):
                    batch = _to_device(batch, self.device)
                    vrows.append(_batch_rows(batch))
                    vlosses.append(model.validation_step(batch, batch_idx).detach())
        finally:
            if world > 1 and had is not None:
                model.global_negatives = had
        _hook(model, "on_validation_epoch_end")
        if not vlosses and world == 1:
            return
        w = torch.tensor(vrows, dtype=torch.float64, device=self.device)
        acc = torch.zeros(2, dtype=torch.float64, device=self.device)
        if vlosses:
            acc[0], acc[1] = (torch.stack([v.double() for v in vlosses]) * w).sum(), w.sum()
        if world > 1:
            D.all_reduce_sum(acc, self.group, kind="val_loss_all_reduce")
        if float(acc[1]) > 0:
            self.history["val_loss"].append(float(acc[0] / acc[1]))


class _RecordedStep:
    """A training step recorded as HIP-graph SEGMENTS separated by host-driven exchanges.  While it records it is installed
    as distributed.SEGMENTED_CAPTURE: every collective issued through distributed.py closes the segment under capture,
    is remembered (not run: see exchange) and opens the next segment.  All segments allocate from ONE private pool, so a tensor produced in one segment and consumed in a
    later one keeps its address.  replay() = segment, exchange, segment, ... in the recorded order."""

    def __init__(self, device):
        self.device = device
        self.items = []                 # ("graph", CUDAGraph) | ("call", fn)
        self.pool = torch.cuda.graph_pool_handle()
        self.cur = None

    def begin(self):
        self.cur = torch.cuda.CUDAGraph()
        # thread-local capture mode: the process group's watchdog THREAD polls the events of the exchanges that ran between the
        # segments (hipEventQuery); under the default global mode such a call from any thread while a segment is being captured
        # is an error, the watchdog throws and the process aborts -- about one recording in three with a one-rank RCCL group
        self.cur.capture_begin(pool=self.pool, capture_error_mode="thread_local")

    def exchange(self, fn):
        # The exchange is only REMEMBERED while recording -- nothing of a segment under capture executes, so there is no data to
        # exchange yet, and every rank skips the same collectives in the same order (the step is carried out by the first
        # replay).  Running it for real here put RCCL work between two captures: the process group's watchdog thread polls
        # such work with hipEventQuery while the next segment is being captured, which HIP refuses ("event last recorded in a
        # capturing stream" / "not permitted when stream is capturing") -- the watchdog throws and the process aborts, about
        # one recording in five.
        self.cur.capture_end()
        self.items.append(("graph", self.cur))
        self.items.append(("call", fn))
        self.begin()

    def end(self):
        self.cur.capture_end()
        self.items.append(("graph", self.cur))
        self.cur = None

    def abort(self):
        """A recording that failed half-way: close the segment under capture (the stream must not stay in capture mode) and
        drop every segment recorded so far."""
        if self.cur is not None:
            try:
                self.cur.capture_end()
            except Exception:          # noqa: BLE001 -- the capture is already invalid; the original error is the one to report
                pass
        self.cur = None
        self.items = []

    def replay(self):
        for kind, x in self.items:
            if kind == "graph":
                x.replay()
            else:
                x()

    @property
    def segments(self):
        return sum(1 for k, _ in self.items if k == "graph")

    @property
    def exchanges(self):
        return sum(1 for k, _ in self.items if k == "call")


class GraphedTrainStep:
    """One training step (zero_grad -> training_step -> backward -> gradient all-reduce -> RAdam) recorded as HIP graphs
    and replayed.

    The reference's own batch sizes (32 ... 256) -- and the 128 ... 256 rows a rank keeps when the global batch of 1024 is
    spread over 4 or 8 GPUs -- leave the GPU waiting for the host: a Maven step issues ~1300 launches and takes ~10 ms of
    host time whatever the batch, 1.5 ms of GPU time at batch 32.  The first `warmup` calls run eagerly (they are real
    steps: moment buffers, allocator, code objects); the next call records the step on the caller's batch shapes and every
    later call copies its batch into the recorded input tensors and replays.

    Data parallel (world size > 1): the step is recorded in SEGMENTS around its exchanges (_RecordedStep): embedding
    all-gather | InfoNCE forward | LSE all-gather | loss all-reduce | backward of the loss and the towers + gather of the
    gradient buckets | SUM all-reduce of the buckets | RAdam.  The collectives stay host-driven between two graph replays
    (gloo cannot be captured at all; RCCL needs no capture support this way), so the gradient reduction is the deferred
    form of GradientReducer (overlap=False: every bucket after backward) -- pass such a reducer or none.
    Restrictions: fixed batch shapes (others run eagerly, with the same reducer); synchronised BatchNorm issues
    collectives from inside autograd's backward threads and is refused.
    Dropout seeds are device-resident inside the graph (a base that one launch per replay advances + the ordinal of
    the call), so every replay draws new masks.  The towers' side streams fork from and join the capturing stream, so the
    graph keeps their concurrency.

        step = GraphedTrainStep(model, model.configure_optimizers()["optimizer"])
        for batch in loader: loss = step(batch)          # `loss` is a device tensor overwritten by the next call
    """

    def __init__(self, model, optimizer, warmup=3, concurrent_towers=None, reducer=None, group=None):
        self.concurrent_towers = concurrent_towers      # None: as the model is set (towers fork / join inside the graph)
        self.group = group
        self.world = D.world_size(group)
        if self.world > 1 and reducer is None:
            reducer = D.GradientReducer(model.parameters(), group=group, overlap=False)
        if reducer is not None and reducer.buckets and reducer.overlap:
            raise RuntimeError("GraphedTrainStep needs the deferred GradientReducer (overlap=False): hook-driven all-reduces "
                               "would be issued from autograd's threads in the middle of a graph segment")
        self.reducer = reducer
        self.model, self.optimizer, self.warmup = model, optimizer, int(warmup)
        self.calls, self.graph, self.static, self.loss = 0, None, None, None

    def _eager(self, batch, batch_idx=0):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.model.training_step(batch, batch_idx)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.optimizer.step()
        return loss

    def _capture(self, batch):
        model = self.model
        from . import ops
        if self.world > 1 and ops.BN_SYNC_GROUP is not None:
            raise RuntimeError("GraphedTrainStep: synchronised BatchNorm exchanges statistics from inside backward and "
                               "cannot be recorded; use per-replica statistics or the eager step")
        self.static = tuple(t.clone() if torch.is_tensor(t) else t for t in batch)
        concurrent = getattr(model, "concurrent_towers", False)
        if self.concurrent_towers is not None:
            model.concurrent_towers = bool(self.concurrent_towers)
        self.optimizer.zero_grad(set_to_none=True)
        self.optimizer.graph_prepare()        # device copies of the hyper-parameters and the step count (eager)
        import gc
        gc.collect()                          # no autograd graph of an earlier step (bound to other streams) may survive
        device = self.static_device()
        seed0 = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(device, non_blocking=False)
        ops.GRAPH_SEED = [seed0, 0]
        self._seed_base = seed0
        torch.cuda.synchronize()
        if dist.is_available() and dist.is_initialized() and dist.get_backend(self.group) == "nccl":
            import time
            # the warm-up steps' finished RCCL work must be retired by the process group's watchdog before a segment is under
            # capture (it polls with hipEventQuery): a barrier + synchronize makes the work complete, then > 3 watchdog sweeps
            # (TORCH_NCCL_HEARTBEAT... is not what paces it: the sweep period is 100 ms; MSN_WATCHDOG_QUIESCE_S overrides)
            dist.barrier(self.group)
            torch.cuda.synchronize()
            time.sleep(float(__import__("os").environ.get("MSN_WATCHDOG_QUIESCE_S", "0.35")))
        torch.cuda.empty_cache()              # (it must not be polled while a segment is under capture, see _RecordedStep.exchange)
        rec = _RecordedStep(device)
        stream = torch.cuda.Stream(device=device)
        stream.wait_stream(torch.cuda.current_stream(device))
        D.SEGMENTED_CAPTURE = rec
        failure = None
        try:
            with torch.cuda.stream(stream):
                rec.begin()
                ops.graph_seed_advance()
                self.loss = model.training_step(self.static, 0)
                self.loss.backward()
                if self.reducer is not None:
                    self.reducer.finish()
                self.optimizer.step()
                rec.end()
        except Exception as exc:   # noqa: BLE001 -- out of memory in the private pool, an op that is illegal under capture, ...
            failure = exc
            rec.abort()            # leave capture mode, drop the segments: the next call must not record on a poisoned stream
        finally:
            D.SEGMENTED_CAPTURE = None
            ops.GRAPH_SEED = None
            model.concurrent_towers = concurrent
        torch.cuda.current_stream(device).wait_stream(stream)
        if self.world > 1:
            # every rank must have a recording before any rank replays: the first replay carries out the step's collectives,
            # and a rank that failed to record would leave the others waiting in them
            ok = torch.tensor([0.0 if failure is not None else 1.0], device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.SUM, group=self.group)       # (not a step collective: kept out of COMM_LOG)
            if failure is None and float(ok) < self.world:
                rec.abort()
                failure = RuntimeError("GraphedTrainStep: another rank failed to record the step")
        if failure is not None:
            self.static, self.loss = None, None
            raise failure
        self.graph = rec

    def static_device(self):
        return next(t.device for t in self.static if torch.is_tensor(t))

    def __call__(self, batch, batch_idx=0):
        self.calls += 1
        if self.graph is None:
            if self.calls <= self.warmup:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):             # torch's rule: warm up off the stream that will capture
                    loss = self._eager(batch, batch_idx)
                torch.cuda.current_stream().wait_stream(side)
                return loss
            self._capture(batch)
        same = len(batch) == len(self.static) and all(
            (torch.is_tensor(d) and torch.is_tensor(s_) and d.shape == s_.shape) or (d is None and s_ is None)
            for d, s_ in zip(self.static, batch))
        if not same:                 # e.g. the short last batch of an epoch: one eager step, the graph stays valid
            loss = self._eager(batch, batch_idx)
            self.optimizer.graph_note_eager_step()
            return loss
        for dst, src in zip(self.static, batch):
            if torch.is_tensor(dst):
                dst.copy_(src, non_blocking=True)
        self.optimizer.graph_pre_replay()
        self.graph.replay()
        return self.loss
