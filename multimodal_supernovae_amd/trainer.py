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

from . import distributed as D


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


class Trainer:
    def __init__(self, max_epochs=1, device=None, group=None, log_fn=None, sync_batchnorm=False):
        self.max_epochs = max_epochs
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.group = group
        self.log_fn = log_fn
        self.sync_batchnorm = sync_batchnorm   # pl.Trainer(sync_batchnorm=...): batch statistics over all ranks
        self.history = {"train_loss": [], "val_loss": []}
        self.step_losses = []
        self.global_step = 0

    def fit(self, model, train_dataloaders, val_dataloaders=None):
        model.to(self.device)
        D.broadcast_module(model, group=self.group)
        D.enable_sync_batchnorm(self.group, enabled=self.sync_batchnorm and D.world_size(self.group) > 1)
        optimizer = model.configure_optimizers()["optimizer"]
        self.optimizer = optimizer
        reducer = D.GradientReducer(model.parameters(), group=self.group)   # bucket all-reduces run under backward
        for epoch in range(self.max_epochs):
            model.train()
            model.on_train_epoch_start()
            losses = []
            for batch_idx, batch in enumerate(train_dataloaders):
                batch = _to_device(batch, self.device)
                optimizer.zero_grad(set_to_none=True)
                loss = model.training_step(batch, batch_idx)
                loss.backward()
                reducer.finish()
                optimizer.step()
                losses.append(loss.detach())
                self.global_step += 1
            model.on_train_epoch_end()
            self.step_losses += losses
            if losses:
                self.history["train_loss"].append(float(torch.stack(losses).mean()))
            if val_dataloaders is not None:
                model.eval()
                model.on_validation_start()
                vlosses = []
                with torch.no_grad():
                    for batch_idx, batch in enumerate(val_dataloaders):
                        vlosses.append(model.validation_step(_to_device(batch, self.device), batch_idx).detach())
                model.on_validation_epoch_end()
                if vlosses:
                    self.history["val_loss"].append(float(torch.stack(vlosses).mean()))
            if self.log_fn:
                self.log_fn(epoch, {k: v[-1] for k, v in self.history.items() if v})
        reducer.remove()
        return self
