"""One training step of the reference loop (/root/reference/Main_Final.py:570-584) on the HIP path:
zero_grad -> forward -> BCE -> backward (-> RCCL all-reduce) -> Adam.  No host synchronisation inside
(the reference's per-step `loss.item()` is left to the caller)."""
from __future__ import annotations

import torch

from . import ops
from ._lib import check, lib
from .optim import FusedAdam


def segmentation_loss(outputs, masks):
    """The loss the reference pairs with each model: int64 class masks [N, H, W] -> CrossEntropyLoss on logits (plain U-Net,
    train_water_segmentation.py:304); float masks [N, 1, H, W] -> BCELoss on probabilities, output resized to the mask if the sizes
    differ (Main_Final.py:577-580)."""
    if masks.dtype == torch.int64:
        return ops.cross_entropy(outputs, masks)
    return ops.bce_loss(ops.match_size(outputs, masks), masks)


def _binary_maps(outputs, masks):
    """(prediction, target) as float maps for the per-image metrics (threshold 0.5): class logits -> argmax == 1 (train_water_segmentation.py:384-388)."""
    if masks.dtype == torch.int64:
        return (outputs.argmax(dim=1) == 1).float(), (masks == 1).float()
    return ops.match_size(outputs, masks), masks


class TrainStep:
    """graph=True: after `graph_warmup` ordinary steps the whole step (zero_grad, forward, loss, backward, Adam: ~650 kernel launches)
    is captured once into a hipGraph and replayed; inputs are copied into static buffers.  Worth it when the step is launch-bound
    (2 images per GPU: 12.4 ms eager); at 16 images per GPU the GPU is the bottleneck either way.  Needs static shapes and a
    FusedAdam in capturable mode (device-side step counter and hyper-parameters, so LR schedulers keep working without re-capture);
    a `grad_sync` (RCCL gradient all-reduce on the communication stream) is captured with the step."""

    def __init__(self, model, lr=1e-4, weight_decay=1e-4, grad_sync=None, graph=False, graph_warmup=2, loss_scale=None, scale_window=2000):
        """loss_scale (fp16 operands: model.set_precision("fp16")): the loss is multiplied by it before backward so that the gradients
        the data-gradient / weight-gradient kernels round to fp16 stay above its 6e-5 normal range; the fused Adam divides it out again
        (grad_scale), a device-side check finds Inf / NaN gradients and makes the optimizer skip that step without a host round trip
        (hipGraph-capturable).  The scale is read from DEVICE memory by the step (a one-element tensor multiplied into the loss), so a
        captured step follows `adjust_loss_scale()` (one host sync; call it once per epoch or every few hundred steps), which halves the
        scale after skipped steps and doubles it after `scale_window` clean STEPS, like torch.amp.GradScaler.
        grad_sync + graph: the bucketed RCCL all-reduces are stream-ordered work on the communication stream, which forks from and
        joins the capturing stream through events - they are captured with the step (BASELINE.json configs[4])."""
        self.model = model
        self.optimizer = FusedAdam(model.parameters(), lr=lr, weight_decay=weight_decay)
        self.grad_sync = grad_sync
        if grad_sync is not None:
            grad_sync.attach(self.optimizer)
        self.loss_scale = float(loss_scale) if loss_scale else None
        self.scale_window, self._clean_steps, self._seen_skips = int(scale_window), 0, 0
        self._steps, self._steps_at_adjust = 0, 0           # train steps issued (host count) / at the last adjust_loss_scale()
        self._scale_dev = None
        if self.loss_scale is not None:
            self.optimizer.capturable = True          # step counter on the device: a skipped step must not advance the bias correction
            self._base_grad_scale = self.optimizer.grad_scale
            self.optimizer.grad_scale = self._base_grad_scale / self.loss_scale
            self._flag = None
        self.graph_mode = bool(graph)
        if self.graph_mode:
            self.optimizer.capturable = True
            if hasattr(model, "grad_arena"):
                model.grad_arena()        # gradients live at fixed addresses (views of one arena): the Adam pointer table stays valid
            # (DeepLabV3+ / the plain U-Net: ops.deliver_grads keeps per-parameter gradient buffers for the life of the model)
        self._eager_left = int(graph_warmup)
        self._graph = None
        self._static = None

    def _scale_tensor(self, device):
        """The loss scale as a device scalar (created / refreshed OUTSIDE any capture: a replayed graph reads the current value)."""
        if self._scale_dev is None:
            self._scale_dev = torch.tensor(self.loss_scale, device=device, dtype=torch.float32)
            self._scale_host = self.loss_scale
        elif self._scale_host != self.loss_scale:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the loss scale changed during a capture")
            self._scale_dev.fill_(self.loss_scale)
            self._scale_host = self.loss_scale
        return self._scale_dev

    def _body(self, images, masks):
        self.optimizer.zero_grad(set_to_none=True)
        loss = segmentation_loss(self.model(images), masks)
        if self.loss_scale is None:
            loss.backward()
        else:
            (loss * self._scale_tensor(loss.device)).backward()
        if self.grad_sync is not None:
            self.grad_sync.finish()
        if self.loss_scale is not None:
            flat = self.model.grad_arena().flat if hasattr(self.model, "grad_arena") else None
            if flat is None:
                raise TypeError("loss scaling needs a model with a flat gradient arena (RobustUNet)")
            if self._flag is None:
                self._flag = torch.zeros(2, device=flat.device, dtype=torch.int32)
                self.optimizer.skip_flag = self._flag
            check(lib.runet_nonfinite_flag(flat.data_ptr(), flat.numel(), self._flag.data_ptr(), ops.stream()))
        self.optimizer.step()
        return loss

    def set_loss_scale(self, scale):
        """Set the loss scale by hand (also under a captured step: the device copies are refreshed before the next replay)."""
        if self.loss_scale is None:
            raise RuntimeError("this TrainStep was built without loss scaling")
        self.loss_scale = float(scale)
        self.optimizer.grad_scale = self._base_grad_scale / self.loss_scale

    def adjust_loss_scale(self):
        """Dynamic loss scale (one host sync): -> (current scale, steps skipped so far).  Clean steps are counted per TRAIN STEP (host
        count of issued steps minus the device's skip counter), not per call; the host-side Adam step (state_dict) is corrected by the
        skipped steps, which the device-side counter never advanced."""
        if self.loss_scale is None or self._flag is None:
            return self.loss_scale, 0
        skipped = int(self._flag[1].item())
        new_skips = skipped - self._seen_skips
        issued = self._steps - self._steps_at_adjust
        self._steps_at_adjust = self._steps
        if new_skips > 0:
            self.loss_scale = max(1.0, self.loss_scale / 2.0 ** min(4, new_skips))
            self._clean_steps = 0
            for st in self.optimizer.state.values():
                if "step" in st:
                    st["step"] = max(0, st["step"] - new_skips)
        else:
            self._clean_steps += max(issued, 0)
            if self._clean_steps >= self.scale_window:
                self.loss_scale, self._clean_steps = self.loss_scale * 2.0, 0
        self._seen_skips = skipped
        self.optimizer.grad_scale = self._base_grad_scale / self.loss_scale
        if self._scale_dev is not None:
            self._scale_tensor(self._scale_dev.device)          # device copy of the scale: the next (replayed) step reads it
        if self.optimizer.capturable:
            self.optimizer.sync_hyper()
        return self.loss_scale, skipped

    def __call__(self, images, masks):
        self._steps += 1
        if not self.graph_mode:
            return self._body(images, masks)
        if self._graph is not None and (images.shape != self._static[0].shape or masks.shape != self._static[1].shape):
            return self._body(images, masks)          # ragged last batch: an ordinary step
        if self._eager_left > 0:
            self._eager_left -= 1
            return self._body(images, masks)
        if self._graph is None:
            self._static = [images.clone(), masks.clone(), None]
            self.optimizer.sync_hyper()
            if self.loss_scale is not None:
                self._scale_tensor(images.device)
                if self._flag is None:        # graph_warmup=0: the overflow flag must exist before the capture (it persists across replays)
                    self._flag = torch.zeros(2, device=images.device, dtype=torch.int32)
                    self.optimizer.skip_flag = self._flag
            torch.cuda.synchronize()
            ops.PIN_SCRATCH = True        # the graph bakes in scratch addresses: pools may grow later but never free what it uses
            g = torch.cuda.CUDAGraph()
            # under a process group the collective library's watchdog thread polls events while this thread captures: only THIS
            # thread's calls are checked against the capture then ("thread_local"), as torch documents for NCCL + graphs
            mode = "thread_local" if self.grad_sync is not None else "global"
            with torch.cuda.graph(g, capture_error_mode=mode):
                self._static[2] = self._body(self._static[0], self._static[1])
            self._graph = g
        else:
            self._static[0].copy_(images, non_blocking=True)
            self._static[1].copy_(masks, non_blocking=True)
        self.optimizer.sync_hyper()
        if self.loss_scale is not None:
            self._scale_tensor(images.device)
        self._graph.replay()
        self.optimizer.advance_host_step()
        ops.bump_weight_epoch()          # the replay updated the weights: cached derived weights (eager fallback steps) are stale
        return self._static[2].detach().clone()


def fit(model, train_loader, val_loader, device, epochs=200, lr=1e-4, weight_decay=1e-4, save_dir="./models", lr_patience=10,
        stop_patience=20, grad_sync=None, log=print):
    """Checkpointing / early-stopping loop with the behaviour of the reference's older trainer
    (/root/reference/train_water_segmentation.py:514-645) around the Robust U-Net step:
    ReduceLROnPlateau(factor 0.5, patience `lr_patience`) stepped on the VALIDATION loss, the state_dict of the best
    validation IoU saved to `<save_dir>/best_water_segmentation_model.pth`, early stop after `stop_patience` epochs without an
    IoU improvement.  Under a `grad_sync` (one process per GPU) every rank takes the same decisions: ragged batches are dropped (all
    ranks run the same number of equal-sized steps), the BatchNorm running statistics of rank 0 are broadcast before each
    validation pass, validation loss / IoU / accuracy are averaged over the ranks before the scheduler and the early-stop test see
    them, and only rank 0 writes the checkpoint and the history.  The checkpoint holds plain contiguous OIHW tensors under the reference's keys, so the reference's
    `RobustUNet.load_state_dict(torch.load(path))` accepts it.  History keys follow the reference (`train_losses`,
    `val_losses`, `accuracies`, `iou_scores`, `learning_rates`, `best_model_epoch`, `training_time`); it is written as JSON
    (the reference pickles it).  Validation IoU / accuracy are the mean of Main_Final.py's per-image metrics."""
    import json
    import os
    import time

    import numpy as np
    import torch
    import torch.distributed as dist

    from .data import DevicePrefetcher
    from .evaluator import ModelEvaluator

    os.makedirs(save_dir, exist_ok=True)
    step = TrainStep(model, lr=lr, weight_decay=weight_decay, grad_sync=grad_sync)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(step.optimizer, mode="min", factor=0.5, patience=lr_patience)
    ev = ModelEvaluator(device)
    hist = {"train_losses": [], "val_losses": [], "accuracies": [], "iou_scores": [], "learning_rates": [], "best_model_epoch": 0,
            "training_time": 0.0}
    best_iou, waited, t0 = 0.0, 0, time.time()
    ckpt = os.path.join(save_dir, "best_water_segmentation_model.pth")
    multi = grad_sync is not None and dist.is_initialized() and grad_sync.world > 1
    writer = not multi or dist.get_rank(grad_sync.group) == 0
    full_batch = None
    for epoch in range(epochs):
        model.train()
        losses = []
        for images, masks in DevicePrefetcher(train_loader, device):
            if full_batch is None:
                full_batch = images.shape[0]
            if multi and images.shape[0] != full_batch:
                continue                                       # drop_last: a ragged step would desynchronise the ranks' collectives
            losses.append(step(images, masks))                 # device scalars: one host sync per epoch, not per step
        train_loss = float(torch.stack(losses).mean().item())
        if multi:
            grad_sync.broadcast_buffers(0)
        model.eval()
        vloss, mets = [], []
        with torch.no_grad():
            for images, masks in DevicePrefetcher(val_loader, device):
                outputs = model(images)
                vloss.append(segmentation_loss(outputs, masks))
                mets += ev.batch_metrics(*_binary_maps(outputs, masks))
        val_loss = float(torch.stack(vloss).mean().item())
        iou, acc = float(np.mean([m["iou"] for m in mets])), float(np.mean([m["accuracy"] for m in mets]))
        if multi:       # every rank must see the same numbers: the LR schedule and the early stop are collective decisions
            agg = torch.tensor([train_loss, val_loss, iou, acc], dtype=torch.float64, device=device if dist.get_backend(grad_sync.group) == "nccl" else "cpu")
            dist.all_reduce(agg, op=dist.ReduceOp.SUM, group=grad_sync.group)
            train_loss, val_loss, iou, acc = (agg / grad_sync.world).tolist()
        sched.step(val_loss)
        hist["train_losses"].append(train_loss)
        hist["val_losses"].append(val_loss)
        hist["accuracies"].append(acc)
        hist["iou_scores"].append(iou)
        hist["learning_rates"].append(step.optimizer.param_groups[0]["lr"])
        if iou > best_iou:
            best_iou, waited = iou, 0
            hist["best_model_epoch"] = epoch
            if writer:
                torch.save({k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}, ckpt)
        else:
            waited += 1
        log(f"epoch {epoch + 1}/{epochs}: train {train_loss:.4f} val {val_loss:.4f} IoU {iou:.4f} acc {acc:.4f} "
            f"lr {hist['learning_rates'][-1]:.2e} best IoU {best_iou:.4f} (epoch {hist['best_model_epoch'] + 1})")
        if waited >= stop_patience:
            log(f"early stop: {stop_patience} epochs without IoU improvement")
            break
    hist["training_time"] = time.time() - t0
    if writer:
        with open(os.path.join(save_dir, "training_history.json"), "w") as f:
            json.dump(hist, f)
    return hist
