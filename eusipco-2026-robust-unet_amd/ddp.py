"""Minibatch data parallelism for the Robust U-Net path: one process per GPU, RCCL over xGMI through
`torch.distributed` (backend "nccl" IS RCCL on ROCm; "gloo" on CPU for tests).

The reference is single-process; this is the new capability the
north_star asks for.  Design for MI355X:

* gradients live in ONE flat fp32 arena laid out in backward-completion order (model.GradArena), so a
  bucket is a contiguous slice - no gather/scatter copies, and few, large collectives (xGMI links are
  point-to-point, ring all-reduce is per-link bound: prefer 30-90 MB buckets over many small ones);
* the network's backward reports "everything up to offset e is final" after each block; when at least
  `bucket_floats` new floats are final the slice is all-reduced (SUM) on a dedicated HIP stream that waits
  on an event recorded on the compute stream - the remaining backward (the high-resolution encoder blocks,
  ~26 % of backward FLOPs) overlaps it;
* `finish()` makes the compute stream wait for the communication stream before the optimizer step; the
  1/world_size averaging is folded into the fused Adam kernel (`FusedAdam.grad_scale`).

BatchNorm uses per-rank batch statistics by default (what torch's DistributedDataParallel does); `sync_bn=True`
installs `SyncBatchNorm`, which makes the N-rank step equal to the single-process step on the concatenated batch.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def _ops():
    from . import ops
    return ops


class GradAllReducer:
    def __init__(self, model, bucket_floats=8 << 20, process_group=None, average_in_optimizer=True, sync_bn=False):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_floats = int(bucket_floats)
        self.average_in_optimizer = average_in_optimizer
        self.arena = None
        self.comm_stream = None
        self._sent = 0
        self._work = []
        self.buckets_last_step = []
        self._sync_bn = SyncBatchNorm(process_group) if sync_bn else None
        if sync_bn:
            model.sync_bn_hook = self._sync_bn

    def set_sync_bn(self, on):
        """Switch between cross-rank (SyncBatchNorm) and per-rank BatchNorm statistics."""
        if on and self._sync_bn is None:
            self._sync_bn = SyncBatchNorm(self.group)
        self.model.sync_bn_hook = self._sync_bn if on else None

    # -- wiring ------------------------------------------------------------------------------
    def attach(self, optimizer=None):
        """Hook the model's gradient arena; fold 1/world into the optimizer when it supports it."""
        if not hasattr(self.model, "grad_arena"):
            raise TypeError("GradAllReducer needs a model that exposes grad_arena() (RobustUNet); the DeepLabV3+ baseline is single-process")
        self.arena = self.model.grad_arena()
        self.arena.on_block_done = self._on_block_done
        if self.arena.flat.is_cuda and self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(device=self.arena.flat.device)
        if optimizer is not None and self.average_in_optimizer:
            if not hasattr(optimizer, "grad_scale"):
                raise TypeError("optimizer has no grad_scale; construct GradAllReducer(average_in_optimizer=False)")
            optimizer.grad_scale = 1.0 / self.world
        return self

    def broadcast_parameters(self, src=0):
        """Make every rank start from rank `src`'s parameters and buffers: ONE message per dtype (fp32 parameters + BatchNorm running
        statistics packed into a flat staging buffer, the int64 `num_batches_tracked` counters into another) instead of 290 small ones."""
        self._broadcast(list(self.model.parameters()) + list(self.model.buffers()), src)

    def broadcast_buffers(self, src=0):
        """BatchNorm running statistics of rank `src` to every rank (per-rank statistics drift apart; torch DDP's broadcast_buffers)."""
        self._broadcast(list(self.model.buffers()), src)

    def _broadcast(self, tensors, src):
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault(t.dtype, []).append(t)
        with torch.no_grad():
            for dtype, ts in by_dtype.items():
                views = [_dense(t) for t in ts]
                flat = torch.cat(views)
                dist.broadcast(flat, src, group=self.group)
                off = 0
                for v in views:
                    v.copy_(flat[off:off + v.numel()])
                    off += v.numel()
        _ops().bump_weight_epoch()       # written through .data views: the parameters' version counters did not move

    # -- per-step ----------------------------------------------------------------------------
    def _launch(self, lo, hi):
        if hi <= lo:
            return
        sl = self.arena.flat[lo:hi]
        self.buckets_last_step.append((lo, hi))
        if sl.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            side = _ops().side_stream()            # weight gradients of the slice may still be running on the side stream
            if side is not None:
                self.comm_stream.wait_stream(side)
            with torch.cuda.stream(self.comm_stream):
                dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group)
                if not self.average_in_optimizer:
                    sl.mul_(1.0 / self.world)
        else:
            self._work.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _on_block_done(self, end):
        if self._sent == 0:
            self.buckets_last_step = []
        if end - self._sent >= self.bucket_floats:
            self._launch(self._sent, end)
            self._sent = end

    def finish(self):
        """Call after backward, before optimizer.step(): flush the tail bucket and join the streams."""
        if self.arena is None:
            raise RuntimeError("call attach() first")
        self._launch(self._sent, self.arena.total)
        self._sent = 0
        if self.arena.flat.is_cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for w in self._work:
                w.wait()
            self._work = []
            if not self.average_in_optimizer:
                self.arena.flat.mul_(1.0 / self.world)


class SyncBatchNorm:
    """Cross-rank batch statistics for every BatchNorm of the path (per-rank BN at 2 images/GPU is a
    different function from the reference's global-batch BN).  Forward: the per-image (mean, M2) rows of all ranks are
    all-gathered and Chan-combined by `runet_bn_finalize` exactly as a single process would combine its own images.
    Backward: the two per-channel sums that enter dx are all-reduced (the parameter gradients stay local sums and are
    averaged with everything else by the gradient all-reduce).  Messages are [2*N*C] / [2*C] floats: latency-bound, so BatchNorms that
    reach their statistics at the same point of the pass share one message (blocks.bn_coeff_pair; the paired backward sums)."""

    def __init__(self, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.messages = 0                      # collectives issued so far (tests / tuning)

    def gather_stats_many(self, items):
        """items: [(mean_nc, m2_nc, n, c), ...] of BatchNorms whose inputs are ready at the same point -> [(mean_all, m2_all, n * world), ...]
        from ONE all-gather.  Every rank must hold the same number of images `n` (equal slots): use drop_last / equal shards, as
        trainer.fit does under a GradAllReducer."""
        local = torch.cat([t.reshape(-1) for it in items for t in it[:2]])
        flat = torch.empty(self.world * local.numel(), device=local.device, dtype=local.dtype)
        dist.all_gather_into_tensor(flat, local, group=self.group)
        self.messages += 1
        rows = flat.view(self.world, local.numel())
        out, off = [], 0
        for _, _, n, c in items:
            k = n * c
            out.append((rows[:, off:off + k].reshape(-1), rows[:, off + k:off + 2 * k].reshape(-1), n * self.world))
            off += 2 * k
        return out

    def gather_stats(self, mean_nc, m2_nc, n, c):
        return self.gather_stats_many([(mean_nc, m2_nc, n, c)])[0]

    def reduce_sums_many(self, sums_list, local_counts):
        """The (dgamma | dbeta) sums of several BatchNorm backward passes in ONE all-reduce -> [(global sums, global element count), ...]"""
        g = torch.cat([t.reshape(-1) for t in sums_list])
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
        self.messages += 1
        out, off = [], 0
        for t, cnt in zip(sums_list, local_counts):
            out.append((g[off:off + t.numel()], cnt * self.world))
            off += t.numel()
        return out

    def reduce_sums(self, sums, local_count):
        return self.reduce_sums_many([sums], [local_count])[0]


def _dense(t):
    """Storage-order flat view of a dense (possibly permuted: HWIO-stored conv weights) tensor for collectives."""
    return torch.as_strided(t.data, (t.numel(),), (1,), t.storage_offset())
