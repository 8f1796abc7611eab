"""Fused multi-tensor Adam with the semantics of `torch.optim.Adam(params, lr, weight_decay)` as the
reference uses it (/root/reference/Main_Final.py:552,582): L2-coupled weight decay (grad += wd * p),
bias-corrected, eps added after the sqrt.  All parameter tensors are updated by ONE kernel launch
(runet_adam_multi): a device table of (param, grad, exp_avg, exp_avg_sq, numel) rows plus a chunk list.

It is a torch.optim.Optimizer (param_groups / zero_grad / state_dict / lr schedulers such as
ReduceLROnPlateau work unchanged).  Parameters may have any dense memory layout (model.py stores conv
weights HWIO): the update is elementwise over the underlying storage, and the gradient must be laid
out like its parameter (RobustUNet's backward produces exactly that).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._lib import check, lib


def _dense_like(p, g):
    return g.shape == p.shape and g.stride() == p.stride()


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}
        self.grad_scale = 1.0     # multiplied into the gradient (1/world_size after a SUM all-reduce)
        self.capturable = False   # True: hyper-parameters and the step counter are read from device memory (hipGraph replay, trainer.TrainStep)
        self._dev_state = {}      # group index -> (hyper float[6], step int32[1], host copy of hyper)
        self.skip_flag = None     # device int32[2] set by trainer.TrainStep under loss scaling: [0] != 0 -> this update is skipped on the device

    def _device_hyper(self, gi, group, dev, step_host):
        b1, b2 = group["betas"]
        want = (float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), float(self.grad_scale))
        st = self._dev_state.get(gi)
        if st is None:
            st = [torch.tensor(want, dtype=torch.float32, device=dev), torch.tensor([step_host], dtype=torch.int32, device=dev), want]
            self._dev_state[gi] = st
        elif st[2] != want:       # e.g. ReduceLROnPlateau changed lr: an ordinary stream-ordered copy, issued OUTSIDE any capture
            st[0].copy_(torch.tensor(want, dtype=torch.float32), non_blocking=False)
            st[2] = want
        return st

    def sync_hyper(self):
        """Refresh the device copy of the hyper-parameters (call before replaying a captured step)."""
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None or p in self.state]
            if plist:
                self._device_hyper(gi, group, plist[0].device, self.state[plist[0]].get("step", 0) if plist[0] in self.state else 0)

    def advance_host_step(self):
        """A graph replay advanced the device step counter: keep the host-side `state[p]["step"]` (state_dict) in line."""
        for st in self.state.values():
            if "step" in st:
                st["step"] += 1

    def _table(self, gi, plist):
        key_ptrs = tuple(t.data_ptr() for p in plist for t in (p, p.grad, self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]))
        cached = self._tables.get(gi)
        if cached is not None and cached[0] == key_ptrs:
            return cached[1], cached[2], cached[3]
        T = len(plist)
        tab = np.zeros((5, T), dtype=np.int64)
        chunk = lib.runet_adam_chunk_elems()
        chunks = []
        for i, p in enumerate(plist):
            st = self.state[p]
            tab[:, i] = (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
            chunks += [(i, c) for c in range((p.numel() + chunk - 1) // chunk)]
        dev = plist[0].device
        d_tab = torch.from_numpy(tab).to(dev)
        d_chunks = torch.tensor(chunks, dtype=torch.int32).to(dev)
        self._tables[gi] = (key_ptrs, d_tab, d_chunks, len(chunks))
        return d_tab, d_chunks, len(chunks)

    @torch.no_grad()
    def step(self, closure=None):
        ops.bump_weight_epoch()      # derived weights (Winograd-domain filters, packed bf16 copies) are stale after this step
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            for p in plist:
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam updates HIP-device parameters only")
                if not _dense_like(p, p.grad):
                    p.grad = torch.empty_like(p, memory_format=torch.preserve_format).copy_(p.grad)
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
            step = self.state[plist[0]]["step"]
            d_tab, d_chunks, n_chunks = self._table(gi, plist)
            if self.capturable:
                capturing = torch.cuda.is_current_stream_capturing()
                if capturing:     # the captured launch must not depend on host state: no hyper refresh, the replay driver counts steps
                    for p in plist:
                        self.state[p]["step"] -= 1
                    hyper, step_dev, _ = self._dev_state[gi]
                else:
                    hyper, step_dev, _ = self._device_hyper(gi, group, plist[0].device, step - 1)
                check(lib.runet_adam_multi_dev(d_tab.data_ptr(), len(plist), d_chunks.data_ptr(), n_chunks, hyper.data_ptr(), step_dev.data_ptr(),
                                               self.skip_flag.data_ptr() if self.skip_flag is not None else None,
                                               ops.stream()))
                continue
            b1, b2 = group["betas"]
            check(lib.runet_adam_multi(d_tab.data_ptr(), len(plist), d_chunks.data_ptr(), n_chunks, float(group["lr"]), float(b1), float(b2),
                                       float(group["eps"]), float(group["weight_decay"]), int(step), float(self.grad_scale),
                                       self.skip_flag.data_ptr() if self.skip_flag is not None else None,
                                       ops.stream()))
        return loss
