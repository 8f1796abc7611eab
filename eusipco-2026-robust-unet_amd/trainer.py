"""One training step of the reference loop (/root/reference/Main_Final.py:570-584) on the HIP path:
zero_grad -> forward -> BCE -> backward (-> RCCL all-reduce) -> Adam.  No host synchronisation inside
(the reference's per-step `loss.item()` is left to the caller)."""
from __future__ import annotations

from . import ops
from .optim import FusedAdam


class TrainStep:
    def __init__(self, model, lr=1e-4, weight_decay=1e-4, grad_sync=None):
        self.model = model
        self.optimizer = FusedAdam(model.parameters(), lr=lr, weight_decay=weight_decay)
        self.grad_sync = grad_sync
        if grad_sync is not None:
            grad_sync.attach(self.optimizer)

    def __call__(self, images, masks):
        self.optimizer.zero_grad(set_to_none=True)
        prob = self.model(images)
        loss = ops.bce_loss(prob, masks)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync.finish()
        self.optimizer.step()
        return loss
