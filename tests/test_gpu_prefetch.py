"""GPU: derived weights (Winograd-domain filters, packed bf16 / fp16 weights) refilled on the side stream at the start of the forward pass
(ops.prefetch_derived) give exactly the parameters of the lazy, on-first-use path - the refill only moves launches; forward-pass branches
(ops.side_branch) likewise."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(pkg, oracle, flags, precision, steps=5):
    ops = importlib.import_module("eusipco-2026-robust-unet_amd.ops")
    trainer = importlib.import_module("eusipco-2026-robust-unet_amd.trainer")
    old = ops.PREFETCH_DERIVED, ops.FWD_BRANCHES, ops.BRANCH_MIN_PIXELS
    ops.PREFETCH_DERIVED, ops.FWD_BRANCHES = flags
    ops.BRANCH_MIN_PIXELS = 0                         # the models switch branching off for small (host-bound) steps: force it for 2 x 64^2
    ops._derived.clear()
    try:
        m = pkg.RobustUNet(3, 1, 64)
        m.load_state_dict(oracle.init_state(3, 1, 64, seed=7, perturb_bn=True))
        m = m.to(DEV).train().set_precision(precision)
        m.set_dropout_masks({k: v.to(DEV) for k, v in oracle.dropout_masks(2, 64, seed=7).items()})
        step = trainer.TrainStep(m, lr=1e-3, weight_decay=1e-4, loss_scale=1024.0 if precision == "fp16" else None)
        losses, refilled = [], 0
        for i in range(steps):
            x, y = pkg.synthetic_batch(2, 64, seed=80 + i)
            losses.append(step(x.to(DEV), y.to(DEV)).detach().clone())
            refilled += sum(1 for e in ops._derived.values() if e[4] is not None)
        torch.cuda.synchronize()
        return m, losses, refilled
    finally:
        ops.PREFETCH_DERIVED, ops.FWD_BRANCHES, ops.BRANCH_MIN_PIXELS = old


@pytest.mark.parametrize("precision", ["f32", "bf16", "fp16"])
def test_prefetch_and_branches_change_nothing(pkg, oracle, precision):
    ma, la, refilled = _run(pkg, oracle, (True, True), precision)
    mb, lb, none = _run(pkg, oracle, (False, False), precision)
    assert refilled > 0, "no derived weight was refilled ahead of use"
    assert none == 0
    for i, (a, b) in enumerate(zip(la, lb)):
        assert torch.equal(a, b), (i, float(a), float(b))
    for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.equal(pa, pb), k
    for (k, ba), (_, bb) in zip(ma.named_buffers(), mb.named_buffers()):
        assert torch.equal(ba, bb), k


def test_torch_side_weight_write_is_seen(pkg, oracle):
    """a version-counter write after a refill (load_state_dict, torch.optim) must win over the refilled copy"""
    ops = importlib.import_module("eusipco-2026-robust-unet_amd.ops")
    m, _, _ = _run(pkg, oracle, (True, True), "f32", steps=2)
    x, _ = pkg.synthetic_batch(2, 64, seed=3)
    x = x.to(DEV)
    m.eval()
    with torch.no_grad():
        y0 = m(x).clone()
        for p in m.parameters():
            if p.dim() == 4:
                p.mul_(0.5)
        y1 = m(x).clone()
        ops._derived.clear()
        y2 = m(x).clone()
    assert not torch.equal(y0, y1)
    assert torch.equal(y1, y2)


def test_cache_does_not_keep_a_model_alive(pkg):
    """the derived-weight cache stores its refill closures: they capture raw pointers, the parameter is held by weak reference only"""
    import gc
    import weakref
    ops = importlib.import_module("eusipco-2026-robust-unet_amd.ops")
    m = pkg.RobustUNet(3, 1, 64).to(DEV).train()
    x, y = pkg.synthetic_batch(2, 64, seed=5)
    out = m(x.to(DEV))
    out.mean().backward()
    refs = [weakref.ref(p) for p in m.parameters()]
    assert any(e[2]() is not None for e in ops._derived.values())
    del m, out
    gc.collect()
    assert all(r() is None for r in refs), "a parameter survived its model: something in ops._derived holds it"
