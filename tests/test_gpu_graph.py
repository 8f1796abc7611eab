"""GPU: the hipGraph-captured train step (trainer.TrainStep(graph=True), BASELINE.json config 5's "hipGraph-captured step") replays
exactly what the eager step launches: same losses, bit-identical parameters, Adam's device-side step counter and learning rate
follow the host optimizer (LR schedulers keep working without re-capture)."""
import importlib
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _make(pkg, oracle, graph):
    trainer = importlib.import_module("eusipco-2026-robust-unet_amd.trainer")
    m = pkg.RobustUNet(3, 1, 16)
    m.load_state_dict(oracle.init_state(3, 1, 16, seed=4, perturb_bn=True))
    m = m.to(DEV).train()
    m.set_dropout_masks({k: v.to(DEV) for k, v in oracle.dropout_masks(2, 16, seed=4).items()})
    step = trainer.TrainStep(m, lr=1e-3, weight_decay=1e-4, graph=graph, graph_warmup=2)
    step.optimizer.capturable = True          # both sides on the device-hyper Adam kernel: the comparison can be bitwise
    return m, step


def test_graph_step_equals_eager_step(pkg, oracle):
    ma, sa = _make(pkg, oracle, graph=False)
    mb, sb = _make(pkg, oracle, graph=True)
    batches = [pkg.synthetic_batch(2, 64, seed=50 + i) for i in range(7)]
    for i, (x, y) in enumerate(batches):
        if i == 5:                             # an LR scheduler acting between steps
            for st in (sa, sb):
                st.optimizer.param_groups[0]["lr"] = 2.5e-4
        x, y = x.to(DEV), y.to(DEV)
        la, lb = sa(x, y), sb(x, y)
        assert torch.equal(la.detach(), lb.detach()), (i, float(la), float(lb))
    assert sb._graph is not None
    for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.equal(pa, pb), k
    for (k, ba), (_, bb) in zip(ma.named_buffers(), mb.named_buffers()):
        assert torch.equal(ba, bb), k
    steps = {st["step"] for st in sb.optimizer.state.values()}
    assert steps == {7}
    # a different batch shape falls back to an ordinary step
    mb.set_dropout_masks(None)                 # the injected masks are for 2 images
    x, y = pkg.synthetic_batch(1, 64, seed=99)
    assert torch.isfinite(sb(x.to(DEV), y.to(DEV)))


def _time_steps(pkg, trainer, graph, n, size, base=64, steps=20):
    torch.manual_seed(0)
    m = pkg.RobustUNet(3, 1, base).to(DEV).train()
    step = trainer.TrainStep(m, graph=graph)
    x, y = pkg.synthetic_batch(n, size, seed=1)
    x, y = x.to(DEV), y.to(DEV)
    for _ in range(6):
        step(x, y)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):                      # best of three windows: the box is shared
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step(x, y)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    assert torch.isfinite(loss)
    return best


def test_graph_step_pays_where_the_step_is_launch_bound(pkg, oracle):
    """BASELINE config 1 (2 x 64^2, ~650 launches of a few microseconds each): the eager step is bound by the host issuing launches
    (measured 10.1 ms, 8.6 ms after the host-side fixes), the replayed graph by the device-side kernel boundaries (8.4 -> 7.4 ms).  At 2 x 256^2 the GPU is the bottleneck either
    way and the replay only loses the weight-gradient side stream (a captured fork / join replays 2x slower on this runtime: 25.4 vs
    12.4 ms, profiles/README.md), so it may be a few per cent slower - bounded here."""
    trainer = importlib.import_module("eusipco-2026-robust-unet_amd.trainer")
    e1, g1 = _time_steps(pkg, trainer, False, 2, 64), _time_steps(pkg, trainer, True, 2, 64)
    e2, g2 = _time_steps(pkg, trainer, False, 2, 256), _time_steps(pkg, trainer, True, 2, 256)
    print(f"\n2 x 64^2: eager {e1 * 1e3:.2f} ms, hipGraph {g1 * 1e3:.2f} ms;  2 x 256^2: eager {e2 * 1e3:.2f} ms, hipGraph {g2 * 1e3:.2f} ms")
    # measured after the round's host-side work: 2 x 64^2 eager 8.6 ms / graph 7.4 ms, 2 x 256^2 eager 9.8 / graph 10.2; the eager side is
    # host time and varies ~10 % between boxes, hence the margins
    assert g1 <= 1.1 * e1, (g1, e1)
    assert g2 <= 1.3 * e2, (g2, e2)          # not a speed claim: the replay is single-stream by design, the bound only catches a 2x regression


def test_graph_replay_at_config5_tile(pkg, oracle):
    """BASELINE.json config 5: one 1024 x 1024 tile per GPU through the hipGraph-captured step - replayed losses and parameters
    bit-identical to the eager step's."""
    trainer = importlib.import_module("eusipco-2026-robust-unet_amd.trainer")
    res = {}
    for graph in (False, True):
        m = pkg.RobustUNet(3, 1, 64)
        m.load_state_dict(oracle.init_state(3, 1, 64, seed=8, perturb_bn=True))
        m = m.to(DEV).train()
        m.set_dropout_masks({k: v.to(DEV) for k, v in oracle.dropout_masks(1, 64, seed=8).items()})
        step = trainer.TrainStep(m, lr=1e-4, weight_decay=1e-4, graph=graph, graph_warmup=1)
        step.optimizer.capturable = True
        losses = []
        for i in range(4):
            x, y = pkg.synthetic_batch(1, 1024, seed=70 + i)
            losses.append(step(x.to(DEV), y.to(DEV)).detach().clone())
        res[graph] = (losses, [p.detach().clone() for p in m.parameters()])
        if graph:
            assert step._graph is not None
    for a, b in zip(res[False][0], res[True][0]):
        assert torch.equal(a, b)
    for a, b in zip(res[False][1], res[True][1]):
        assert torch.equal(a, b)


def test_models_without_an_arena_keep_their_gradients_in_place_and_capture(pkg):
    """DeepLabV3+ (and the plain U-Net) have no flat gradient arena: ops.deliver_grads lands their gradients in per-parameter buffers that live
    as long as the model, so p.grad keeps its address from step to step (FusedAdam's device pointer table is uploaded once, the host runs ahead
    of the GPU) and TrainStep(graph=True) can capture the step: replay == eager, bit for bit."""
    trainer = importlib.import_module("eusipco-2026-robust-unet_amd.trainer")
    res = {}
    for graph in (False, True):
        torch.manual_seed(4)
        m = pkg.DeepLabV3Plus(n_classes=1).to(DEV).train()
        st = trainer.TrainStep(m, lr=1e-3, weight_decay=1e-4, graph=graph)
        st.optimizer.capturable = True             # the device-side step counter / bias correction in both (a captured step needs it)
        ptrs, losses = [], []
        for i in range(6):
            x, y = pkg.synthetic_batch(2, 64, seed=70 + i)
            losses.append(st(x.to(DEV), y.to(DEV)).detach().clone())
            ptrs.append([p.grad.data_ptr() for p in m.parameters()])
        torch.cuda.synchronize()
        if graph:
            assert st._graph is not None
        else:
            assert all(a == ptrs[0] for a in ptrs[1:]), "p.grad moved between eager steps"
        res[graph] = (losses, [p.detach().clone() for p in m.parameters()], [b.detach().clone() for b in m.buffers()])
    for a, b in zip(res[False][0], res[True][0]):
        assert torch.equal(a, b), (float(a), float(b))
    for a, b in zip(res[False][1] + res[False][2], res[True][1] + res[True][2]):
        assert torch.equal(a, b)
