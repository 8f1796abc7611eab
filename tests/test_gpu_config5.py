"""GPU: BASELINE.json configs[4] AS WRITTEN - "Robust U-Net 1024x1024 tiles fp16, all-reduce/backward overlap + hipGraph-captured step":
one 1 x 1024^2 tile through set_precision("fp16") + TrainStep(loss_scale=1024, graph=True)

  * the replayed step is bit-identical to the eager fp16 step (losses, parameters, BatchNorm buffers), no step skipped at scale 1024;
  * the step agrees with the fp32 CPU oracle within the fp16 bands stated below (the reference is fp32 only: the bands are this
    repo's, "parity unpinned" by the reference for reduced precision);
  * the dynamic loss scale works under the captured step: a forced overflow is skipped on the device, adjust_loss_scale() lowers the
    scale the REPLAYED graph multiplies the loss by (it is read from device memory), eager and graph stay bit-identical;
  * the captured step carries the RCCL gradient all-reduce: a ONE-rank RCCL process group (the one-GPU box cannot hold more), buckets on
    the communication stream, replayed parameters bit-equal to the eager data-parallel step (child process: it owns the process group).
"""
import importlib
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fp16_model(pkg, oracle, st, masks):
    m = pkg.RobustUNet(3, 1, 64)
    m.load_state_dict(st)
    m = m.to(DEV).train().set_precision("fp16")
    m.set_dropout_masks({k: v.to(DEV) for k, v in masks.items()})
    return m


@pytest.mark.timeout(900)
def test_config5_fp16_graph_step_against_the_oracle(pkg, oracle):
    trainer = importlib.import_module("eusipco-2026-robust-unet_amd.trainer")
    n, size, seed, scale = 1, 1024, 47, 1024.0
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    st = oracle.init_state(3, 1, 64, seed=seed, perturb_bn=True)
    masks = oracle.dropout_masks(n, 64, seed=seed)
    batches = [pkg.synthetic_batch(n, size, seed=seed + i) for i in range(3)]
    res = {}
    for graph in (False, True):
        m = _fp16_model(pkg, oracle, st, masks)
        step = trainer.TrainStep(m, lr=1e-4, weight_decay=1e-4, graph=graph, graph_warmup=1, loss_scale=scale)
        losses, g0 = [], None
        for i, (x, y) in enumerate(batches):
            losses.append(step(x.to(DEV), y.to(DEV)).detach().clone())
            if i == 0:
                g0 = (m.grad_arena().flat.detach().double() / scale).cpu()       # p.grad holds the SCALED gradient
                named = {k: (p.grad.detach().double() / scale).cpu() for k, p in m.named_parameters()}
        assert step.adjust_loss_scale() == (scale, 0)                          # nothing overflowed, nothing skipped
        assert {s["step"] for s in step.optimizer.state.values()} == {3}
        if graph:
            assert step._graph is not None
        res[graph] = (losses, [p.detach().clone() for p in m.parameters()], [b.detach().clone() for b in m.buffers()], g0, named)
    for a, b in zip(res[False][0], res[True][0]):
        assert torch.equal(a, b), (float(a), float(b))
    for a, b in zip(res[False][1] + res[False][2], res[True][1] + res[True][2]):
        assert torch.equal(a, b)
    # ---- against the fp32 oracle (first step)
    x, y = batches[0]
    m = _fp16_model(pkg, oracle, st, masks)
    with torch.no_grad():
        prob, logit = m(x.to(DEV), return_logits=True)
    names = oracle.param_names(3, 1, 64)
    P = {k: v.clone() for k, v in st.items()}
    for k in names:
        P[k].requires_grad_(True)
    rp, rl = oracle.forward(P, x, True, masks)
    rloss = oracle.bce_mean(rp, y)
    rloss.backward()
    loss = res[True][0][0].detach()
    lscale = float(rl.detach().abs().max())
    lerr = float((logit.cpu() - rl.detach()).abs().max())
    perr = float((prob.cpu() - rp.detach()).abs().max())
    g = torch.cat([res[True][4][k].reshape(-1) for k in names])
    r = torch.cat([P[k].grad.double().reshape(-1) for k in names])
    cos = float((g @ r) / (g.norm() * r.norm()))
    # per-tensor norms.  A bias in front of a BatchNorm (the attention gates' W_g.0 / W_x.0 / psi.0, the DilatedBlock's conv biases) has an
    # analytically ZERO gradient - BatchNorm removes the mean it adds - so what either side computes there is rounding noise of a
    # cancelling sum; those tensors are left out of the norm statistic (they are inside the cosine, where they weigh nothing)
    zero_grad = [k for k in names if (k.startswith("att") and k.endswith(".0.bias")) or (k.startswith("bottleneck.1.conv") and k.endswith(".bias"))]
    keep = [k for k in names if k not in zero_grad]
    gn = np.array([float(res[True][4][k].norm()) for k in keep])
    rn = np.array([float(P[k].grad.double().norm()) for k in keep])
    rel = np.abs(gn - rn) / np.maximum(rn, 1e-30)
    worst = keep[int(rel.argmax())]
    print(f"\nconfig 5 (1 x 1024^2 fp16, scale {scale:g}, hipGraph): loss {float(loss):.5f} vs oracle {float(rloss):.5f}; logit err {lerr:.3e} of scale "
          f"{lscale:.2f}; prob err {perr:.3e}; gradient cosine {cos:.6f}; grad-norm rel err 50/90/100 %: {np.percentile(rel, [50, 90, 100])} (worst {worst}, {len(zero_grad)} zero-gradient biases left out)")
    # fp16 keeps 11 significant bits.  Stated bands (same as the 2 x 64^2 fp16 test, tests/test_gpu_bf16.py): loss within 0.2 % + two
    # BCE-clamp quanta (a pixel rounding to p == 1.0 against label 0 costs 100 / numel), logits within 0.5 % of their scale, gradient
    # cosine >= 0.995, per-tensor gradient norms: median within 1 %, none off by more than its own size
    assert abs(float(loss) - float(rloss)) <= 2e-3 * max(1.0, abs(float(rloss))) + 2 * 100.0 / y.numel()
    assert lerr <= 5e-3 * lscale and perr <= 0.05
    assert cos >= 0.995
    assert np.median(rel) <= 1e-2 and rel.max() <= 1.0


def test_dynamic_loss_scale_under_the_captured_step(pkg, oracle):
    """ADVICE (round 2): the loss scale used to be a Python float baked into the capture.  Scenario on an eager and on a graph step:
    normal step, overflowing step (absurd scale -> skipped on the device), adjust_loss_scale(), scale set back, normal steps."""
    trainer = importlib.import_module("eusipco-2026-robust-unet_amd.trainer")
    base, n, size, seed = 16, 2, 64, 5
    st = oracle.init_state(3, 1, base, seed=seed, perturb_bn=True)
    masks = oracle.dropout_masks(n, base, seed=seed)
    x, y = pkg.synthetic_batch(n, size, seed=seed)
    x, y = x.to(DEV), y.to(DEV)
    out = {}
    for graph in (False, True):
        m = pkg.RobustUNet(3, 1, base)
        m.load_state_dict(st)
        m = m.to(DEV).train().set_precision("fp16")
        m.set_dropout_masks({k: v.to(DEV) for k, v in masks.items()})
        step = trainer.TrainStep(m, lr=1e-3, weight_decay=0.0, graph=graph, graph_warmup=1, loss_scale=256.0, scale_window=3)
        step(x, y)
        step(x, y)                                    # graph: capture + first replay
        snap = [p.detach().clone() for p in m.parameters()]
        step.set_loss_scale(1e38)                     # scaled gradients overflow fp16 -> Inf -> skipped on the device
        step(x, y)
        for a, p in zip(snap, m.parameters()):
            assert torch.equal(a, p.detach())
        scale, skipped = step.adjust_loss_scale()
        assert skipped == 1 and scale == 1e38 / 2.0
        assert {s["step"] for s in step.optimizer.state.values()} == {2}          # the skipped step did not advance Adam
        step.set_loss_scale(256.0)
        for _ in range(3):
            step(x, y)
        gmax = float(m.grad_arena().flat.abs().max())
        assert np.isfinite(gmax) and gmax / 256.0 < 1e3                            # the replay multiplied by 256, not by 1e38
        moved = max(float((a - p.detach()).abs().max()) for a, p in zip(snap, m.parameters()))
        assert 0 < moved <= 5e-3                                                   # three un-scaled Adam steps of lr 1e-3 (not ~1e35)
        scale, skipped = step.adjust_loss_scale()
        assert skipped == 1 and scale == 512.0                                     # 3 clean STEPS >= scale_window -> doubled
        assert {s["step"] for s in step.optimizer.state.values()} == {5}
        step(x, y)
        out[graph] = [p.detach().clone() for p in m.parameters()]
        if graph:
            assert step._graph is not None
    for a, b in zip(out[False], out[True]):
        assert torch.equal(a, b)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
def test_captured_step_carries_the_rccl_allreduce():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="8")
    # torch's ProcessGroupNCCL watchdog thread polls the end events of collectives every 100 ms; if a poll falls into the ~15 ms in which this
    # process captures the step it can hit the event of a collective recorded inside the capture ("operation not permitted on an event last
    # recorded in a capturing stream") and aborts the process - seen once in about 25 runs of this child, never reproduced in 10 runs in a row
    # (DESIGN.md section 6).  That is torch's thread, not the code under test: such a run is repeated, any other failure is not.
    for attempt in range(3):
        env["MASTER_PORT"] = str(_free_port())
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "graph_ddp_child.py")], env=env, capture_output=True, text=True, timeout=500)
        print(r.stdout[-3000:], r.stderr[-3000:])
        if r.returncode == 0 or "last recorded in a capturing stream" not in r.stderr:
            break
    assert r.returncode == 0, r.stderr[-2000:]
    assert "GRAPH_DDP_OK" in r.stdout
