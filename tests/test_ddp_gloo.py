"""CPU, world_size 2 over gloo: the bucketed gradient all-reduce (ddp.GradAllReducer) on the model's flat
gradient arena.  The kernels cannot run here, so the per-rank gradients come from the oracle (test checker
standing in for the device backward) in eval-BN mode, where data-parallel averaging is exactly the
full-batch gradient."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

PKG = "eusipco-2026-robust-unet_amd"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    pkg = importlib.import_module(PKG)
    M = importlib.import_module(PKG + ".model")
    oracle = importlib.import_module("oracle.robust_unet_ref")
    base, n, size = 16, 4, 32
    torch.manual_seed(100 + rank)                       # ranks start from DIFFERENT weights ...
    model = pkg.RobustUNet(3, 1, base)
    red = pkg.GradAllReducer(model, bucket_floats=300_000, average_in_optimizer=False)
    red.broadcast_parameters(0)                         # ... and must end up with rank 0's
    red.attach()
    arena = model.grad_arena()
    st = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = oracle.param_names(3, 1, base)
    x, y = pkg.synthetic_batch(n, size, seed=3)

    def grads(xs, ys):
        P = {k: v.clone() for k, v in st.items()}
        for k in names:
            P[k].requires_grad_(True)
        prob, _ = oracle.forward(P, xs, training=False)
        oracle.bce_mean(prob, ys).backward()
        return {k: P[k].grad for k in names}

    half = n // world
    local = grads(x[rank * half:(rank + 1) * half], y[rank * half:(rank + 1) * half])
    named = dict(model.named_parameters())
    for k in names:                                      # what the device backward does: write through arena views
        arena.grad_view(k, named[k]).copy_(local[k])
    for blk in M.BACKWARD_ORDER:                         # blocks complete in backward order; buckets launch on the way
        arena.done(blk)
    red.finish()
    full = grads(x, y)
    worst = max(float((arena.grad_view(k, named[k]) - full[k]).abs().max() / (full[k].abs().max() + 1e-12)) for k in names)
    w0 = float(model.inc.conv1.weight.double().sum())
    q.put((rank, worst, red.buckets_last_step, arena.total, w0))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_bucketed_allreduce_world2_equals_full_batch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=540) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, worst0, buckets0, total, w0), (r1, worst1, buckets1, _, w1) = out
    assert worst0 < 1e-4 and worst1 < 1e-4                 # mean of the two half-batch gradients == full-batch gradient
    assert buckets0 == buckets1 and len(buckets0) >= 3     # several buckets, identical on every rank
    assert buckets0[0][0] == 0 and buckets0[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(buckets0, buckets0[1:]))      # contiguous cover of the arena
    assert w0 == w1                                        # broadcast_parameters made the replicas identical
