
"""Micro-benchmark of the batched fp32 MFMA GEMM entry points at the F(4x4,3x3) position-GEMM shapes (batch 36)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

L = importlib.import_module("eusipco-2026-robust-unet_amd._lib")
lib, check = L.lib, L.check
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
SHAPES = [(4096, 256, 256), (4096, 512, 256), (1024, 512, 512), (1024, 1024, 512), (256, 1024, 1024), (16384, 256, 128)]   # rows, k, n


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for rows, k, n in SHAPES:
    B = 36
    a = torch.randn(B, rows, k, device=dev)
    b = torch.randn(B, k, n, device=dev)
    if os.environ.get("BENCH_ZERO") == "1":      # DVFS probe: all-zero operands draw less power -> higher clock (MI355X_MICROARCH.md)
        a.zero_(); b.zero_()
    c = torch.empty(B, rows, n, device=dev)
    t = timeit(lambda: check(lib.runet_gemm_batched(a.data_ptr(), k, rows * k, b.data_ptr(), k * n, c.data_ptr(), n, rows * n, B, rows, k, n, st())))
    ref = torch.bmm(a[:2], b[:2])
    err = float((c[:2] - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    flop = 2.0 * B * rows * k * n
    # TN: dU[z][k][n] = A[z][rows][k]^T . Bz[z][rows][n]
    bz = torch.randn(B, rows, n, device=dev)
    rps = (rows + 15) // 16 * 16
    blocks = ((k + 127) // 128) * ((n + 127) // 128) * B
    s = max(1, min(rows // 64, -(-512 // blocks)))
    rps = (-(-rows // s) + 15) // 16 * 16
    splits = -(-rows // rps)
    cu = torch.empty(splits, B, k, n, device=dev)
    t2 = timeit(lambda: check(lib.runet_gemm_tn_batched(a.data_ptr(), k, rows * k, bz.data_ptr(), n, rows * n, cu.data_ptr(), B, rows, k, n, rps, st())))
    ref2 = torch.bmm(a[:2].transpose(1, 2), bz[:2])
    err2 = float((cu.sum(0)[:2] - ref2).abs().max() / ref2.abs().max().clamp_min(1e-30))
    print(f"rows {rows:6d} k {k:5d} n {n:5d}: NN {t:7.3f} ms {flop / t / 1e9:6.1f} TF (err {err:.1e}) | TN x{splits} {t2:7.3f} ms {flop / t2 / 1e9:6.1f} TF (err {err2:.1e})",
          flush=True)
