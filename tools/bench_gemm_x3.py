"""The position-GEMM shapes of the F(4x4,3x3) path (batch 36) through the f32-MFMA kernels (gemm.hip) and through the split-operand
kernels on the bf16 matrix cores (gemm_split.hip): time, executed TFLOP/s and the error against a float64 product of the same operands
(max |c - ref| / max |ref| and RMS), side by side.  BENCH_SCALE=<float> multiplies the operands (range check of the split)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

L = importlib.import_module("eusipco-2026-robust-unet_amd._lib")
lib, check = L.lib, L.check
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
SHAPES = [(4096, 256, 256), (4096, 512, 256), (1024, 512, 512), (1024, 1024, 512), (256, 1024, 1024), (16384, 256, 128), (1024, 512, 256), (200, 64, 36)]


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def errs(c, ref):
    d = (c.double() - ref)
    return float(d.abs().max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())


scale = float(os.environ.get("BENCH_SCALE", "1"))
for rows, k, n in SHAPES:
    B = 36
    g = torch.Generator(device=dev).manual_seed(rows + k + n)
    a = torch.randn(B, rows, k, device=dev, generator=g) * scale
    b = torch.randn(B, k, n, device=dev, generator=g)
    c0 = torch.empty(B, rows, n, device=dev)
    c1 = torch.empty(B, rows, n, device=dev)
    bp = torch.empty(lib.runet_gemm_x3_pack_elems(B, k, n), device=dev, dtype=torch.bfloat16)
    check(lib.runet_gemm_x3_pack(b.data_ptr(), k * n, bp.data_ptr(), B, k, n, st()))
    t0 = timeit(lambda: check(lib.runet_gemm_batched(a.data_ptr(), k, rows * k, b.data_ptr(), k * n, c0.data_ptr(), n, rows * n, B, rows, k, n, st())))
    t1 = timeit(lambda: check(lib.runet_gemm_x3_batched(a.data_ptr(), k, rows * k, bp.data_ptr(), c1.data_ptr(), n, rows * n, B, rows, k, n, st())))
    ref = torch.bmm(a[:3].double(), b[:3].double())
    e0, e1 = errs(c0[:3], ref), errs(c1[:3], ref)
    flop = 2.0 * B * rows * k * n
    line = (f"rows {rows:6d} k {k:5d} n {n:5d}: NN f32-mfma {t0:6.3f} ms {flop / t0 / 1e9:6.1f} TF err {e0[0]:.1e}/{e0[1]:.1e} | "
            f"bf16x6 [{lib.runet_gemm_x3_kernel_name(B, rows, k, n).decode()[-4:-1]:>3}] {t1:6.3f} ms {flop / t1 / 1e9:6.1f} TF err {e1[0]:.1e}/{e1[1]:.1e}")
    if rows % 16 == 0:
        bz = torch.randn(B, rows, n, device=dev, generator=g)
        blocks = ((k + 127) // 128) * ((n + 127) // 128) * B
        s = max(1, min(rows // 64, -(-512 // blocks)))
        rps = (-(-rows // s) + 15) // 16 * 16
        splits = -(-rows // rps)
        cu0 = torch.empty(splits, B, k, n, device=dev)
        cu1 = torch.empty(splits, B, k, n, device=dev)
        t2 = timeit(lambda: check(lib.runet_gemm_tn_batched(a.data_ptr(), k, rows * k, bz.data_ptr(), n, rows * n, cu0.data_ptr(), B, rows, k, n, rps, st())))
        t3 = timeit(lambda: check(lib.runet_gemm_x3_tn_batched(a.data_ptr(), k, rows * k, bz.data_ptr(), n, rows * n, cu1.data_ptr(), B, rows, k, n, rps, st())))
        ref2 = torch.bmm(a[:3].double().transpose(1, 2), bz[:3].double())
        e2, e3 = errs(cu0.double().sum(0)[:3], ref2), errs(cu1.double().sum(0)[:3], ref2)
        line += (f" || TN x{splits} f32-mfma {t2:6.3f} ms {flop / t2 / 1e9:6.1f} TF err {e2[0]:.1e}/{e2[1]:.1e} | bf16x6 {t3:6.3f} ms {flop / t3 / 1e9:6.1f} TF "
                 f"err {e3[0]:.1e}/{e3[1]:.1e}")
    print(line, flush=True)
