"""Run ONE batched position-GEMM shape a few times (for rocprofv3 --pmc): gemm_x3_one.py rows k n [nn|tn] [iters] [batch]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
L = importlib.import_module("eusipco-2026-robust-unet_amd._lib")
lib, check = L.lib, L.check
rows, k, n = (int(v) for v in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "nn"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
B = int(sys.argv[6]) if len(sys.argv) > 6 else 36
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
a = torch.randn(B, rows, k, device=dev)
b = torch.randn(B, k, n, device=dev)
z = torch.randn(B, rows, n, device=dev)
c = torch.empty(B, rows, n, device=dev)
bp = torch.empty(lib.runet_gemm_x3_pack_elems(B, k, n), device=dev, dtype=torch.bfloat16)
check(lib.runet_gemm_x3_pack(b.data_ptr(), k * n, bp.data_ptr(), B, k, n, st))
blocks = ((k + 127) // 128) * ((n + 127) // 128) * B
s = max(1, min(rows // 64, -(-512 // blocks)))
rps = (-(-rows // s) + 15) // 16 * 16
cu = torch.empty(-(-rows // rps), B, k, n, device=dev)
for _ in range(iters):
    if mode == "nn":
        check(lib.runet_gemm_x3_batched(a.data_ptr(), k, rows * k, bp.data_ptr(), c.data_ptr(), n, rows * n, B, rows, k, n, st))
    else:
        check(lib.runet_gemm_x3_tn_batched(a.data_ptr(), k, rows * k, z.data_ptr(), n, rows * n, cu.data_ptr(), B, rows, k, n, rps, st))
torch.cuda.synchronize()
