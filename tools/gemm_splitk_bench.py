"""lmkd_gemm_f32 vs lmkd_gemm_f32_splitk on the head's GEMM shapes (400 frames / 700 tuples), microseconds per launch."""
import sys
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import litemkd_amd  # noqa: F401
from litemkd_amd import ops

dev = torch.device("cuda", 0)
SH = [("K", "K", 400, 1152, 2048, 2, "projections fwd"), ("K", "N", 400, 2048, 1152, 1, "dX"), ("M", "N", 1152, 2048, 400, 2, "dW"),
      ("K", "K", 200, 512, 8192, 1, "fc-like"), ("K", "K", 700, 700, 1152, 1, "S = Qk Sk^T"), ("K", "N", 700, 1152, 140, 5, "proto"),
      ("M", "N", 700, 1152, 700, 1, "dSk")]
for layA, layB, M, N, K, b, name in SH:
    A = torch.randn(b, M, K, device=dev) if layA == "K" else torch.randn(b, K, M, device=dev)
    B = torch.randn(b, N, K, device=dev) if layB == "K" else torch.randn(b, K, N, device=dev)
    C = torch.empty(b, M, N, device=dev)
    out = []
    for split in (False, True):
        ops.GEMM_SPLIT_K = split
        f = lambda: ops.gemm(layA, layB, M, N, K, A, A.shape[2], B, B.shape[2], C, N, batch=b, sA=A[0].numel(), sB=B[0].numel(), sC=M * N)
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1000 / 50)
    fl = 2.0 * M * N * K * b
    print("%-18s %s%s M=%d N=%d K=%d b=%d   unsplit %.1f us (%.0f TF/s)   split %.1f us (%.0f TF/s)" % (name, layA, layB, M, N, K, b, out[0], fl / out[0] / 1e6, out[1], fl / out[1] / 1e6))
