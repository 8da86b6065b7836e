"""lmkd_gemm_f32 on conv-equivalent GEMM shapes (tuning aid)."""
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import litemkd_amd
from litemkd_amd import ops
dev = torch.device("cuda", 0)
def bench(M, N, K, la="K", lb="N", reps=10):
    A = torch.relu(torch.randn((M, K) if la == "K" else (K, M), device=dev)); B = torch.randn((N, K) if lb == "K" else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    f = lambda: ops.gemm(la, lb, M, N, K, A, A.shape[1], B, B.shape[1], C, N)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("gemm %s%s %dx%dx%d: %.3f ms %.1f TF" % (la, lb, M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
bench(8192, 8192, 4096)
bench(156800, 128, 1152)
bench(39200, 256, 2304)
bench(9800, 512, 4608)
bench(627200, 64, 576)
