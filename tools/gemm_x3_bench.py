"""The head GEMMs of an episode on the native fp32 MFMA kernel (lmkd_gemm_set_mode 0), on the 3 x bf16 kernel (1) and with one bf16 plane (2):
us per call, TFLOP/s, relative L2 error against an fp64 product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd  # noqa: F401
from litemkd_amd import ops
lib = litemkd_amd.lib()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)


def run(name, layA, layB, M, N, K, batch=1, **kw):
    # operands in their memory layouts
    A = rn(batch, M, K) if layA == "K" else rn(batch, K, M)
    B = rn(batch, N, K) if layB == "K" else rn(batch, K, N)
    C = torch.zeros(batch, M, N, device=dev)
    A64 = (A if layA == "K" else A.transpose(1, 2)).double()
    B64 = (B if layB == "K" else B.transpose(1, 2)).double()
    ref = A64 @ B64.transpose(1, 2)
    lda = K if layA == "K" else M
    ldb = K if layB == "K" else N
    out = []
    for mode, tile in ((0, 0), (1, 0), (1, 2), (1, 1), (2, 0), (2, 1)):
        lib.call("lmkd_gemm_set_mode", mode)
        lib.call("lmkd_gemm_set_tile", tile)
        f = lambda: ops.gemm(layA, layB, M, N, K, A, lda, B, ldb, C, N, batch=batch, sA=A[0].numel(), sB=B[0].numel(), sC=M * N)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 100)
        err = float((C.double() - ref).norm() / ref.norm())
        out.append("%6.1f us %5.1f TF %.0e" % (best, 2.0 * M * N * K * batch / best / 1e6, err))
    lib.call("lmkd_gemm_set_mode", -1)
    lib.call("lmkd_gemm_set_tile", 0)
    print("%-34s %s%s M %4d N %4d K %4d x%d | fp32 %s | x3 64x64 %s | x3 64x128 %s | x3 128x128 %s | bf16 64x64 %s | bf16 128x128 %s" % ((name, layA, layB, M, N, K, batch) + tuple(out)))


run("TRX projection (per k / v weight)", "K", "K", 400, 1152, 2048, 2)
run("TRX projection, one Wcat", "K", "K", 400, 4608, 2048)
run("TRX projection dW", "M", "N", 1152, 2048, 400, 2)
run("TRX projection dX (per block)", "K", "N", 400, 2048, 1152)
run("TRX projection dX (Wcat, K 4608)", "K", "N", 400, 2048, 4608)
run("TRX dX as 2-way split", "K", "N", 400, 2048, 2304, 2)
run("fc forward", "K", "K", 400, 2048, 512)
run("fc dX (both heads)", "K", "N", 400, 512, 4096)
run("fc dW", "M", "N", 2048, 512, 400)
run("scores Qk Sk^T", "K", "K", 700, 700, 1152)
run("prototypes (per class)", "K", "N", 700, 1152, 140, 5)
run("d scores", "K", "K", 700, 140, 1152, 5)
run("dQk", "K", "N", 700, 1152, 700)
run("dSk", "M", "N", 700, 1152, 700)
run("1-shot projection", "K", "K", 240, 4608, 2048)
