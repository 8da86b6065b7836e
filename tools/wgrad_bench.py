"""3x3 / stride-1 weight gradients of the trunk at 200 frames: rolling-window kernel (wgrad_win.h) against the im2col-gather kernel
(wgrad_x3.h), per arithmetic mode, configurations interleaved (best of three rounds), with the relative difference of the results.
`gpurun -- python tools/wgrad_bench.py [mode ...]`   modes: fp32x3 bf16 bf16act"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import litemkd_amd  # noqa: F401
from litemkd_amd import ops
from litemkd_amd._lib import lib
dev = torch.device("cuda", 0)


def tm(f, reps=6):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


N = int(os.environ.get("FRAMES", "200"))
shapes = [("l1", 64, 56, 64), ("l2", 128, 28, 128), ("l3", 256, 14, 256), ("l4", 512, 7, 512)]
WINS = [int(v) for v in os.environ.get("WINS", "0,1").split(",")]
modes = sys.argv[1:] or ["fp32x3", "bf16", "bf16act"]
for mode in modes:
    ops.set_conv_compute_dtype("bf16" if mode.startswith("bf16") else mode)
    ops.set_activation_dtype("bf16" if mode == "bf16act" else "fp32")
    dt = torch.bfloat16 if mode == "bf16act" else torch.float32
    for (name, C, H, Cout) in shapes:
        x = torch.relu(torch.randn(N, H, H, C, device=dev)).to(dt)
        gy = torch.randn(N, H, H, Cout, device=dev).to(dt)
        fl = 2.0 * N * H * H * Cout * C * 9
        res, best = {}, {}
        for rnd_ in range(3):
            for win in WINS:
                lib().call("lmkd_conv_set_wgrad_window", win)
                if rnd_ == 0:
                    res[win] = ops.conv_bwd_weight(x, gy, (Cout, C, 3, 3), 1, 1)
                t = tm(lambda: ops.conv_bwd_weight(x, gy, (Cout, C, 3, 3), 1, 1))
                best[win] = min(best.get(win, 1e9), t)
        ref = torch.nn.functional.conv2d(x.float().permute(3, 0, 1, 2).double()[:, :8], gy.float().permute(3, 0, 1, 2).double()[:, :8], padding=1).permute(1, 0, 2, 3) if False else None
        d = float((res[WINS[0]].double() - res[WINS[-1]].double()).norm() / res[WINS[0]].double().norm())
        print("%-8s %-3s " % (mode, name) + " | ".join("%s %6.1f us (%5.1f TF)" % ("gather" if w_ == 0 else "window(%d)" % w_, best[w_] * 1e3, fl / best[w_] / 1e9) for w_ in WINS) + " | rel diff %.2e" % d, flush=True)
    lib().call("lmkd_conv_set_wgrad_window", 1)
ops.set_activation_dtype("fp32")
ops.set_conv_compute_dtype("fp32")
