"""per-tensor gradient errors (HIP and oracle-fp32 vs oracle-fp64) of one small episode, with the 16x16x32 patch kernel on / off"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import litemkd_amd
import _anchor
from litemkd_amd import ops
rows = []
orig = _anchor.anchored
def spy(name, hip, cpu32, ref64, factor=3.0, floor=2e-6, abs_floor=0.0):
    import math
    add = abs_floor * math.sqrt(max(1, ref64.numel()))
    e_hip, e_cpu = _anchor.rel_l2(hip, ref64, add), _anchor.rel_l2(cpu32, ref64, add)
    rows.append((name, e_hip, e_cpu))
    return e_hip, e_cpu
import test_gpu_episode as T
T.anchored = spy
dev = torch.device("cuda:0")
for p16 in (1, 0):
    litemkd_amd.lib().call("lmkd_conv_set_patch16", p16)
    ops.set_conv_compute_dtype("fp32x3")
    rows.clear()
    T._episode_matches_oracle(dev, 1, 1, 96, "TRX_2fcsup", "fc_2_sup_dist", "resnet18_2fc", False)
    rows.sort(key=lambda r: -r[1] / (r[2] + 1e-7))
    print("patch16 =", p16)
    for n, a, b in rows[:8]:
        print("   %-40s hip %.3e cpu %.3e ratio %.1f" % (n, a, b, a / (b + 1e-12)))
