"""Inference episodes (test(), trainwandb.py:359-417; test.py:65-318): forward only, model.eval(), 5-way 5-shot 8x224^2 (400 frames),
with the eval-mode BatchNorm (+ residual, ReLU) in the convolution epilogue (lmkd_conv2d_fwd_bn) against the two-pass form, in each
arithmetic mode; plus the convolution kernels' rate in a serialized eval episode (HIP events per launch).
Prints one JSON line (-> profiles/rNN_eval.json).   `gpurun -- python tools/eval_bench.py`"""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import litemkd_amd  # noqa: F401
from litemkd_amd import ops, trainloop as TL
from litemkd_amd.model.model_select import Student
from litemkd_amd.model.backbone import resnet as R
from litemkd_amd.options import default_args
from litemkd_amd.utils import aggregate_accuracy
dev = torch.device("cuda", 0)
cfg = default_args(shot=5, device=dev, trans_dropout=0.1)
torch.manual_seed(0)
student = Student(cfg).to(dev).eval()
src = TL.SyntheticEpisodes(cfg, base_seed=7, rank=0, device=dev)
pool = [src.episode(e) for e in range(2)]


def run(n):
    with torch.no_grad():
        for i in range(n):
            TL.test_task(pool[i % 2], student, aggregate_accuracy, cfg)


out = {"metric": "inference episodes/s (5-way 5-shot, 400 frames of 224^2, model.eval())", "modes": {}}
PEAK = {"fp32h2": 2500.0 / 3, "fp32x3": 2500.0 / 6, "fp32": 157.3, "bf16": 2500.0}
for mode in ("fp32h2", "fp32x3", "fp32", "bf16"):
    ops.set_conv_compute_dtype(mode)
    res = {}
    for fused in (True, False):
        ops.FUSE_EVAL_BN = fused
        best = 1e9
        for rep in range(2):
            run(3); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(20); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 20)
        res["fused" if fused else "two_pass"] = {"ms_per_episode": best * 1e3, "episodes_per_s": 1 / best}
    ops.FUSE_EVAL_BN = True
    # convolution rate of the fused forward, serialized (one stream), HIP events around every launch
    R.OVERLAP_TRUNK_CALLS = False
    ops.CONV_TIMING = []
    run(2); torch.cuda.synchronize()
    rec, ops.CONV_TIMING = ops.CONV_TIMING, None
    R.OVERLAP_TRUNK_CALLS = True
    fl = sum(r[1] for r in rec); tt = sum(r[2].elapsed_time(r[3]) for r in rec) * 1e-3
    res["conv_tflops_serial"] = fl / tt / 1e12
    res["conv_frac_of_peak"] = fl / tt / 1e12 / PEAK[mode]
    res["conv_launches_per_episode"] = len(rec) / 2
    out["modes"][mode] = res
ops.reset_compute_dtypes()
print(json.dumps(out), flush=True)
