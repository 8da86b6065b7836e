"""fp64-anchored error criteria shared by the GPU parity tests.

A HIP result and a torch-CPU-fp32 result of the same computation are both compared with an fp64 evaluation of it; the HIP
path passes when its error is no worse than `factor` x the error torch's own fp32 CPU kernels make on the same inputs
(plus a floor at fp32 rounding level).  For gradients that pass through ReLU masks this replaces a fixed relative-L2 budget:
if a pre-activation within rounding of zero flips a mask, BOTH fp32 evaluations show errors of that size against fp64, and a
scheduling / indexing bug in the HIP backward cannot hide inside a percent-level budget when the CPU fp32 error is 1e-6."""
import math

import torch


def rel_l2(a, ref64, floor=0.0):
    a = a.detach().cpu().double().reshape(-1)
    r = ref64.detach().cpu().double().reshape(-1)
    return float((a - r).norm() / (r.norm() + floor))


def anchored(name, hip, cpu32, ref64, factor=3.0, floor=2e-6, abs_floor=0.0):
    """-> (e_hip, e_cpu); asserts e_hip <= factor * e_cpu + floor.  abs_floor (same units as the tensor, times sqrt(numel))
    is added to the reference norm so that tensors whose exact value is 0 are compared on an absolute scale."""
    add = abs_floor * math.sqrt(max(1, ref64.numel()))
    e_hip, e_cpu = rel_l2(hip, ref64, add), rel_l2(cpu32, ref64, add)
    assert e_hip <= factor * e_cpu + floor, "%s: HIP-vs-fp64 rel-L2 %.3e > %g x torch-CPU-fp32-vs-fp64 %.3e + %.0e" % (
        name, e_hip, factor, e_cpu, floor)
    return e_hip, e_cpu


def anchored_dict(hip, cpu32, ref64, factor=3.0, floor=2e-6, zero_scale=1e-7):
    """per-tensor anchored() over dicts of gradients with the same keys; tensors with (near-)zero exact gradient are judged
    on an absolute floor of zero_scale x the largest |gradient| in the set.  -> worst (ratio, key, e_hip, e_cpu)"""
    gmax = max(float(v.abs().max()) for v in ref64.values())
    worst = (0.0, "", 0.0, 0.0)
    for k, r in ref64.items():
        e_hip, e_cpu = anchored(k, hip[k], cpu32[k], r, factor, floor, zero_scale * gmax)
        ratio = e_hip / (e_cpu + floor / factor)
        if ratio > worst[0]:
            worst = (ratio, k, e_hip, e_cpu)
    return worst


def record(name, pairs):
    """append the measured (HIP-vs-fp64, torch-CPU-fp32-vs-fp64) relative-L2 error pairs of an anchored test to the file named
    by $LMKD_PARITY_LOG (the GPU run that produces profiles/rNN_parity_errors.txt sets it), so the fp32-class claim of the
    arithmetic modes can be audited from a tracked file"""
    import os
    path = os.environ.get("LMKD_PARITY_LOG")
    if not path:
        return
    with open(path, "a") as f:
        f.write("%-64s %s\n" % (name, "  ".join("%s: hip %.3e cpu %.3e" % (k, v[0], v[1]) for k, v in pairs.items())))
