"""debug aid: running statistics / gradients of Schedule.bench() vs Schedule.serial() on a small episode"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import litemkd_amd  # noqa: F401
from litemkd_amd.schedule import Schedule
import test_gpu_schedule as T
dev = torch.device("cuda", 0)
shot, img, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
over = dict(a.split("=") for a in sys.argv[4:])
over = {k: v == "1" for k, v in over.items()}
a = T._run(dev, Schedule.serial(), shot, img, n)
b = T._run(dev, Schedule.bench(**over), shot, img, n)
print("losses equal", torch.equal(a[0], b[0]), a[0].tolist(), b[0].tolist())
print("grad max rel diff", float((a[1] - b[1]).abs().max() / a[1].abs().max()))
for k in a[3]:
    d = (a[3][k] - b[3][k]).abs().max()
    if float(d) > 0:
        print(k, "max abs diff", float(d), "rel", float(d / a[3][k].abs().max()))
