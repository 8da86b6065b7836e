import os, sys, time, random
sys.path.insert(0, "/root/repo")
import torch
import litemkd_amd
from litemkd_amd.video_transform import GpuFrameTransform
dev = torch.device("cuda", 0)
tf = GpuFrameTransform(224, dev)
u8 = torch.randint(0, 256, (400, 240, 320, 3), dtype=torch.uint8, device=dev)
params = [tf.draw(240, 320, True) for _ in range(50)]
out = torch.empty((400, 224, 224, 4), device=dev)
for _ in range(3): tf.batch(u8, params, 8, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(10): tf.batch(u8, params, 8, out=out)
e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
print("transform: gpu %.3f ms, host enqueue %.3f ms per episode" % (e0.elapsed_time(e1) / 10, (t1 - t0) * 100))
h = torch.randint(0, 256, (400, 240, 320, 3), dtype=torch.uint8).pin_memory()
for _ in range(2): u8.copy_(h, non_blocking=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): u8.copy_(h, non_blocking=True)
e1.record(); torch.cuda.synchronize()
print("H2D 92 MB: %.3f ms" % (e0.elapsed_time(e1) / 10))
