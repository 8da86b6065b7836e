"""Register / scratch / occupancy table of the convolution kernels (hipcc -Rpass-analysis=kernel-resource-usage on csrc/conv.hip; no GPU needed).
A hot instance that spills (scratch beyond the 32 - 64 bytes of the tap tables' indexing) shows here: the 128 x 128 two-plane patch instance
did once (256 registers + 224 bytes: layer 2 went from 240 to 290 us).  usage: python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c",
                        os.path.join(root, "lite-mkd_amd", "csrc", "conv.hip"), "-o", os.path.join(d, "conv.o"), "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
txt = r.stderr
rows = []
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split(" ")[0]
    if not any(k in name for k in ("patch16", "conv_patch_x3", "wgrad_win16", "stem_wgrad", "conv_stem_patch", "conv_wgrad_x3")):
        continue
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("void ", "").split("(")[0]
    rows.append((dem, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
print("# kernel instance | VGPRs | AGPRs | scratch bytes/lane | waves/SIMD | static LDS bytes")
for r_ in sorted(rows):
    print("%-78s %4d %4d %5d %3d %6d" % r_)
bad = [r_ for r_ in rows if r_[3] > 64]
print("# instances with more than 64 bytes of scratch: %d" % len(bad))
sys.exit(1 if bad else 0)
