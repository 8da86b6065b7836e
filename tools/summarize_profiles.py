"""Turns the rocprofv3 outputs merged into gpurun_out/ (prof_r1c = --kernel-trace --stats of `bench.py --serial`,
pmc_fetch / pmc_write = the two PMC passes) into the summaries committed under profiles/."""
import collections, csv, glob, json, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_v4"
def load(d, counter):
    f = sorted(glob.glob('gpurun_out/%s/*/*_counter_collection.csv' % d))[-1]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        n = r['Kernel_Name']
        fam = 'conv_gemm_kernel' if 'conv_gemm' in n else ('conv_wgrad_kernel' if 'conv_wgrad' in n else n.split('(')[0][:40])
        agg[fam][0] += float(r['Counter_Value']); agg[fam][1] += 1
    return agg
F = load('pmc_fetch', 'FETCH_SIZE'); W = load('pmc_write', 'WRITE_SIZE')
out = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python bench.py --steps 2 --warmup 1 "
               "--no-cpu-baseline --serial --roofline-episodes 0; counters in KiB; FETCH_SIZE doubled (gfx950 counts the 128-B requests of "
               "16-B/lane reads as 64 B, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are included in FETCH_SIZE"}
for fam in ('conv_gemm_kernel', 'conv_wgrad_kernel', 'bn_bwd_apply_kernel', 'bn_bwd_reduce_kernel', 'bn_apply_kernel', 'wgrad_reduce_kernel'):
    f, nf = F[fam]; w, nw = W[fam]
    fetch = 2 * f * 1024 / max(nf, 1); write = w * 1024 / max(nw, 1)
    out[fam] = {"launches": nf, "fetch_bytes_per_launch_corrected": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write}
    print(fam, nf, "fetch %.1f MB write %.1f MB" % (fetch / 1e6, write / 1e6))
json.dump(out, open('profiles/r01_hbm_traffic.json', 'w'), indent=1)
f = sorted(glob.glob('gpurun_out/prof_r1c/*/*_kernel_stats.csv'))[-1]
shutil.copy(f, 'profiles/%s_kernel_stats_serial.csv' % tag)
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("sum of kernel time per episode (5 episodes): %.2f ms" % (tot / 1e6 / 5))
tr = sorted(glob.glob('gpurun_out/prof_r1c/*/*_kernel_trace.csv'))[-1]
g = collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    n = r['Kernel_Name']
    if 'conv_gemm' in n or 'conv_wgrad' in n:
        key = (n.replace('void ', ''), int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z'], r['LDS_Block_Size'], r['VGPR_Count'])
        g[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
with open('profiles/%s_conv_by_shape.txt' % tag, 'w') as fo:
    fo.write("# rocprofv3 --kernel-trace --stats of `python bench.py --steps 4 --warmup 1 --no-cpu-baseline --serial` (5 episodes), conv kernels by launch shape\n"
             "# kernel | workgroups x,y,z | LDS bytes | VGPRs | launches | avg us | total ms\n")
    for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
        fo.write("%s | %s,%s,%s | %s | %s | %d | %.1f | %.2f\n" % (k[0], k[1], k[2], k[3], k[4], k[5], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
