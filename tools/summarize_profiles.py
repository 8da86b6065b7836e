"""Turns the rocprofv3 outputs of tools/profile_round.sh (gpurun_out/prof_<tag>/) into the summaries committed under profiles/:
  <tag>_kernel_stats_serial_<mode>.csv   per-kernel statistics of `bench.py --serial` in each arithmetic mode (+ _kernel_time_per_episode_<mode>.txt)
  <tag>_conv_by_shape.txt                conv launches by kernel instance and grid (headline mode)
  <tag>_hbm_traffic.json                 FETCH_SIZE x2 + WRITE_SIZE per launch and kernel family, with the hash of the kernel
                                         sources it was measured on (bench.py reports `roofline.traffic` only when it matches)
  <tag>_mfma_util.txt                    MFMA / LDS / VALU occupancy of the conv kernels from the SQ counters"""
import collections, csv, glob, hashlib, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
base = "gpurun_out/prof_%s" % tag
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "lite-mkd_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def family(n):
    for k in ("conv_patch_x3_kernel", "conv_stem_patch_kernel", "conv_gemm_x3_kernel", "conv_wgrad_win_kernel", "conv_wgrad_x3_kernel", "conv_gemm_kernel", "conv_wgrad_kernel"):
        if k in n:
            return k
    return n.replace("void ", "").split("(")[0].split("<")[0][:48]


def find(d, pat):
    fs = sorted(glob.glob("%s/%s/**/*%s" % (base, d, pat), recursive=True))
    return fs[-1] if fs else None


def load_pmc(d, counter):
    f = find(d, "counter_collection.csv")
    agg = collections.defaultdict(lambda: [0.0, 0])
    if f is None:
        return agg
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        fam = family(r["Kernel_Name"])
        agg[fam][0] += float(r["Counter_Value"])
        agg[fam][1] += 1
    return agg


os.makedirs("profiles", exist_ok=True)
for mode in ("f32", "f32x3", "f32native", "bf16"):
    f = find("trace_" + mode, "kernel_stats.csv")
    if f is None:
        continue
    shutil.copy(f, "profiles/%s_kernel_stats_serial_%s.csv" % (tag, mode))
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    fam = collections.defaultdict(float)
    for r in rows:
        fam[family(r["Name"])] += float(r["TotalDurationNs"])
    neps = max([int(r["Calls"]) for r in rows if "d2m_loss_kernel" in r["Name"]] + [1])      # one loss kernel per episode (warm-up + timed + roofline pass)
    nl = sum(int(r["Calls"]) for r in rows)
    lines = ["%s: %d serialized episodes in the trace; sum of kernel time per episode %.2f ms, %.0f launches per episode" % (mode, neps, tot / 1e6 / neps, nl / neps)]
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:24]:
        lines.append("    %-40s %7.3f ms/episode  %5.1f %%" % (k, v / 1e6 / neps, 100 * v / tot))
    print("\n".join(lines[:14]))
    with open("profiles/%s_kernel_time_per_episode_%s.txt" % (tag, mode), "w") as fo:
        fo.write("# tools/summarize_profiles.py from profiles/%s_kernel_stats_serial_%s.csv (rocprofv3 --kernel-trace --stats of bench.py --serial)\n" % (tag, mode))
        fo.write("\n".join(lines) + "\n")

tr = find("trace_f32", "kernel_trace.csv")
if tr:
    g = collections.defaultdict(list)
    for r in csv.DictReader(open(tr)):
        n = r["Kernel_Name"]
        if "conv_gemm" in n or "conv_wgrad" in n or "conv_patch" in n or "conv_stem" in n:
            key = (n.replace("void ", "").split("(")[0], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["LDS_Block_Size"], r["VGPR_Count"])
            g[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open("profiles/%s_conv_by_shape.txt" % tag, "w") as fo:
        fo.write("# rocprofv3 --kernel-trace --stats of `python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-modes --serial` (9 episodes with\n"
                 "# the roofline pass, headline arithmetic), conv kernels by instance and grid\n# kernel | workgroups | LDS bytes | VGPRs | launches | avg us | total ms\n")
        for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
            fo.write("%s | %s | %s | %s | %d | %.1f | %.2f\n" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))

F, W = load_pmc("pmc_fetch", "FETCH_SIZE"), load_pmc("pmc_write", "WRITE_SIZE")
out = {"_how": "tools/profile_round.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python3 bench.py --steps 2 "
               "--warmup 1 --no-cpu-baseline --no-other-modes --serial --roofline-episodes 0 (3 serialized episodes, headline arithmetic); counters in "
               "KiB; FETCH_SIZE doubled (gfx950 tallies the 128-B requests of 16-B/lane reads as 64 B, MI355X_MICROARCH.md HBM section); "
               "Infinity-Cache hits are included in FETCH_SIZE",
       "kernel_src_hash": kernel_source_hash()}
for fam in sorted(set(F) | set(W), key=lambda k: -(F[k][0] + W[k][0])):
    f, nf = F[fam]
    w, nw = W[fam]
    if nf == 0 and nw == 0:
        continue
    fetch = 2 * f * 1024 / max(nf, 1)
    write = w * 1024 / max(nw, 1)
    out[fam] = {"launches": nf or nw, "fetch_bytes_per_launch_corrected": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write}
# round 5: the two-plane launches of layer 1 run conv_patch_x3_kernel<.., 3, ..> (32x32x16), the others conv_patch16_x3_kernel: bench.py's
# dominant family (all LDS-patch launches of an episode) is the launch-weighted mean of the two
PK = [k for k in ("conv_patch16_x3_kernel", "conv_patch_x3_kernel") if k in out]
if len(PK) == 2:
    n = sum(out[k]["launches"] for k in PK)
    out["conv_patch16_x3_kernel+conv_patch_x3_kernel"] = {
        "launches": n, "kernels": PK,
        "fetch_bytes_per_launch_corrected": sum(out[k]["fetch_bytes_per_launch_corrected"] * out[k]["launches"] for k in PK) / n,
        "write_bytes_per_launch": sum(out[k]["write_bytes_per_launch"] * out[k]["launches"] for k in PK) / n,
        "traffic_bytes_per_launch": sum(out[k]["traffic_bytes_per_launch"] * out[k]["launches"] for k in PK) / n}
if len(out) > 2:
    json.dump(out, open("profiles/%s_hbm_traffic.json" % tag, "w"), indent=1)
    for fam in list(out)[2:10]:
        print("traffic %-28s launches %4d  fetch %.1f MB  write %.1f MB per launch" % (fam, out[fam]["launches"], out[fam]["fetch_bytes_per_launch_corrected"] / 1e6,
                                                                                  out[fam]["write_bytes_per_launch"] / 1e6))

agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for f in glob.glob(base + "/pmc_sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        fam = family(r["Kernel_Name"])
        if not fam.startswith("conv_"):
            continue
        agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CYCLES":
            n[fam] += 1
if agg:
    with open("profiles/%s_mfma_util.txt" % tag, "w") as fo:
        fo.write("# tools/profile_round.sh: SQ counters of the conv kernels inside the serialized benchmark (3 episodes, headline arithmetic),\n"
                 "# summed over all launches of a family.  SQ_BUSY_CYCLES is summed over 32 shader engines: kernel cycles = SQ_BUSY_CYCLES / 32;\n"
                 "# MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs)\n")
        for fam, d in agg.items():
            cyc = d["SQ_BUSY_CYCLES"] / 32
            fo.write("%s  launches %d\n" % (fam, n[fam]))
            for c, v in sorted(d.items()):
                fo.write("   %-28s %.4g\n" % (c, v))
            if cyc > 0:
                line = "   => MFMA utilisation %.1f %%   LDS busy %.1f %% of CU cycles (bank conflicts %.1f %% of LDS cycles)   VALU issue %.1f %% of SIMD cycles\n" % (
                    100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 100 * d["SQ_LDS_IDX_ACTIVE"] / (cyc * 256),
                    100 * d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1), 100 * d["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024))
                fo.write(line)
                print(fam, line.strip())
