#!/bin/bash
# MFMA / LDS / wait counters of the conv kernels inside the real benchmark (serialized episodes).  Run on the GPU box from the
# repo root: tools/pmc_bench.sh OUTNAME [bench args].  Separate rocprofv3 --pmc passes, nothing but --kernel-trace next to them.
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial --roofline-episodes 0 "$@" > $out/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - "$out" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        fam = "conv_gemm_kernel" if "conv_gemm_kernel" in k else ("conv_gemm_x3_kernel" if "conv_gemm_x3" in k else ("conv_wgrad_kernel" if "conv_wgrad" in k else None))
        if fam is None: continue
        agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_BUSY_CYCLES", "SQ_LDS_IDX_ACTIVE"): n[(fam, r["Counter_Name"])] += 1
print("# per kernel family, summed over all launches of 3 serialized episodes (bench.py --serial); SQ_BUSY_CYCLES is summed over 32 shader engines,")
print("# so kernel cycles = SQ_BUSY_CYCLES / 32 and MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs)")
for fam, d in agg.items():
    cyc = d["SQ_BUSY_CYCLES"] / 32
    print(fam, " launches", n[(fam, "SQ_BUSY_CYCLES")])
    for c, v in sorted(d.items()): print("   %-28s %.4g" % (c, v))
    if cyc > 0:
        print("   => MFMA utilisation %.1f %%   LDS busy %.1f %% of CU cycles (bank conflicts %.1f %% of LDS cycles)   VALU issue %.1f %% of SIMD cycles" % (
            100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 100 * d["SQ_LDS_IDX_ACTIVE"] / (cyc * 256),
            100 * d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1), 100 * d["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024)))
PY
