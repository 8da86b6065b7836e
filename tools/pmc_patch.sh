#!/bin/bash
# PMC passes over tools/patch_bench.py (one mode): MFMA / LDS / VALU / memory-side counters of the patch and gather kernels.
# usage (GPU box, repo root): tools/pmc_patch.sh fp32x3
mode=${1:-fp32x3}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_patch_$mode
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -o r -- python3 $R/tools/patch_bench.py $mode > $out/g$i.log 2>&1
  echo "group $i done rc=$?"
done
cd $R
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob("$out/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "conv_" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open("$out/summary.txt", "w") as o:
    for k in sorted(agg):
        o.write(k + "\n")
        for c in sorted(agg[k]): o.write("   %-32s %.4g\n" % (c, agg[k][c]))
print(open("$out/summary.txt").read())
PY
rm -rf $out/g*/
