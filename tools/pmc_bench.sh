#!/bin/bash
# Extra SQ counter passes over the serialized benchmark (2 episodes): instruction mix, wait cycles, memory / LDS queue levels per conv
# kernel family.  usage (GPU box, repo root): tools/pmc_bench.sh [bench dtype]
dt=${1:-f32}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_bench_$dt
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -o r -- python3 $R/bench.py --dtype $dt --warmup 1 --steps 1 --no-cpu-baseline --no-other-modes --serial --roofline-episodes 0 > $out/g$i.log 2>&1
  echo "group $i done rc=$?"
done
cd $R
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$out/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "conv_" not in k: continue
        k = k.split("<")[0] + ("<" + k.split("<")[2].split(">")[0] + ">" if k.count("<") > 1 else "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open("$out/summary.txt", "w") as o:
    for k in sorted(agg):
        d = agg[k]
        o.write(k + "\n")
        for c in sorted(d): o.write("   %-28s %.4g\n" % (c, d[c]))
        wc = max(d["SQ_WAVE_CYCLES"], 1)
        o.write("   => per wave-cycle: wait_any %.2f  wait_inst_any %.2f | per MFMA: VALU %.2f LDS %.2f VMEM_RD %.3f SALU %.2f | avg VMEM latency %.0f, LDS latency %.0f (level / insts)\n" % (
            d["SQ_WAIT_ANY"] / wc, d["SQ_WAIT_INST_ANY"] / wc, d["SQ_INSTS_VALU"] / max(d["SQ_INSTS_MFMA"], 1), d["SQ_INSTS_LDS"] / max(d["SQ_INSTS_MFMA"], 1),
            d["SQ_INSTS_VMEM_RD"] / max(d["SQ_INSTS_MFMA"], 1), d["SQ_INSTS_SALU"] / max(d["SQ_INSTS_MFMA"], 1),
            d["SQ_INST_LEVEL_VMEM"] / max(d["SQ_INSTS_VMEM_RD"], 1), d["SQ_INST_LEVEL_LDS"] / max(d["SQ_INSTS_LDS"], 1)))
print(open("$out/summary.txt").read())
PY
rm -rf $out/g*/
