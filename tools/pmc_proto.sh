#!/bin/bash
# PMC counters of tools/proto_bench.py's launches (3x3 layers, fp32x3 and fp32h2): gpurun_out/pmc_proto.txt
set -e
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_proto
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -o r -- python3 $R/tools/proto_bench.py ${1:-400} > $out/p$i.log 2>&1
  echo "pass $i done"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_proto/p*/**/r_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "patch16" not in n and "win16" not in n and "conv_patch_x3" not in n:
            continue
        key = (n.split("(")[0][:70], r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_proto.txt", "w") as w:
    for key in sorted(agg):
        c = {k: sum(v) / len(v) for k, v in agg[key].items()}
        cyc = c.get("SQ_BUSY_CYCLES", 0) / 32      # per-XCD-SE aggregate -> kernel cycles (DESIGN 9.x normalisation)
        w.write("%s grid %s\n" % key)
        w.write("   kernel cycles %.0f   MFMA busy %.1f %%   insts: mfma %.0f valu %.0f lds %.0f\n" % (
            cyc, 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(cyc * 1024, 1), c.get("SQ_INSTS_MFMA", 0), c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_LDS", 0)))
        wc = max(c.get("SQ_WAVE_CYCLES", 1), 1)
        w.write("   of wave cycles: waiting (waitcnt/barrier) %.1f %%  issue stall %.1f %%  active %.1f %%  | VALU active %.1f %%  LDS active %.1f %% (bank conflict %.1f %%)\n" % (
            100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * c.get("SQ_LDS_IDX_ACTIVE", 0) / wc, 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / wc))
PY
