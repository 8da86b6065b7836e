#!/bin/bash
# PMC counters of the stem weight-gradient kernels (GPU box, repo root): tools/pmc_stem.sh
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/pmc_stem; mkdir -p $out; cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -o r -- python3 $R/tools/stem_wgrad_bench.py 40 > $out/p$i.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob("gpurun_out/pmc_stem/p*/r_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0][:40]
        if "wgrad" not in n or "reduce" in n: continue
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(n, r["Counter_Name"])] += 1
    for n, d in agg.items():
        print(n, {k: round(v / cnt[(n, k)]) for k, v in d.items()})
PY
