#!/bin/bash
# Round profile set, run on the GPU box from the repo root:  tools/profile_round.sh r02
# Writes under gpurun_out/prof_<tag>/ : kernel trace + stats of the serialized benchmark in the headline arithmetic and in the native
# fp32 MFMA arithmetic, two separate PMC passes for HBM traffic (FETCH_SIZE / WRITE_SIZE) and two for MFMA / LDS / VALU occupancy.
# Counters are collected with --kernel-trace only (no other trace domains); the program after `--` is python itself.
set -e
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --warmup 1 --no-cpu-baseline --no-other-modes --serial --prime 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_f32 -o r -- python3 $B --steps 4 > $out/trace_f32.log 2>&1
echo "trace f32 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_f32x3 -o r -- python3 $B --steps 4 --dtype f32x3 > $out/trace_f32x3.log 2>&1
echo "trace f32x3 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_f32native -o r -- python3 $B --steps 4 --dtype f32native > $out/trace_f32native.log 2>&1
echo "trace f32native done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_bf16 -o r -- python3 $B --steps 4 --dtype bf16 > $out/trace_bf16.log 2>&1
echo "trace bf16 done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o r -- python3 $B --steps 2 --roofline-episodes 0 > $out/pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o r -- python3 $B --steps 2 --roofline-episodes 0 > $out/pmc_write.log 2>&1
echo "pmc write done"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/pmc_sq$i -o r -- python3 $B --steps 2 --roofline-episodes 0 > $out/pmc_sq$i.log 2>&1
  echo "pmc sq$i done"
done
cd $R
python3 tools/summarize_profiles.py $tag
