#!/bin/bash
# The measurement set committed under profiles/ at the end of a round (GPU box, repo root):  tools/final_round.sh r03
# bench lines (default incl. other modes + cpu baseline + streamed inputs, bf16, 1-shot, ResNet-50 + live MFM, 2 cores, world-8 cadence,
# hipGraph, pipelining), inference, layer tables, the numerical and ablation probes, host-side probes.  Everything lands in gpurun_out/final_<tag>/.
# (6 s between processes: a process started while the previous one's memory is still being released runs slow - DESIGN 8.8)
tag=${1:-r03}; part=${2:-all}; out=gpurun_out/final_$tag; mkdir -p $out      # part: 1 | 2 | 3 | all (a gpurun call is capped at 20 minutes)
B="python bench.py"
[ "$part" = all -o "$part" = 1 ] && { sleep 6; $B --steps 20 --warmup 5 --stream-inputs --layer-table > $out/bench.json 2> $out/layer_table.txt; echo "bench $?"; }
[ "$part" = all -o "$part" = 1 ] && { sleep 6; $B --steps 32 --warmup 5 --no-cpu-baseline --no-other-modes --dtype bf16 --layer-table > $out/bench_bf16.json 2> $out/layer_table_bf16.txt; echo "bf16 $?"; }
[ "$part" = all -o "$part" = 1 ] && { sleep 6; $B --steps 32 --warmup 5 --no-cpu-baseline --no-other-modes --dtype f32x3 --layer-table > $out/bench_f32x3.json 2> $out/layer_table_f32x3.txt; echo "f32x3 $?"; }
[ "$part" = all -o "$part" = 1 ] && { sleep 6; $B --steps 32 --warmup 5 --no-cpu-baseline --no-other-modes --dtype f32native --layer-table > $out/bench_f32native.json 2> $out/layer_table_f32native.txt; echo "native $?"; }
[ "$part" = all -o "$part" = 1 ] && { sleep 6; $B --steps 32 --warmup 5 --no-other-modes --shot 1 --cpu-episodes 1 > $out/bench_shot1.json 2>/dev/null; echo "shot1 $?"; }
[ "$part" = all -o "$part" = 1 ] && { sleep 6; $B --steps 8 --warmup 3 --no-cpu-baseline --no-other-modes --backbone resnet50_2fc --live-mfm > $out/bench_r50_mfm.json 2>/dev/null; echo "r50 $?"; }
[ "$part" = all -o "$part" = 1 ] && { sleep 6; taskset -c 0,1 $B --steps 32 --warmup 5 --no-cpu-baseline --no-other-modes > $out/bench_2cores.json 2>/dev/null; echo "2cores $?"; }
[ "$part" = all -o "$part" = 1 ] && { sleep 6; taskset -c 0,1 $B --steps 32 --warmup 5 --no-cpu-baseline --no-other-modes --emulate-world 8 > $out/bench_world8_cadence_2cores.json 2>/dev/null; echo "w8 $?"; }
[ "$part" = all -o "$part" = 1 ] && { sleep 6; taskset -c 0,1 $B --steps 32 --warmup 5 --no-cpu-baseline --no-other-modes --dtype bf16 > $out/bench_bf16_2cores.json 2>/dev/null; echo "bf16 2cores $?"; }
[ "$part" = all -o "$part" = 2 ] && { sleep 6; $B --steps 32 --warmup 5 --no-cpu-baseline --no-other-modes --graph > $out/bench_graph.json 2>/dev/null; echo "graph $?"; }
[ "$part" = all -o "$part" = 2 ] && { sleep 6; $B --steps 32 --warmup 5 --no-cpu-baseline --no-other-modes --graph --dtype bf16 > $out/bench_graph_bf16.json 2>/dev/null; echo "graph bf16 $?"; }
[ "$part" = all -o "$part" = 2 ] && { sleep 6; $B --steps 32 --warmup 5 --no-cpu-baseline --no-other-modes --pipeline > $out/bench_pipeline.json 2>/dev/null; echo "pipeline $?"; }
[ "$part" = all -o "$part" = 2 ] && { sleep 6; python tools/eval_bench.py 2>/dev/null | tail -1 > $out/eval.json; echo "eval $?"; }
[ "$part" = all -o "$part" = 2 ] && { sleep 6; python tools/x3_bias_probe.py 40 2>/dev/null > $out/x3_bias_probe.txt; echo "bias $?"; }
[ "$part" = all -o "$part" = 2 ] && { sleep 6; python tools/abl_bench.py 2>/dev/null | grep "^L" > $out/patch32_vs_16.txt; echo "abl $?"; }
[ "$part" = all -o "$part" = 2 ] && { sleep 6; python tools/host_ahead.py 10 2>/dev/null > $out/host_ahead.txt; echo "host $?"; }
[ "$part" = all -o "$part" = 2 ] && { sleep 6; python tools/aten_ops.py f32 2 2>/dev/null > $out/launches_f32.txt; python tools/aten_ops.py f32x3 2 2>/dev/null > $out/launches_f32x3.txt; echo "launches $?"; }
[ "$part" = all -o "$part" = 2 ] && { sleep 6; python tools/phase_times.py 14 2>/dev/null | tail -1 > $out/phase_times.txt; echo "phases $?"; }
[ "$part" = all -o "$part" = 3 ] && { sleep 6; python tools/stem_wgrad_bench.py 2>/dev/null | tail -2 > $out/stem_wgrad_bench.txt; echo "stem wgrad $?"; }
[ "$part" = all -o "$part" = 3 ] && { sleep 6; python tools/gemm_splitk_bench.py 2>/dev/null | tail -7 > $out/gemm_splitk_bench.txt; echo "splitk $?"; }
[ "$part" = all -o "$part" = 3 ] && { sleep 6; LMKD_DGRAD_BN=0 $B --steps 32 --warmup 5 --no-cpu-baseline --no-other-modes > $out/bench_no_dgrad_bn_sums.json 2>/dev/null; echo "no dgrad bn $?"; }
[ "$part" = all -o "$part" = 3 ] && { sleep 6; $B --steps 32 --warmup 5 --no-cpu-baseline --no-other-modes > $out/bench_repeat.json 2>/dev/null; echo "repeat $?"; }
[ "$part" = all -o "$part" = 3 ] && { sleep 6; python tools/h2_error.py 40 2>/dev/null > $out/h2_error.txt; echo "h2 error $?"; }
[ "$part" = all -o "$part" = 3 ] && { sleep 6; python tools/proto_bench.py 400 2>/dev/null > $out/h2_layers_400.txt; echo "h2 layers $?"; }
[ "$part" = all -o "$part" = 3 ] && { sleep 6; python tools/h2_episode.py 3 serial 2>/dev/null > $out/h2_episode.txt; echo "h2 episode $?"; }
