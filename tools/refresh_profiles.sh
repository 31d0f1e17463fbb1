#!/bin/bash
# Re-measure every bench line kept under profiles/ (run on the GPU box; writes gpurun_out/refresh/).
# Usage: tools/refresh_profiles.sh
out=$GRAFT_REPO_ROOT/gpurun_out/refresh
mkdir -p $out
cd $GRAFT_REPO_ROOT
b() { name=$1; shift; timeout -k 10 280 python bench.py "$@" > $out/$name.log 2>&1; tail -1 $out/$name.log > $out/$name.json; echo "$name: $(python3 -c "import json,sys; d=json.load(open('$out/$name.json')); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"; }
b bench_configs1
b bench_configs1_pose --pose-grads --no-cpu-baseline
b bench_fgbg --workload configs1-fgbg --no-cpu-baseline
b bench_fgbg_pose --workload configs1-fgbg --pose-grads --no-cpu-baseline
b bench_configs2 --workload configs2 --no-cpu-baseline
b bench_T24 --log2-T 24 --no-cpu-baseline
b bench_T24_B16384 --log2-T 24 --rays 16384 --no-cpu-baseline
b bench_reference_default_T24_B16384_fgbg_pose --workload configs1-fgbg --log2-T 24 --rays 16384 --pose-grads --no-cpu-baseline
b bench_render --workload configs4-render --no-cpu-baseline
