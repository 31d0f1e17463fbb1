#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py command -> gpurun_out/<outdir>/kernel_stats.txt.  Usage (GPU box): tools/stats_only.sh <outdir> [bench args]
out=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --no-cpu-baseline --no-side-legs --steps 10 --warmup 2 "$@" > $out/stats.log 2>&1 || echo "stats pass failed"
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py stats $out/stats > $out/kernel_stats.txt 2>&1
find $out -name "*.csv" -size +1M -delete
head -14 $out/kernel_stats.txt
