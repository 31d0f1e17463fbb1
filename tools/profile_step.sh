#!/bin/bash
# Kernel-time summary and HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the default bench step.
# Usage (on the GPU box): tools/profile_step.sh <outdir under gpurun_out> [bench args]
out=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-legs "$@" > $out/stats.log 2>&1 || echo "stats pass failed"
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py stats $out/stats > $out/kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-side-legs "$@" > $out/$c.log 2>&1 || echo "$c pass failed"
  python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py pmc $out/$c > $out/$c.txt 2>&1
done
find $out -name "*.csv" -size +1M -delete
cat $out/kernel_stats.txt; cat $out/FETCH_SIZE.txt $out/WRITE_SIZE.txt | head -40
