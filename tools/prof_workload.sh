#!/bin/bash
# rocprofv3 kernel stats of one bench workload, per step.  Usage (on the GPU box): tools/prof_workload.sh <outdir under gpurun_out> <bench args...>
out=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $out/stats.log 2>&1 || echo "profiling failed"
python3 $GRAFT_REPO_ROOT/tools/kernels_per_step.py $out/stats 12 30
find $out -name "*.csv" -size +1M -delete
