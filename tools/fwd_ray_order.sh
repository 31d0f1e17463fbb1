#!/bin/bash
# Round-5 forward experiment: do rays ordered by the cell they start in (and their direction) raise the L2 hit
# rate of k_render_fwd_h3?  Time (live) and TCC_HIT / TCC_MISS / TCC_EA0_RDREQ per launch for SORT = 0 (random order), 3, 5 (Morton
# order of 8^3 / 32^3 origin cells) and SORT_DIR = 1 (then by direction octant + dominant axis).
# Usage (on the GPU box): tools/fwd_ray_order.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
: > $out/summary.txt
for cfg in "0 0" "3 0" "5 0" "3 1" "5 1"; do
  set -- $cfg
  echo "== SORT=$1 SORT_DIR=$2" >> $out/summary.txt
  SORT=$1 SORT_DIR=$2 ARITH=h3 python3 $GRAFT_REPO_ROOT/tools/bench_fwd.py 2>/dev/null | grep "torch.float32 xstash=True" >> $out/summary.txt
  SORT=$1 SORT_DIR=$2 ARITH=h3 timeout -k 10 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $out/s$1_$2 -- python3 $GRAFT_REPO_ROOT/tools/bench_fwd.py > $out/s$1_$2.log 2>&1 || echo "pmc pass failed" >> $out/summary.txt
  python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py pmc $out/s$1_$2 2>/dev/null | grep "k_render_fwd_h3<0" >> $out/summary.txt
  find $out/s$1_$2 -name "*.csv" -size +2M -delete
done
cat $out/summary.txt
