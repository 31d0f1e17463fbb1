#!/bin/bash
# A/B matrix of the T = 2^24 default iteration under timing switches (EXP=1 build). Usage (GPU box): tools/t24_ab.sh <outdir> VAR=VAL[,VAR=VAL] ...
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
for cfg in "$@"; do
  (
    IFS=','; for kv in $cfg; do [ "$kv" != "base" ] && export "$kv"; done
    python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --no-cpu-baseline --no-side-legs --steps 10 --warmup 2 --workload configs1-fgbg --log2-T 24 --rays 16384 --pose-grads 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$cfg', round(d['ms_per_step'],3), {a:round(b['avg_launch_ms'],3) for a,b in k.items()})"
  ) | tee -a $out/ab.txt
done
