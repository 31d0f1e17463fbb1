# Is the accumulate's time the tail of the few heavy bins of the dense coarse levels (level 0: 3.4e7 records in 5 bins)?  Experiments
# build (EXP=1 tools/rebuild.sh): SCANERF_ACC_DBG bits 8..15 = N: bins of levels below N do nothing; bit 16: the others do nothing.
# Round 6, one box: all 1.546 ms | without level 0: 1.382 | without 0-1: 1.303 | without 0-3: 1.049 | ONLY 0-3: 0.844 | ONLY 0-1: 0.82 | ONLY 0: 0.82
for v in 0 256 512 1024 66560 66048 65792; do echo -n "SCANERF_ACC_DBG=$v: "; SCANERF_ACC_DBG=$v python bench.py --no-cpu-baseline --no-side-legs --arith-side-off --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(round(d['ms_per_step'],3), {n: round(v.get('avg_launch_ms',0),3) for n,v in k.items() if 'accum' in n or 'adam' in n})"; done
