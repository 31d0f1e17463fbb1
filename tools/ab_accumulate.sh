#!/bin/bash
# One-box A/B asked for by the round-4 review (item 3): does the skewed emission of the t16s backward (waves 4-7 park a tile's dX and
# emit behind the next tile's compositing: round 4) change what the accumulate costs?  SCANERF_BWD_PARK=0 restores round 3's emission
# order (every wave at its tile's end); the accumulate kernel's code is the same in both (r03 -> r04 diff of csrc/scatter.hip: one
# conditional instruction at its start).  Three runs each, interleaved; prints ms per step and the live per-kernel averages.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for park in 1 0; do
    SCANERF_BWD_PARK=$park python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-side-legs --arith-side-off 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['roofline']['kernels']
print('park=$park rep=$rep: step %.3f ms  forward %.3f  backward %.3f  accumulate+adam %.3f' % (l['ms_per_step'], k['render_forward']['avg_launch_ms'], k['render_backward']['avg_launch_ms'], k['table_grad_accumulate_adam']['avg_launch_ms']))"
  done
done
