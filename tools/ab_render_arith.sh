#!/bin/bash
# Same-box A/B of the render-time decode kernels: frame time of configs[4]'s render leg per --infer-arith value (t16 / h3 / f32).
# Usage: tools/ab_render_arith.sh <outdir under gpurun_out> <arith> [<arith> ...]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
for ar in "$@"; do
  timeout -k 10 200 python bench.py --infer-arith $ar --workload configs4-render --steps 5 --warmup 2 --no-cpu-baseline --no-side-legs > $out/render_$ar.json 2> $out/render_$ar.err || echo "$ar failed"
  python - "$out/render_$ar.json" "$ar" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "ms_per_frame", round(d["ms_per_step"],2))
except Exception as e: print(sys.argv[2], "no line", e)
PY
done
