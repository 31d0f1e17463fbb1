D=$(ls -d scanerf-*/lib/debug)
bash tools/ab_bwd.sh plainhead fasthead fasthead2
for rep in 1 2; do for tag in plainhead fastfwd; do echo -n "$tag: "; SCANERF_LIB=$D/libscanerf_hip_$tag.so python bench.py --no-cpu-baseline --no-side-legs --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'].get('kernels_live_ms', d['roofline'].get('achieved')))"; done; done
