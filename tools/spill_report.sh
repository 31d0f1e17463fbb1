#!/bin/bash
# VGPR spill counts of every kernel of the library's HIP sources (hipcc -S of each file with the Makefile's flags).
# Usage: tools/spill_report.sh [file.hip ...]   (from anywhere; default: the render kernels)
cd "$(dirname "$0")/../scanerf-scalable-bundle-adjusting-neural-radiance-fields-for-large-scale-scene-rendering_amd/csrc"
files=${@:-render.hip render_bwd_t16.hip render_bwd_h3.hip render_time.hip}
mkdir -p /tmp/isa
for f in $files; do
  extra=""
  case $f in render.hip) extra="-DH3_REGIONS=1";; render_time.hip) extra="-ffp-contract=off -DH3_REGIONS=1";; esac
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -fvisibility=hidden -std=c++17 -munsafe-fp-atomics -I../../include -I. $extra -S --cuda-device-only -o /tmp/isa/${f%.hip}.s $f 2>/dev/null
  python3 - /tmp/isa/${f%.hip}.s <<'PY'
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)', s):
    if int(m.group(2)): print(f"{int(m.group(2)):5d}  {m.group(1)[:110]}")
PY
done
