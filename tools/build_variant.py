#!/usr/bin/env python3
"""Investigation builds of libscanerf_hip.so: chosen translation units recompiled with other flags, linked with the
standard objects of the rest into <pkg>/lib/debug/libscanerf_hip_<tag>.so (select with SCANERF_LIB=...; never loaded by
the product).

    tools/build_variant.py <tag> unit="flags" [unit="flags" ...]
    e.g. tools/build_variant.py slp_guarded render="-DH3_OPAQUE_ADDR=1 -DSCANERF_GUARDS=1 -DH3_REGIONS=1"   (rounds 1-2's form)
         tools/build_variant.py slp_bare render="-DH3_OPAQUE_ADDR=1"                                       (differs in every launch)

`flags` replace the unit's EXTRA of csrc/Makefile (the common flags stay)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "scanerf-scalable-bundle-adjusting-neural-radiance-fields-for-large-scale-scene-rendering_amd")
CSRC, OBJ, DBG = os.path.join(PKG, "csrc"), os.path.join(PKG, "lib", "obj"), os.path.join(PKG, "lib", "debug")
COMMON = ("-O3 --offload-arch=gfx950 -fPIC -fvisibility=hidden -std=c++17 -munsafe-fp-atomics -Wall -I../../include -I. "
          "-fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops").split()   # (csrc/Makefile COMMON; a unit's flags may undo them)
UNITS = "rays adam render_time voxelize api hashgrid render scatter render_bwd render_bwd_h3 render_bwd_t16 h3_selftest loss compact decoder composite".split()
tag, specs = sys.argv[1], dict(a.split("=", 1) for a in sys.argv[2:])
assert all(u in UNITS for u in specs), specs
os.makedirs(DBG, exist_ok=True)
procs, objs = [], []
for u in UNITS:
    if u in specs:
        o = os.path.join(OBJ, f"{u}_{tag}.o")
        procs.append((u, subprocess.Popen(["/opt/rocm/bin/hipcc"] + COMMON + specs[u].split() + ["-c", u + ".hip", "-o", o], cwd=CSRC)))
    else:
        o = os.path.join(OBJ, u + ".o")
    objs.append(o)
for u, p in procs:
    if p.wait():
        raise SystemExit(f"build_variant: {u} failed")
out = os.path.join(DBG, f"libscanerf_hip_{tag}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
