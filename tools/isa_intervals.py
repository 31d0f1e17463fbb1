#!/usr/bin/env python3
"""Static instruction mix of the -DT16_STAMPS build of the t16s backward between consecutive s_memtime stamps (program order):
vector / transcendental / matrix / LDS / vector-memory / scratch instructions and a lower bound of the vector-issue cycles
(4 per vector instruction, 8 per transcendental and per MFMA: MI355X_MICROARCH.md, 'vector-instruction ISSUE cost').

    tools/build_variant.py stamps render_bwd_t16="-DT16_STAMPS" && tools/isa_intervals.py [kernel-substring]
"""
import os, re, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "scanerf-scalable-bundle-adjusting-neural-radiance-fields-for-large-scale-scene-rendering_amd", "lib", "obj", "render_bwd_t16_stamps.o")
LLVM = "/opt/rocm/lib/llvm/bin"
want = sys.argv[1] if len(sys.argv) > 1 else "k_render_bwd_t16ILi0ELi2ELb0ELb1E"
TRANS = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_")
with tempfile.TemporaryDirectory() as tmp:
    dst = os.path.join(tmp, "u.o")
    shutil.copy(OBJ, dst)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", dst], check=True, capture_output=True)
    co = [os.path.join(tmp, f) for f in os.listdir(tmp) if f.startswith("u.o.") and "gfx950" in f][0]
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", co], check=True, capture_output=True, text=True).stdout
blk = [b for b in re.split(r"\n(?=<[^>]+>:\n)", dis) if re.match(r"<[^>]*" + re.escape(want), b)]
assert len(blk) == 1, [b[:80] for b in blk]
rows, cur = [], dict(valu=0, trans=0, mfma=0, lds=0, vmem=0, scratch=0, salu=0, wait=0, barrier=0, branch=0)
for l in blk[0].split("\n")[1:]:
    i = l.split("//")[0].strip()
    if not i or i.startswith("<"):
        continue
    op = i.split()[0]
    if op == "s_memtime":
        rows.append(cur)
        cur = dict.fromkeys(cur, 0)
        continue
    if op.startswith("v_mfma"): cur["mfma"] += 1
    elif TRANS.match(op): cur["trans"] += 1
    elif op.startswith("v_"): cur["valu"] += 1
    elif op.startswith("ds_"): cur["lds"] += 1
    elif op.startswith("scratch_"): cur["scratch"] += 1
    elif op.startswith(("global_", "buffer_", "flat_")): cur["vmem"] += 1
    elif op == "s_waitcnt": cur["wait"] += 1
    elif op == "s_barrier": cur["barrier"] += 1
    elif op.startswith(("s_cbranch", "s_branch")): cur["branch"] += 1
    elif op.startswith("s_"): cur["salu"] += 1
rows.append(cur)
print("seg   valu trans mfma  lds vmem scratch salu wait barrier branch   vector-issue cycles >=")
for k, r in enumerate(rows):
    cyc = 4 * r["valu"] + 8 * r["trans"] + 8 * r["mfma"]
    print(f"{k:3d} {r['valu']:6d} {r['trans']:5d} {r['mfma']:4d} {r['lds']:4d} {r['vmem']:4d} {r['scratch']:7d} {r['salu']:4d} {r['wait']:4d} {r['barrier']:7d} {r['branch']:6d}   {cyc:6d}")
