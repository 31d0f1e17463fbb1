"""Which per-CU state does the withdrawn forward build's wrong encoder output depend on, and what IS the wrong value?
(DESIGN.md 4.10.  Run with SCANERF_LIB=<pkg>/lib/debug/libscanerf_hip_<tag>.so -- built by
`tools/build_variant.py <tag> render="-DH3_OPAQUE_ADDR=1 -fslp-vectorize"` -- for the withdrawn build, without it for the shipped
one.  Needs tools/probe/libstate_poison.so: built here on first use, `hipcc --offload-arch=gfx950 -shared -fPIC`.)

Back-to-back launches of the training forward (plan counts, x-stash, ray mask: k_render_fwd_h3<F32, COUNT>) on the same
inputs; between two launches ONE of:
    none      nothing
    evict     ~30 different torch kernels (round 2's EVICT=1: the only thing that brought the fault back)
    icache    tools/probe/state_poison.hip icache_sweep: 300 KB of straight-line code, no LDS, 3 registers
    poisonP   poison_regs_lds(1234.5f): every VGPR of every SIMD and all LDS of every CU set to 1234.5, small code
    poison0   the same with 0
    both      icache + poisonP
For every differing x-stash element the 8 (weight, corner value) pairs of its (sample, level) are recomputed on the host and the
value that, put in place of ONE corner, explains the wrong output is printed:  stale_c = f_c + (wrong - right) / w_c,  next to
1234.5 (the poison), 0 and the float whose bits are the low dword of that corner's address (the load's destination register
holds its own address until the data lands).
"""
import ctypes, os, struct, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scanerf_amd  # noqa
from scanerf_amd import render
from scanerf_amd.tile_model import TileModel, train_step_fused
DEV = "cuda:0"
_PSO = os.path.join(ROOT, "tools", "probe", "libstate_poison.so")
if not os.path.exists(_PSO):   # (git-ignored build product)
    import subprocess
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-O2", "-o", _PSO,
                           os.path.join(ROOT, "tools", "probe", "state_poison.hip")])
P = ctypes.CDLL(_PSO)
torch.manual_seed(9)
B, S = int(os.environ.get("B", 8192)), 128
N = int(os.environ.get("N", 300))
LOG2T = int(os.environ.get("LOG2T", 16))
o = torch.rand(B, 3, device=DEV) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1) * (0.5 + torch.rand(B, 1, device=DEV))
tgt = torch.rand(B, 3, device=DEV)
m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=LOG2T, seed=1)
with torch.no_grad():
    m.features.mul_(100.0)
opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
for i in range(2):
    train_step_fused(m, opt, o, d, tgt, S, 20000 + i)
z, dist = m.sample(o, d, S)
m.packed.pack(m.decoder.blob(), m.weight_feature(20000))
box = (m.min_bbox.tolist(), m.bbox_size.tolist(), render.FORE, False)
table = m.gather_table()
valid = render.ray_valid(z)
sink = torch.zeros(1024, device=DEV)
EV = torch.rand(1 << 16, device=DEV)
stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
f2u = lambda f: struct.unpack("<I", struct.pack("<f", f))[0]
u2f = lambda u: struct.unpack("<f", struct.pack("<I", u & 0xffffffff))[0]


def evict():
    t = EV
    for f in (torch.sin, torch.cos, torch.exp, torch.erf, torch.tanh, torch.sigmoid, torch.sqrt, torch.abs, torch.floor, torch.ceil,
              torch.log1p, torch.atan, torch.sinh, torch.cosh, torch.round, torch.trunc, torch.neg, torch.reciprocal, torch.square, torch.sign):
        t = f(t.abs() + 1.0)
    t = torch.cumsum(t, 0); t = torch.sort(t)[0]; t = t.half().float(); t = (t.view(256, -1) @ t.view(-1, 256)); t = torch.softmax(t, -1)
    return t.to(torch.bfloat16).to(torch.float64).sum()


def between(mode):
    if mode == "evict":
        evict()
    if mode in ("icache", "both"):
        assert P.icache_sweep(ctypes.c_void_p(sink.data_ptr()), stream()) == 0
    if mode in ("poisonP", "both"):
        assert P.poison_regs_lds(ctypes.c_uint32(f2u(1234.5)), ctypes.c_void_p(sink.data_ptr()), stream()) == 0
    if mode == "poison0":
        assert P.poison_regs_lds(ctypes.c_uint32(0), ctypes.c_void_p(sink.data_ptr()), stream()) == 0


def launch():
    tile_T = torch.empty(B, render.tile_T_columns(S), device=DEV)
    xs = torch.empty(B * S, 32, device=DEV)
    out = render.render_forward(o, d, z, dist, table, m.resolution, m.packed, *box, want_weights=False, tile_T=tile_T, xstash=xs, plan=True,
                                ray_valid=valid)[0]
    return out, xs


def corners(ray, s, level):
    """(w_c, f_c [2], entry index) of the 8 corners of (ray, sample, level): csrc/hashgrid_common.h restated in numpy f32."""
    f32 = np.float32
    oo, dd, zz = o[ray].cpu().numpy(), d[ray].cpu().numpy(), f32(z[ray, s].item())
    mn, sz = np.float32(m.min_bbox.cpu().numpy()), np.float32(m.bbox_size.cpu().numpy())
    res = m.resolution[level].cpu().numpy()
    b, t = [], []
    for k in range(3):
        w = f32(oo[k] + f32(zz * dd[k]))
        p = f32(f32(f32(f32(w - mn[k]) / sz[k]) * f32(4.0)) - f32(2.0))
        p01 = f32(f32(p + f32(2.0)) / f32(4.0))
        v = f32(p01 * f32(res[k] - 1))
        b.append(int(v)); t.append(f32(v - f32(int(v))))
    T = table.shape[1]
    out = []
    for c in range(8):
        dx, dy, dz = c >> 2, (c >> 1) & 1, c & 1
        idx = ((b[0] + dx) ^ (((b[1] + dy) * 2654435761) & 0xffffffff) ^ (((b[2] + dz) * 805459861) & 0xffffffff)) & (T - 1)
        w = f32((t[0] if dx else 1 - t[0]) * (t[1] if dy else 1 - t[1]) * (t[2] if dz else 1 - t[2]))
        out.append((w, table[level, idx].detach().float().cpu().numpy(), idx))
    return out


ref_out, ref_xs = launch()
torch.cuda.synchronize()
ref_out, ref_xs = ref_out.clone(), ref_xs.clone()
modes = os.environ.get("MODES", "none,evict,icache,poisonP,poison0,both").split(",")
for mode in modes:
    bad = 0
    shown = 0
    for it in range(N):
        between(mode)
        out, xs = launch()
        if torch.equal(xs, ref_xs) and torch.equal(out, ref_out):
            continue
        bad += 1
        if shown >= int(os.environ.get("SHOW", 6)):
            continue
        shown += 1
        R3, X3 = ref_xs.view(B, S, 32), xs.view(B, S, 32)
        dif = (X3 != R3).nonzero()
        print(f"[{mode}] launch {it}: {dif.shape[0]} differing x-stash elements; positions {sorted(set(dif[:, 2].tolist()))}; "
              f"ray {dif[0, 0].item()} samples {dif[:, 1].min().item()}..{dif[:, 1].max().item()}; out rows {int((out != ref_out).any(1).sum())}", flush=True)
        for (ray, s, pos) in dif[:3].tolist():
            h, jj, ft = pos >> 4, (pos & 15) >> 1, pos & 1
            level = 4 * (jj >> 1) + 2 * h + (jj & 1)
            good, wrong = R3[ray, s, pos].item(), X3[ray, s, pos].item()
            cs = corners(ray, s, level)
            host = float(sum(np.float32(w) * f[ft] for w, f, _ in cs))
            print(f"    (ray {ray}, sample {s}, lane {32 * h + (s & 31)}, level {level}, feature {ft}): right {good:+.7e} (host {host:+.7e}) wrong {wrong:+.7e}")
            for c, (w, f, idx) in enumerate(cs):
                stale = f[ft] + (wrong - good) / float(w) if w != 0 else float("nan")
                addr = table.data_ptr() + (level * table.shape[1] + idx) * 8
                print(f"        corner {c}: w {float(w):.5f} f {f[ft]:+.6e}  stale {stale:+.6e}   [addr lo as f32 {u2f(addr):+.3e}, addr hi as f32 {u2f(addr >> 32):+.3e}, "
                      f"other feature {f[1 - ft]:+.6e}]")
    torch.cuda.synchronize()
    print(f"== {mode}: {bad} of {N} launches differ from the reference launch", flush=True)
print("done")
