"""Fused forward only, both decoder arithmetics, at BASELINE config 2 (65 536 rays x 128 samples, T=2^19)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd
from scanerf_amd import network, render
from scanerf_amd.tile_model import TileModel
dev = "cuda:0"
B, S = int(os.environ.get("B", 65536)), 128
torch.manual_seed(0)
m = TileModel([-4, -4, -4], [8, 8, 8], dev, log2_T=int(os.environ.get("LOG2T", 19)))
o = torch.rand(B, 3, device=dev) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
sort = os.environ.get("SORT", "0")
if sort != "0":  # ray-ordering experiment: rays sorted by the Morton code of their origin's cell (2^k cells per axis)
    k = int(sort)
    c = ((o + 4) / 8 * (1 << k)).long().clamp(0, (1 << k) - 1)
    code = torch.zeros(B, dtype=torch.long, device=dev)
    for b in range(k):
        for a in range(3):
            code |= ((c[:, a] >> b) & 1) << (3 * b + a)
    if os.environ.get("SORT_DIR", "0") != "0":   # ... and, inside a cell, by the direction's octant and dominant axis (coherent bundles)
        dd = torch.nn.functional.normalize(d, dim=-1)
        octant = ((dd[:, 0] > 0).long() | ((dd[:, 1] > 0).long() << 1) | ((dd[:, 2] > 0).long() << 2))
        code = code * 32 + octant * 4 + dd.abs().argmax(-1)
    order = torch.argsort(code)
    o, d = o[order].contiguous(), d[order].contiguous()
z, dist = m.sample(o, d, S)
m.packed.pack(m.decoder.blob(), network.weight_feature(40000, dev))
box = (m.min_bbox.tolist(), m.bbox_size.tolist(), render.FORE, False)
outs = {}
for ar in os.environ.get("ARITH", "f32,h3").split(","):
    render.set_arith(ar)
    for dt in (torch.float32, torch.bfloat16):
        F = m.features.detach().to(dt).contiguous()
        for xs in (None, torch.empty(B * S, 32, device=dev)):
            f = lambda: render.render_forward(o, d, z, dist, F, m.resolution, m.packed, *box, want_weights=False, xstash=xs)
            for _ in range(3): out, _ = f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f"render_forward arith={ar} table={dt} xstash={xs is not None}: {ms:.3f} ms  {B/ms*1e3:.3e} rays/s", flush=True)
        outs[(ar, dt)] = out.clone()
if ("f32", torch.float32) in outs and ("h3", torch.float32) in outs:
    a, b = outs[("f32", torch.float32)], outs[("h3", torch.float32)]
    e = ((a - b).abs() / (1e-6 + 1e-4 * a.abs()))[:, :15]
    print("h3 vs f32 (allclose units, <=1 passes): max", e.max().item(), " n>1:", int((e > 1).sum()))
