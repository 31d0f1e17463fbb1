"""Backward: h3 vs f32 kernels on the same inputs, error per decoder-blob section and for dfeat."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd
from scanerf_amd import network, render
from scanerf_amd.tile_model import TileModel
DEV = "cuda:0"
B, S_ = int(os.environ.get("B", 8192)), int(os.environ.get("S", 64))
torch.manual_seed(3)
m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=14, seed=1)
with torch.no_grad():
    m.features.mul_(30.0)
o = torch.rand(B, 3, device=DEV) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1) * (0.5 + torch.rand(B, 1, device=DEV))
z, dist = m.sample(o, d, S_)
valid = torch.all(z != -1, dim=-1)
wf = network.weight_feature(2000, DEV)
m.packed.pack(m.decoder.blob(), wf)
box = (m.min_bbox.tolist(), m.bbox_size.tolist(), render.FORE, False)
ntile = (S_ + 31) // 32
res = {}
for ar in ("f32", "h3", "h3"):
    render.set_arith(ar)
    tile_T = torch.empty(B, (S_ + 15) // 16, device=DEV)
    xs = torch.empty(B * S_, 32, device=DEV)
    out, _ = render.render_forward(o, d, z, dist, m.features, m.resolution, m.packed, *box, ray_valid=valid,
                                   want_weights=False, tile_T=tile_T, xstash=xs)
    torch.manual_seed(7)
    gout = torch.randn(B, 16, device=DEV) / B
    bufs = (torch.zeros(B, ntile, device=DEV), torch.zeros(B, 2, 64, device=DEV))
    dfeat, gb = render.render_backward(o, d, z, dist, m.features, m.resolution, m.packed, wf, *box, out, tile_T, gout,
                                       ray_valid=valid, xstash=xs, ray_grad_buffers=bufs)
    torch.cuda.synchronize()
    key = ar if ar not in res else ar + "b"
    res[key] = (dfeat.clone(), gb.clone(), bufs[0].clone(), bufs[1].sum(1).clone())
secs = [("S0.b", 0, 64), ("S0.W", 64, 2112), ("S1.b", 2112, 2176), ("S1.W", 2176, 6272), ("sig", 6272, 6305), ("dif", 6305, 6404),
        ("tint", 6404, 6503), ("D0.b", 6503, 6567), ("D0.W[:32]", 6567, 6567 + 32 * 64), ("D0.W[32:]", 6567 + 32 * 64, 9639),
        ("D1.b", 9639, 9703), ("D1.W", 9703, 13799), ("D2", 13799, 13994)]
a, b = res["f32"], res["h3"]
print("h3 run-to-run identical:", torch.equal(res["h3"][1], res["h3b"][1]), torch.equal(res["h3"][0], res["h3b"][0]))
for name, lo, hi in secs:
    x, y = a[1][lo:hi], b[1][lo:hi]
    print(f"  {name:10s} max|f32| {x.abs().max().item():.3e}  max|diff| {(x - y).abs().max().item():.3e}  rel {((x - y).abs().max() / x.abs().max()).item():.2e}")
for i, nm in ((0, "dfeat"), (2, "g_dnorm"), (3, "g_rowsum")):
    x, y = a[i], b[i]
    print(f"  {nm:10s} max|f32| {x.abs().max().item():.3e}  max|diff| {(x - y).abs().max().item():.3e}  rel {((x - y).abs().max() / x.abs().max()).item():.2e}")
