"""Fused table-gradient path (plan -> backward emits records -> accumulate) against the exact scatter of the same dfeat, for
upstream gradients scaled by 1, 1e10, 1e22, 1e-22 and both backward arithmetics (t16: 8-byte records; h3: 16-byte)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd
from scanerf_amd import network, render
from scanerf_amd.tile_model import TileModel
DEV = "cuda:0"
def run(scale, arith):
    render.set_arith(arith)
    torch.manual_seed(31)
    B, S_ = 2500, 48
    m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=14, seed=2)
    with torch.no_grad():
        m.features.mul_(30.0)
    o = torch.rand(B, 3, device=DEV) * 8 - 4
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1)
    z, dist = m.sample(o, d, S_)
    valid = torch.rand(B, device=DEV) < 0.5
    wf = network.weight_feature(3000, DEV)
    m.packed.pack(m.decoder.blob(), wf)
    box = (m.min_bbox.tolist(), m.bbox_size.tolist(), render.FORE, False)
    T = m.features.shape[1]
    tile_T = torch.empty(B, (S_ + 15) // 16, device=DEV)
    xs = torch.empty(B * S_, 32, device=DEV)
    out, _ = render.render_forward(o, d, z, dist, m.features, m.resolution, m.packed, *box, ray_valid=valid, want_weights=False, tile_T=tile_T, xstash=xs)
    ws = render.scatter_plan(o, d, z, m.resolution, T, *box, ray_valid=valid)
    gout = torch.randn(B, 16, device=DEV) * scale
    args = (o, d, z, dist, m.features, m.resolution, m.packed, wf, *box, out, tile_T, gout)
    dfeat, gb1 = render.render_backward(*args, ray_valid=valid, xstash=xs)
    pts = ((o[:, None, :] + z[:, :, None] * d[:, None, :]).reshape(-1, 3) - m._min_dev) / m._size_dev * 4.0 - 2.0
    g1 = render.scatter_table_grad(pts.contiguous(), dfeat, torch.zeros_like(m.features), m.resolution)
    g2 = torch.zeros_like(m.features)
    _, gb2 = render.render_backward(*args, ray_valid=valid, xstash=xs, scatter=(ws, g2), want_dfeat=False)
    render.scatter_accumulate(ws, g2, B, S_)
    torch.cuda.synchronize()
    sc = float(g1.abs().max())
    print(arith, os.environ.get("SCANERF_REC16"), "scale", scale, "dfeat max", float(dfeat.abs().max()), "g1 max", sc, "g2 max", float(g2.abs().max()),
          "nz1", int((g1 != 0).sum()), "nz2", int((g2 != 0).sum()), "rel L2", float((g2 - g1).double().norm() / g1.double().norm()),
          "gblob equal", bool(torch.equal(gb1, gb2)), "maxbits", ws[: 16 * 2 * 256 * 4 + 2 * 32 * 4 + 64].view(torch.float32)[16 * 2 * 256 + 2 * 32 + 1].item())
for scale in (1.0, 1e10, 1e22, 1e-22):
    for arith in ("t16", "h3"):
        run(scale, arith)
