"""Runs the fused forward N times on the same inputs and counts rays whose outputs differ between runs (must be 0)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd
from scanerf_amd import network, render
from scanerf_amd.tile_model import TileModel
dev = "cuda:0"
B, S = int(os.environ.get("B", 16384)), int(os.environ.get("S", 64))
torch.manual_seed(0)
dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[os.environ.get("DT", "f32")]
m = TileModel([-4, -4, -4], [8, 8, 8], dev, log2_T=14)
with torch.no_grad():
    m.features.mul_(200.0)
o = torch.rand(B, 3, device=dev) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
bg = os.environ.get("MODE", "fore") == "bg"
if bg:
    z, dist, _ = m.inverse_z_sampling(o, d, S)
else:
    z, dist = m.sample(o, d, S)
wf = network.weight_feature(40000, dev)
m.packed.pack(m.decoder.blob(), wf)
box = (m.min_bbox.tolist(), m.bbox_size.tolist(), render.BG if bg else render.FORE, bg)
table = m.features.detach().to(dt).contiguous()
ref = None
bad = 0
for it in range(int(os.environ.get("N", 20))):
    out, w = render.render_forward(o, d, z, dist, table, m.resolution, m.packed, *box)
    torch.cuda.synchronize()
    if ref is None:
        ref = (out.clone(), w.clone())
    else:
        m_o, m_w = (out != ref[0]), (w != ref[1])
        nb = int((m_o.any(1) | m_w.any(1)).sum())
        bad += nb
        if nb and it <= 2:
            for r in torch.nonzero(m_o.any(1) | m_w.any(1))[:4, 0].tolist():
                cols = torch.nonzero(m_o[r])[:, 0].tolist()
                ws = torch.nonzero(m_w[r])[:, 0].tolist()
                if ws:
                    print("     ratio w/ref at differing samples:", [f"{float(w[r, k] / ref[1][r, k]):.5f}" for k in ws[:16]])
                print(f"  it {it} ray {r} (wg slot {r % 8}): out cols {cols} max rel {float(((out[r] - ref[0][r]).abs() / (ref[0][r].abs() + 1e-12)).max()):.2e}; "
                      f"weights differ at samples {ws[:6]}{'...' if len(ws) > 6 else ''} ({len(ws)}) max abs {float((w[r] - ref[1][r]).abs().max()):.2e}")
print(f"arith {os.environ.get('SCANERF_ARITH', 'default')} S={S} mode={'bg' if bg else 'fore'} table {dt}: {bad} differing rays over the repeats")

# ---- backward: dfeat and the decoder gradient are sums in a fixed order -> bit-identical between launches
if os.environ.get("BWD", "1") == "1" and dt == torch.float32:
    tile_T = torch.empty(B, render.tile_T_columns(S), device=dev)
    xs = torch.empty(B * S, 32, device=dev)
    out, _ = render.render_forward(o, d, z, dist, table, m.resolution, m.packed, *box, want_weights=False, tile_T=tile_T, xstash=xs)
    g = torch.randn(B, 16, device=dev) / B
    refb, badb = None, 0
    for it in range(int(os.environ.get("N", 20))):
        dfeat, gblob = render.render_backward(o, d, z, dist, table, m.resolution, m.packed, wf, *box, out, tile_T, g, xstash=xs)
        torch.cuda.synchronize()
        if refb is None:
            refb = (dfeat.clone(), gblob.clone())
        else:
            nd = int((dfeat != refb[0]).any(2).any(0).sum())
            badb += nd + int((gblob != refb[1]).sum() > 0)
            if nd and it <= 2:
                idx = torch.nonzero((dfeat != refb[0]).any(2).any(0))[:6, 0].tolist()
                print("  backward: differing samples (ray, sample):", [(i // S, i % S) for i in idx])
    print(f"backward arith {os.environ.get('SCANERF_ARITH', 'default')} S={S}: {badb} differing samples/launches over the repeats")
