"""Run the fused forward+backward a few times (for rocprof PMC passes on k_render_bwd)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd
from scanerf_amd import network, render
from scanerf_amd.tile_model import TileModel
dev = "cuda:0"
B, S = int(os.environ.get("B", 65536)), 128
torch.manual_seed(0)
m = TileModel([-4, -4, -4], [8, 8, 8], dev, log2_T=19)
o = torch.rand(B, 3, device=dev) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
z, dist = m.sample(o, d, S)
wf = network.weight_feature(5000, dev)
m.packed.pack(m.decoder.blob(), wf)
tile_T = torch.empty(B, (S + 15) // 16, device=dev)
box = (m.min_bbox.tolist(), m.bbox_size.tolist(), render.FORE, False)
XS = torch.empty(B * S, 32, device=dev) if os.environ.get("XS", "1") == "1" else None
out, _ = render.render_forward(o, d, z, dist, m.features, m.resolution, m.packed, *box, want_weights=False, tile_T=tile_T, xstash=XS)
g = torch.randn(B, 16, device=dev)
for _ in range(int(os.environ.get("N", 3))):
    render.render_backward(o, d, z, dist, m.features, m.resolution, m.packed, wf, *box, out, tile_T, g, xstash=XS)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    render.render_backward(o, d, z, dist, m.features, m.resolution, m.packed, wf, *box, out, tile_T, g, xstash=XS)
e1.record(); torch.cuda.synchronize()
print(f"render_backward {e0.elapsed_time(e1)/3:.2f} ms")
