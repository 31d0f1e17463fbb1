"""Times scatter_accumulate alone on the records of one fused training step (BASELINE config 2)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd
from scanerf_amd import network, render
from scanerf_amd.tile_model import TileModel
dev = "cuda:0"
B, S = 65536, 128
torch.manual_seed(0)
m = TileModel([-4, -4, -4], [8, 8, 8], dev, log2_T=19)
o = torch.rand(B, 3, device=dev) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
z, dist = m.sample(o, d, S)
wf = network.weight_feature(40000, dev)
m.packed.pack(m.decoder.blob(), wf)
box = (m.min_bbox.tolist(), m.bbox_size.tolist(), render.FORE, False)
tile_T = torch.empty(B, 4, device=dev); xs = torch.empty(B * S, 32, device=dev)
out, _ = render.render_forward(o, d, z, dist, m.features, m.resolution, m.packed, *box, want_weights=False, tile_T=tile_T, xstash=xs)
g = torch.randn(B, 16, device=dev) / B
T = m.features.shape[1]
ws = render.scatter_plan(o, d, z, m.resolution, T, *box)
gt = torch.zeros_like(m.features)
render.render_backward(o, d, z, dist, m.features, m.resolution, m.packed, wf, *box, out, tile_T, g, xstash=xs, scatter=(ws, gt), want_dfeat=False)
for _ in range(2): render.scatter_accumulate(ws, gt, B, S)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): render.scatter_accumulate(ws, gt, B, S)
e1.record(); torch.cuda.synchronize()
print(f"variant {os.environ.get('SCANERF_ACC_VARIANT', '0')}: scatter_accumulate {e0.elapsed_time(e1)/5:.3f} ms")
