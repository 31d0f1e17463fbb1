"""Times the fused backward WITH record emission (plan + backward; no accumulate), for emission experiments.
SCANERF_DEBUG_BWD (timing experiments only): 1 = no record stores, 2 = no cursor atomics, 3 = neither (index arithmetic only)."""
import os, sys, torch
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
z, dist = m.sample(o, d, S)
wf = network.weight_feature(40000, dev)
m.packed.pack(m.decoder.blob(), wf)
box = (m.min_bbox.tolist(), m.bbox_size.tolist(), render.FORE, False)
tile_T = torch.empty(B, (S + 15) // 16, device=dev); xs = torch.empty(B * S, 32, device=dev)
out, _ = render.render_forward(o, d, z, dist, m.features, m.resolution, m.packed, *box, want_weights=False, tile_T=tile_T, xstash=xs)
g = torch.randn(B, 16, device=dev) / B
T = m.features.shape[1]
gt = torch.zeros_like(m.features)
def step():
    ws = render.scatter_plan(o, d, z, m.resolution, T, *box)
    render.render_backward(o, d, z, dist, m.features, m.resolution, m.packed, wf, *box, out, tile_T, g, xstash=xs, scatter=(ws, gt), want_dfeat=False)
for _ in range(2): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4): step()
e1.record(); torch.cuda.synchronize()
print(f"arith {os.environ.get('SCANERF_ARITH', 'default')} dbg {os.environ.get('SCANERF_DEBUG_BWD', '0')}: plan + backward(emit) {e0.elapsed_time(e1)/4:.3f} ms")
