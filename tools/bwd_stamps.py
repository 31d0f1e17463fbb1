"""Where the 16-sample-tile backward spends its time: per-interval shader-cycle sums of the -DT16_STAMPS build
(csrc/render_bwd_t16.hip STAMP), averaged over all waves.

    tools/build_variant.py stamps render_bwd_t16="-DT16_STAMPS"
    SCANERF_LIB=<pkg>/lib/debug/libscanerf_hip_stamps.so python tools/bwd_stamps.py
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd  # noqa
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
render._KEEP_DW_PARTIAL = []
def step():
    ws = render.scatter_plan(o, d, z, m.resolution, T, *box)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    render.render_backward(o, d, z, dist, m.features, m.resolution, m.packed, wf, *box, out, tile_T, g, xstash=xs, scatter=(ws, gt), want_dfeat=False)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for _ in range(2): step()
ms = step()
dw, nblk = render._KEEP_DW_PARTIAL
st = dw[nblk:2 * nblk].reshape(nblk, -1)[:, :256].reshape(nblk, 8, 32)[:, :, :23].double()   # [wg][wave][interval]
tot = st.sum(-1)
names = ["fwd recompute", "compositing + adjoint", "wait S", "scale + narrow staging", "wait A1", "wgrad nar + chain rgb", "wait B1",
         "stage D1", "wait A2", "wgrad D1 + chain D1", "wait B2", "stage D0", "wait A3", "wgrad D0 + chain D0/heads", "wait B3",
         "stage L1 + u0 recompute", "wait A4", "wgrad L1 + chain L1", "wait B4", "stage L0", "wait A5", "wgrad L0 + chain L0 (+pose)",
         "load next + emission (+ per-ray prologue)"]
print(f"backward (stamps build) {ms:.3f} ms; cycles per wave: mean {tot.mean():.4g} (min {tot.min():.4g}, max {tot.max():.4g}) = {tot.mean() / ms / 1e3:.0f} MHz x ms")
mean = st.mean((0, 1))
ntile = B * (S // 16) / (nblk * 8)
waits = 0.0
lo, hi = st[:, :4].mean((0, 1)) / ntile, st[:, 4:].mean((0, 1)) / ntile   # waves 0-3 (first wave of each SIMD) / 4-7
for i, n in enumerate(names):
    print(f"{i:2d} {n:44s} {mean[i] / ntile:9.0f} cyc/tile  {100 * mean[i] / tot.mean():5.1f} %   waves 0-3: {lo[i]:7.0f}  4-7: {hi[i]:7.0f}")
    if n.startswith("wait"):
        waits += float(mean[i])
print(f"barrier waits together {100 * waits / tot.mean():.1f} %; per tile {tot.mean() / ntile:.0f} cycles")
# spread between the waves of a workgroup in the emission interval (who arrives late at barrier S?)
em = st[:, :, 22] / ntile
print("emission interval per wave index (mean over workgroups):", [round(float(x)) for x in em.mean(0)])
