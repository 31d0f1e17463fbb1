"""Can the forward of one half-batch (bound by the chip's L2-miss request rate) and the backward of the other half (bound by
vector issue) run AT THE SAME TIME on disjoint sets of CUs, and what does each cost then?  Timing only.
    X = CUs (persistent workgroups) given to the forward; the backward gets 256 - X.
Prints, per X: forward alone on X CUs, backward alone on 256 - X CUs, both together (two streams), and the sum."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd  # noqa
from scanerf_amd import network, render
from scanerf_amd.tile_model import TileModel
dev = "cuda:0"
Bh, S = int(os.environ.get("BH", 32768)), 128
torch.manual_seed(0)
m = TileModel([-4, -4, -4], [8, 8, 8], dev, log2_T=19)
wf = network.weight_feature(40000, dev)
m.packed.pack(m.decoder.blob(), wf)
box = (m.min_bbox.tolist(), m.bbox_size.tolist(), render.FORE, False)
T = m.features.shape[1]
halves = []
for h in range(2):
    o = torch.rand(Bh, 3, device=dev) * 8 - 4
    d = torch.nn.functional.normalize(torch.randn(Bh, 3, device=dev), dim=-1)
    z, dist = m.sample(o, d, S)
    halves.append(dict(o=o, d=d, z=z, dist=dist, tile_T=torch.empty(Bh, (S + 15) // 16, device=dev), xs=torch.empty(Bh * S, 32, device=dev),
                       out=torch.empty(Bh, 16, device=dev), g=torch.randn(Bh, 16, device=dev) / Bh, gt=torch.zeros_like(m.features),
                       gblob=torch.zeros(network.PARAMSIZE, device=dev)))
def fwd(h):
    render.render_forward(h["o"], h["d"], h["z"], h["dist"], m.features, m.resolution, m.packed, *box, want_weights=False, out_ray=h["out"], tile_T=h["tile_T"], xstash=h["xs"])
def bwd(h):
    render.render_backward(h["o"], h["d"], h["z"], h["dist"], m.features, m.resolution, m.packed, wf, *box, h["out"], h["tile_T"], h["g"], xstash=h["xs"],
                           scatter=(h["wsp"], h["gt"]), want_dfeat=False, grad_blob=h["gblob"])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn_a, fn_b, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        s1.wait_event(e0); s2.wait_event(e0)
        if fn_a:
            with torch.cuda.stream(s1): fn_a()
        if fn_b:
            with torch.cuda.stream(s2): fn_b()
        e1.record(s1); e2.record(s2)
        torch.cuda.synchronize()
        best = min(best, max(e0.elapsed_time(e1), e0.elapsed_time(e2)))
    return best
A, Bq = halves
for X in [int(x) for x in os.environ.get("XS", "256,128,96,64,48,32").split(",")]:
    os.environ["SCANERF_FWD_GRID"] = str(X)
    os.environ["SCANERF_BWD_GRID"] = str(max(256 - X, 1) if X < 256 else 256)
    # half A: forward + plan under THIS backward grid (the plan's ranges are per producer workgroup)
    os.environ["SCANERF_FWD_GRID"] = "256"
    fwd(A)
    os.environ["SCANERF_FWD_GRID"] = str(X)
    A["wsp"] = render.scatter_plan(A["o"], A["d"], A["z"], m.resolution, T, *box)
    torch.cuda.synchronize()
    tf = timed(lambda: fwd(Bq), None)
    tb = timed(None, lambda: bwd(A))
    tfb = timed(lambda: fwd(Bq), lambda: bwd(A))
    print(f"forward on {X:3d} CUs {tf:6.3f} ms | backward on {int(os.environ['SCANERF_BWD_GRID']):3d} CUs {tb:6.3f} ms | together {tfb:6.3f} ms | sum {tf + tb:6.3f}  (half batches of {Bh} rays)")
