import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scanerf_amd  # noqa
from scanerf_amd import renderer as R, tile_model as tm
dev = torch.device("cuda:0")
H, W, ntile = 1080, 1920, 4
tiles = []
with tempfile.TemporaryDirectory() as tmp:
    for t in range(ntile):
        m = tm.TileModel([-4.0 * ntile + 8.0 * t, -4, -4], [8, 8, 8], dev, log2_T=19, seed=t, sampler_log2dim=7)
        m.set_occupancy(tm.sphere_shell_occupancy(m, 3.0, 0.5))
        with torch.no_grad():
            m.features.mul_(300.0)
        R.export_tile(os.path.join(tmp, f"tile{t}"), m)
        tiles.append(R.load_tile(os.path.join(tmp, f"tile{t}")))
        del m
rend = R.TileSetRenderer(dev, tiles)
import scanerf_amd.renderer as RR
ev = []
orig = RR.pts_inference_tracing
def timed(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(*a, **k); e1.record()
    ev.append((e0, e1, a[2]))
RR.pts_inference_tracing = timed
for name, K, c2w in (("bench view", [1600.0, 0, W / 2, 0, 1600.0, H / 2, 0, 0, 1], torch.tensor([[1.0, 0, 0, 0.0], [0, 1, 0, 0.5], [0, 0, 1, -14.0]])),
                     ("close view", [800.0, 0, W / 2, 0, 800.0, H / 2, 0, 0, 1], torch.tensor([[1.0, 0, 0, -4.0], [0, 1, 0, 0.0], [0, 0, 1, -4.2]]))):
    for it in range(3):
        ev.clear()
        rend.render(H, W, K, c2w)
        torch.cuda.synchronize()
    for e0, e1, z in ev:
        live = int((z.reshape(-1) != -1).sum())
        ms = e0.elapsed_time(e1)
        print(name, f"fg inference {ms:.2f} ms, sampled slots {live} of {z.numel()} -> {live / ms / 1e6:.2f} G samples/s")
