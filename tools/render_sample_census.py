"""How many samples do the inference kernels of one configs[4] frame decode?  Wraps the renderer's inference calls and counts the
samples with a non-zero alpha after each (fg: listed and occupied; bg: every sample of a ray with a background tile)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from scanerf_amd import renderer as R

counts = []
def wrap(name):
    f = getattr(R, name)
    def g(*a, **k):
        f(*a, **k)
        pa = a[-1] if name != "bg_pts_inference_v2" else a[-1]
        live = (pa > 0).reshape(-1)
        n16 = live.numel() // 16 * 16
        tiles = live[:n16].reshape(-1, 16)   # 16 consecutive slots = one decoder tile of the 16-sample-tile kernel (any layout)
        any_t = tiles.any(dim=1)
        counts.append((name, int(live.sum()), live.numel(), int(any_t.sum()), float(tiles[any_t].float().mean()) if bool(any_t.any()) else 0.0))
    setattr(R, name, g)
for n in ("pts_inference_tracing", "pts_inference", "bg_pts_inference_v2"):
    wrap(n)
args = types.SimpleNamespace(tiles_per_gpu=1, log2_T=19, samples=128)
bench.time_render(args, 1, 0, torch.device("cuda:0"), 1, 0)
for c in counts:
    print(c[0], "alpha > 0:", c[1], "of", c[2], f"({c[1] / c[2]:.3f}); tiles with a live sample: {c[3]}, filled {c[4]:.3f}")
