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
        counts.append((name, int((pa > 0).sum()), pa.numel()))
    setattr(R, name, g)
for n in ("pts_inference_tracing", "pts_inference", "bg_pts_inference_v2"):
    wrap(n)
args = types.SimpleNamespace(tiles_per_gpu=1, log2_T=19, samples=128)
bench.time_render(args, 1, 0, torch.device("cuda:0"), 1, 0)
for c in counts:
    print(c[0], "alpha > 0:", c[1], "of", c[2], f"({c[1] / c[2]:.3f})")
