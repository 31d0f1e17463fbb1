"""How often does the first forward of a training step on a fresh model differ from the same forward of other fresh models?
(The state is identical; the launch follows the model's construction and the sampler, with cold caches -- the context in which
tests/test_gpu_determinism.py::test_whole_training_step_is_bit_reproducible failed.)  Must print 0."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd  # noqa
from scanerf_amd import render
from scanerf_amd.tile_model import TileModel, train_step_fused
DEV = "cuda:0"
torch.manual_seed(11)
B, S = int(os.environ.get("B", 8192)), 128
RUNS = int(os.environ.get("RUNS", 60))
o = torch.rand(B, 3, device=DEV) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1) * (0.5 + torch.rand(B, 1, device=DEV))
tgt = torch.rand(B, 3, device=DEV)
KEEP = []
_fwd = render.render_forward


def fwd_keep(*a, **k):
    r = _fwd(*a, **k)
    torch.cuda.synchronize()
    KEEP.append((r[0].clone(), k["xstash"].clone()))
    return r


render.render_forward = fwd_keep
bad = 0
ref = None
for run in range(RUNS):
    KEEP.clear()
    m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=16, seed=1)
    with torch.no_grad():
        m.features.mul_(100.0)
    opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    train_step_fused(m, opt, o, d, tgt, S, 20000, pose_grads=bool(int(os.environ.get("POSE", "0"))))
    out, xs = KEEP[0]
    if ref is None:
        ref = (out, xs)
        continue
    if not torch.equal(out, ref[0]) or not torch.equal(xs, ref[1]):
        bad += 1
        xr = (xs != ref[1]).view(B, S, 32)
        print(f"  run {run}: out rows {int((out != ref[0]).any(1).sum())}; x-stash samples {int(xr.any(2).sum())}, per feature "
              f"{[(i, int(n)) for i, n in enumerate(xr.sum((0, 1)).tolist()) if n]}", flush=True)
print(f"{os.environ.get('TAG', '')}: {bad} of {RUNS - 1} first forwards differ from run 0's")
