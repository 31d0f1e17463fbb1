"""One training iteration with and without the pose-gradient outputs (dL/d rays for bundle adjustment), both backward arithmetics."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd
from scanerf_amd import render
from scanerf_amd.tile_model import TileModel, train_step_fused
DEV = "cuda:0"
torch.manual_seed(0)
B, S = 65536, 128
m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=19, seed=1)
o = torch.rand(B, 3, device=DEV) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1)
tgt = torch.rand(B, 3, device=DEV)
opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
for ar in ("t16", "h3"):
    render.set_arith(ar)
    for pose in (False, True):
        for i in range(3): train_step_fused(m, opt, o, d, tgt, S, 20000 + i, pose_grads=pose)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(10): train_step_fused(m, opt, o, d, tgt, S, 20010 + i, pose_grads=pose)
        torch.cuda.synchronize()
        print(f"arith {ar} pose_grads={pose}: {(time.perf_counter() - t0) * 100:.2f} ms per step")
# the reference's default iteration: foreground + background + pose refinement
from scanerf_amd.tile_model import train_step_fgbg
render.set_arith("t16")
for pose in (False, True):
    for i in range(3): train_step_fgbg(m, opt, o, d, tgt, S, S, 20100 + i, pose_grads=pose)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): train_step_fgbg(m, opt, o, d, tgt, S, S, 20110 + i, pose_grads=pose)
    torch.cuda.synchronize()
    print(f"fg + bg iteration (128 + 128 samples), pose_grads={pose}: {(time.perf_counter() - t0) * 100:.2f} ms per step")
