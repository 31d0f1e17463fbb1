"""Time one full training iteration on the 'ops' path at config 2 (probe, not the judged bench)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd  # noqa
from scanerf_amd.tile_model import TileModel, train_step_ops

dev = "cuda:0"
B, S = int(os.environ.get("B", 65536)), 128
torch.manual_seed(0)
m = TileModel([-4, -4, -4], [8, 8, 8], dev, log2_T=19)
opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
o = torch.rand(B, 3, device=dev) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1) * (0.5 + torch.rand(B, 1, device=dev))
tgt = torch.rand(B, 3, device=dev)
for i in range(2):
    l = train_step_ops(m, opt, o, d, tgt, S, 1000 + i)
torch.cuda.synchronize()
t0 = time.time()
n = 3
for i in range(n):
    l = train_step_ops(m, opt, o, d, tgt, S, 2000 + i)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print(f"ops-path train step: {dt*1e3:.1f} ms  {B/dt:.3e} rays/s  loss {float(l):.5f}  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
