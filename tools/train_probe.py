"""Time one full training iteration (ops vs fused path) at config 2 (probe, not the judged bench)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd  # noqa
from scanerf_amd.tile_model import KernelTimer, TileModel, train_step_fused, train_step_ops

dev = "cuda:0"
B, S = int(os.environ.get("B", 65536)), 128
o = torch.rand(B, 3, device=dev) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1) * (0.5 + torch.rand(B, 1, device=dev))
tgt = torch.rand(B, 3, device=dev)
for name, fn in (("fused", train_step_fused), ("ops", train_step_ops)):
    if os.environ.get("ONLY") and os.environ["ONLY"] != name:
        continue
    torch.manual_seed(0)
    m = TileModel([-4, -4, -4], [8, 8, 8], dev, log2_T=19)
    opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    losses = [float(fn(m, opt, o, d, tgt, S, 1000 + i)) for i in range(3)]
    torch.cuda.synchronize()
    tm = KernelTimer()
    t0 = time.time()
    n = 5
    for i in range(n):
        l = fn(m, opt, o, d, tgt, S, 2000 + i, timer=tm)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    print(f"{name}: {dt*1e3:.1f} ms/step  {B/dt:.3e} rays/s  losses {losses} -> {float(l):.5f}  "
          f"mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    print("   ", {k: round(v, 3) for k, v in tm.summary().items()})
