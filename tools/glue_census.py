"""Which host-side lines launch the small torch kernels (fills, copies, elementwise) of one training iteration: torch.profiler with
stacks on the reference's shipped configuration (T = 2^24, 16 384 rays, fg + bg, pose gradients) or, with `default`, configs[1]'s step."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scanerf_amd  # noqa
from scanerf_amd import tile_model as tm
from torch.profiler import profile, ProfilerActivity

dev = "cuda:0"
default = len(sys.argv) > 1 and sys.argv[1] == "default"
B, S, L2T = (65536, 128, 19) if default else (16384, 128, 24)
g = torch.Generator(device=dev).manual_seed(24)
m = tm.TileModel([-4.0, -4, -4], [8, 8, 8], dev, log2_T=L2T, seed=24, sampler_log2dim=4)
opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
ro = torch.rand(B, 3, device=dev, generator=g) * 8 - 4
rd = torch.nn.functional.normalize(torch.randn(B, 3, device=dev, generator=g), dim=-1) * (0.5 + torch.rand(B, 1, device=dev, generator=g))
tg = torch.rand(B, 3, device=dev, generator=g)
step = (lambda i: tm.train_step_fused(m, opt, ro, rd, tg, S, 20000 + i)) if default else \
       (lambda i: tm.train_step_fgbg(m, opt, ro, rd, tg, S, S, 20000 + i, pose_grads=True))
for i in range(2):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(2)
    torch.cuda.synchronize()
by = collections.Counter()
dur = collections.Counter()
for e in prof.events():
    if e.device_type.name != "CPU" or not e.name.startswith("aten::"):
        continue
    if e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue   # top-level aten ops only
    kern = sum(k.duration for k in e.kernels) if hasattr(e, "kernels") else 0
    st = [s for s in (e.stack or []) if "scanerf" in s or "tile_model" in s or "render.py" in s or "optim" in s]
    key = (e.name, st[0].split("/")[-1] if st else "?")
    by[key] += 1
    dur[key] += e.device_time_total if hasattr(e, "device_time_total") else kern
for k, n in sorted(by.items(), key=lambda kv: -dur[kv[0]])[:60]:
    print(f"{n:4d} x {k[0]:28s} {dur[k]:9.1f} us   {k[1]}")
print("total top-level aten ops:", sum(by.values()), " device us:", sum(dur.values()))
