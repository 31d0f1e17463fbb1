"""GPU idle gaps in the steady state of the autograd route (default), the op-by-op route (`ops`) or the fused step (`fused`): torch.profiler device timeline
of two steps after warm-up; gaps > 3 us with the kernels on either side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scanerf_amd  # noqa
from torch.profiler import profile, ProfilerActivity

route = sys.argv[1] if len(sys.argv) > 1 else "autograd"
dev = "cuda:0"
B, S = 65536, 128
torch.manual_seed(0)
o = torch.rand(B, 3, device=dev) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1) * (0.5 + torch.rand(B, 1, device=dev))
tgt = torch.rand(B, 3, device=dev)
if route in ("ops", "fused"):
    from scanerf_amd import tile_model as tm
    m = tm.TileModel([-4.0, -4, -4], [8, 8, 8], dev, log2_T=19, seed=17, sampler_log2dim=4)
    opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    step = (lambda i: tm.train_step_ops(m, opt, o, d, tgt, S, 20000 + i)) if route == "ops" else \
           (lambda i: tm.train_step_fused(m, opt, o, d, tgt, S, 20000 + i))
else:
    from scanerf_amd import network
    from scanerf_amd.cuda import adam_step_cuda
    from scanerf_amd.hashgrid import HashGrid
    hg = HashGrid(dev, torch.tensor([-4.0, -4, -4]), torch.tensor([8.0, 8, 8]), log2_hashmap_size=19, grid_resolution=[32, 2048], sampler_log2dim=4)
    dec = network.init_model(network.ShallowMLP(32), "xavier").to(dev)
    opt = torch.optim.Adam(dec.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    m1, m2 = torch.zeros_like(hg.HE.features), torch.zeros_like(hg.HE.features)
    K = hg.HE.features.numel() // 8

    def step(i):
        hg.HE.features.grad = None
        opt.zero_grad(set_to_none=True)
        out, ok = hg.render_fore_rays(o, d, S, dec, 0, global_step=20000 + i)
        loss = torch.nn.functional.mse_loss(out["pred_color"], tgt) + 0.01 * out["l2_reg_specular"]
        loss.backward()
        with torch.no_grad():
            adam_step_cuda(hg.HE.features.data.view(K, 8), hg.HE.features.grad.view(K, 8), m1.view(K, 8), m2.view(K, 8), 1e-2, 0.9, 0.99, 1e-15, i)
        opt.step()
for i in range(3):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(3):
        step(3 + i)
    torch.cuda.synchronize()
ks = sorted([e for e in prof.events() if e.device_type.name == "CUDA"], key=lambda e: e.time_range.start)
t_end, last = None, None
gaps = []
for e in ks:
    if t_end is not None and e.time_range.start - t_end > 3:
        gaps.append((e.time_range.start - t_end, last, e.name))
    if t_end is None or e.time_range.end > t_end:
        t_end, last = e.time_range.end, e.name
busy = sum(e.time_range.end - e.time_range.start for e in ks)
span = ks[-1].time_range.end - ks[0].time_range.start
print(f"{route}: 3 steps: device span {span / 3e3:.2f} ms per step, kernel time {busy / 3e3:.2f}, gaps > 3 us: {sum(g[0] for g in gaps) / 3e3:.2f} ms per step in {len(gaps) // 3} places")
for g in sorted(gaps, key=lambda g: -g[0])[:36]:
    print(f"{g[0]:8.0f} us   after {g[1][:64]:64s} before {g[2][:64]}")
