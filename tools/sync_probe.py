"""Which lines of a step stall the host until the device has drained (torch.cuda.set_sync_debug_mode("warn") + the Python stack):
`autograd` (HashGrid.render_fore_rays + torch loss), `fused` (train_step_fused, configs[1]), `fgbg` (train_step_fgbg with poses),
`render` (TileSetRenderer.render)."""
import os, sys, warnings, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scanerf_amd  # noqa
from scanerf_amd import network
dev = "cuda:0"
what = sys.argv[1] if len(sys.argv) > 1 else "autograd"
B, S = (16384 if what == "render" else 65536), 128
torch.manual_seed(0)
o = torch.rand(B, 3, device=dev) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
tgt = torch.rand(B, 3, device=dev)
if what == "autograd":
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
elif what in ("fused", "fgbg"):
    from scanerf_amd import tile_model as tm
    m = tm.TileModel([-4.0, -4, -4], [8, 8, 8], dev, log2_T=19, seed=24, sampler_log2dim=4)
    opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    step = (lambda i: tm.train_step_fused(m, opt, o, d, tgt, S, 20000 + i)) if what == "fused" else \
           (lambda i: tm.train_step_fgbg(m, opt, o, d, tgt, S, S, 20000 + i, pose_grads=True))
else:
    import tempfile
    from scanerf_amd import renderer as R, tile_model as tm
    tiles = []
    with tempfile.TemporaryDirectory() as tmp:
        for t in range(2):
            mm = tm.TileModel([-8.0 + 8.0 * t, -4, -4], [8, 8, 8], dev, log2_T=15, seed=t, sampler_log2dim=5)
            mm.set_occupancy(tm.sphere_shell_occupancy(mm, 3.0, 0.5))
            R.export_tile(os.path.join(tmp, f"t{t}"), mm)
            tiles.append(R.load_tile(os.path.join(tmp, f"t{t}")))
    rend = R.TileSetRenderer(dev, tiles)
    K = [200.0, 0, 64, 0, 200.0, 64, 0, 0, 1]
    c2w = torch.tensor([[1.0, 0, 0, 0.0], [0, 1, 0, 0.5], [0, 0, 1, -14.0]])
    step = lambda i: rend.render(128, 128, K, c2w, num_sample=64, num_bg_sample=64)
for i in range(2):
    step(i)
torch.cuda.synchronize()


def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "scanerf" in f.filename and "sync_probe" not in f.filename]
    print("SYNC <-", " | ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-4:]))


warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
step(2)
torch.cuda.set_sync_debug_mode("default")
