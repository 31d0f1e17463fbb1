import os, sys, warnings, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scanerf_amd  # noqa
from scanerf_amd import network
from scanerf_amd.cuda import adam_step_cuda
from scanerf_amd.hashgrid import HashGrid
dev = "cuda:0"
B, S = 65536, 128
torch.manual_seed(0)
o = torch.rand(B, 3, device=dev) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
tgt = torch.rand(B, 3, device=dev)
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
for i in range(2): step(i)
torch.cuda.synchronize()
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "scanerf" in f.filename or "sync_probe" in f.filename]
    print("SYNC:", str(message)[:60], " <- ", " | ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-4:]))
warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
step(2)
torch.cuda.set_sync_debug_mode("default")
