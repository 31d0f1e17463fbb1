"""Time the op-by-op route -- what a caller that keeps hashgrid/__init__.py's render_batch_rays unchanged runs: sampler op,
encoder op (autograd wrapper), decoder MODULE (network.ShallowMLP -> csrc/decoder.hip), torch compositing, loss.backward(),
torch Adam on the decoder, adam_step_cuda on the table -- at configs[1]'s size, next to the fused step.
    python tools/ops_path_profile.py [rays] [samples] [torch|hip]
Under `rocprofv3 --kernel-trace --stats -- python3 tools/ops_path_profile.py` the kernel table shows where the time goes."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scanerf_amd  # noqa: E402,F401
from scanerf_amd import network  # noqa: E402
from scanerf_amd.cuda import adam_step_cuda  # noqa: E402
from scanerf_amd.hashgrid import HashGrid  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dec_kind = sys.argv[3] if len(sys.argv) > 3 else "hip"
dev = "cuda:0"
torch.manual_seed(0)
hg = HashGrid(dev, torch.tensor([-4.0, -4, -4]), torch.tensor([8.0, 8, 8]), log2_hashmap_size=19, grid_resolution=[32, 2048], sampler_log2dim=4)
hg.fused = False
dec = network.init_model(network.ShallowMLP(32), "xavier").to(dev)
dec.use_hip = dec_kind == "hip"
opt = torch.optim.Adam(dec.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
exp_avg, exp_avg_sq = torch.zeros_like(hg.HE.features), torch.zeros_like(hg.HE.features)
o = torch.rand(B, 3, device=dev) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1) * (0.5 + torch.rand(B, 1, device=dev))
tgt = torch.rand(B, 3, device=dev)
K = hg.HE.features.numel() // 8


def step(i, marks=None):
    def mark(name):
        if marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((name, e))
    mark("start")
    hg.HE.features.grad = None
    opt.zero_grad(set_to_none=True)
    out, ok = hg.render_fore_rays(o, d, S, dec, 0, global_step=20000 + i)
    mark("forward")
    loss = torch.nn.functional.mse_loss(out["pred_color"], tgt) + 0.01 * out["l2_reg_specular"]
    loss.backward()
    mark("backward")
    with torch.no_grad():
        adam_step_cuda(hg.HE.features.data.view(K, 8), hg.HE.features.grad.view(K, 8), exp_avg.view(K, 8), exp_avg_sq.view(K, 8),
                       1e-2, 0.9, 0.99, 1e-15, i)
    opt.step()
    mark("optimisers")
    return loss


for i in range(2):
    step(i)
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for i in range(n):
    step(2 + i)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
marks = []
step(9, marks)
torch.cuda.synchronize()
parts = {b[0]: a[1].elapsed_time(b[1]) for a, b in zip(marks[:-1], marks[1:])}
print(f"op-by-op route ({dec_kind} decoder), {B} rays x {S} samples: {ms:.2f} ms per step; sections (ms): " +
      ", ".join(f"{k} {v:.2f}" for k, v in parts.items()))
