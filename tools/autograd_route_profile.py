"""The autograd route (hashgrid.HashGrid.render_fore_rays -> render.FusedRenderRays, a torch loss, loss.backward(), torch Adam on the
decoder module, adam_step_cuda on the dense table gradient) at configs[1]'s size: ms per step, and under
`rocprofv3 --kernel-trace --stats -- python3 tools/autograd_route_profile.py` the kernel table (what the glue around the three fused
kernels costs)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

a = argparse.Namespace(log2_T=19)
dev = "cuda:0"
B, S = 65536, 128
torch.manual_seed(0)
o = torch.rand(B, 3, device=dev) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1) * (0.5 + torch.rand(B, 1, device=dev))
tgt = torch.rand(B, 3, device=dev)
r = bench.autograd_route_leg(a, dev, o, d, tgt, S, 20000, n=6)
print(f"autograd route: {r['autograd_route_ms_per_step']:.2f} ms per step")
