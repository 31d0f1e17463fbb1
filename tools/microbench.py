"""Ad-hoc kernel timings at BASELINE config 2 (not the judged bench; see bench.py)."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd  # noqa
from scanerf_amd import network, render
from scanerf_amd._capi import check, lib, stream
from scanerf_amd.cuda import sample_points_grid
from scanerf_amd.hashgrid import level_resolutions

dev = "cuda:0"
B, S, T = 65536, 128, 2 ** 19
torch.manual_seed(0)


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


corner = torch.tensor([-4.0, -4, -4], device=dev)
size = torch.tensor([8.0, 8, 8], device=dev)
o = (torch.rand(B, 3, device=dev) * 8 - 4)
d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1) * (0.5 + torch.rand(B, 1, device=dev))
for l2 in (4, 7):
    occ = torch.ones((2 ** l2,) * 3, dtype=torch.bool, device=dev)
    l2d = torch.tensor([l2] * 3, dtype=torch.int32, device=dev)
    z = torch.full((B, S), -1.0, device=dev)
    dist = torch.full((B, S), -1.0, device=dev)
    ms = timeit(lambda: sample_points_grid(o, d, z, dist, corner, size, occ, l2d))
    print(f"sample_points_grid log2dim={l2}: {ms:.3f} ms  ({B / ms * 1e3:.3e} rays/s)")

res = level_resolutions(torch.tensor([32, 32, 32]), torch.tensor([2048, 2048, 2048]), 16).to(dev)
feat = torch.randn(16, T, 2, device=dev) * 0.1
N = B * S
pts = (torch.rand(N, 3, device=dev) * 2 - 1).contiguous()
out = torch.zeros(N, 16, 2, device=dev)
for dt, code in ((torch.float32, 0), (torch.bfloat16, 2)):
    F = feat.to(dt).contiguous()
    for variant, lm in ((1, 0), (1, 1), (2, 0)):
        fn = lambda: check(lib().scanerf_embedding_bg_forward_ex(
            ctypes.c_void_p(pts.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(F.data_ptr()),
            ctypes.c_void_p(res.data_ptr()), N, 16, T, code, variant, lm, stream()), "e")
        ms = timeit(fn, n=5, warm=2)
        gathers = N * 16 * 8
        print(f"embed fwd {dt} variant={variant} level_major={lm}: {ms:.3f} ms  {gathers / ms / 1e6:.1f} Ggather/s  "
              f"alg {gathers * (8 if code == 0 else 4) / ms / 1e9:.2f} TB/s  ({B / ms * 1e3:.3e} rays/s)")

blob = network.xavier_blob(0, dev)
pk = render.PackedDecoder(dev).pack(blob, network.weight_feature(40000, dev))
occ = torch.ones((16,) * 3, dtype=torch.bool, device=dev)
l2d = torch.tensor([4] * 3, dtype=torch.int32, device=dev)
sample_points_grid(o, d, z, dist, corner, size, occ, l2d)
valid = torch.all(z != -1, dim=-1)
print("valid rays", int(valid.sum()))
out_ray = torch.empty(B, 16, device=dev)
w = torch.empty(B, S, device=dev)
for dt in (torch.float32, torch.bfloat16):
    F = feat.to(dt).contiguous()
    ms = timeit(lambda: render.render_forward(o, d, z, dist, F, res, pk, [-8.0] * 3, [16.0] * 3, render.FORE, False,
                                              ray_valid=valid, out_ray=out_ray, weights=w), n=5, warm=2)
    nv = int(valid.sum())
    print(f"render_forward {dt}: {ms:.3f} ms  {nv / ms * 1e3:.3e} valid rays/s  MLP {nv * S * 27456 / ms / 1e9:.1f} TFLOP/s  "
          f"gather alg {nv * S * 128 * (8 if dt == torch.float32 else 4) / ms / 1e9:.2f} TB/s")
