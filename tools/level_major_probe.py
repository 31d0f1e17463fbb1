"""Would a LEVEL-MAJOR forward gather (every CU on the same level at the same time, so that the level's 4 MB table is what the
L2s hold) beat the sample-major one?  Times the stand-alone encoder op over the same 8.4e6 ray-ordered points: all 16 levels
at once against one level at a time (16 launches; outputs of one level each)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scanerf_amd  # noqa
from scanerf_amd import hashgrid as H
from scanerf_amd.hashgrid import level_resolutions

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, S, T = 65536, 128, 2 ** 19
o = (torch.rand(B, 1, 3, device=dev) - 0.5) * 2.0
d = torch.nn.functional.normalize(torch.randn(B, 1, 3, device=dev), dim=-1)
t = torch.linspace(0.0, 1.0, S, device=dev).reshape(1, S, 1)
pts = (o + d * t).clamp(-2, 2).reshape(-1, 3).contiguous()      # samples along random rays, ray-major (the training order)
N = pts.shape[0]
res = level_resolutions(torch.tensor([32, 32, 32]), torch.tensor([2048, 2048, 2048]), 16).to(dev).int().contiguous()
tab = (torch.randn(16, T, 2, device=dev) * 0.1).contiguous()
out16 = torch.zeros(N, 16, 2, device=dev)
out1 = torch.zeros(N, 1, 2, device=dev)

def timed(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

# round 6: the same with half-precision gather copies (2 MB per level against 4 MB of L2 per XCD): `level_major_probe.py f16 bf16`
for name in (sys.argv[1:] or ["f32"]):
    tb = tab.to({"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[name]).contiguous()
    r = {"table": name, "all_levels_ms": timed(lambda: H.embedding_bg_forward_cuda(pts, out16, tb, res))}
    per = []
    for l in range(16):
        tl, rl = tb[l:l + 1].contiguous(), res[l:l + 1].contiguous()
        per.append(timed(lambda: H.embedding_bg_forward_cuda(pts, out1, tl, rl)))
    r["per_level_ms"] = [round(x, 3) for x in per]
    r["sum_of_levels_ms"] = sum(per)
    print(json.dumps(r))
