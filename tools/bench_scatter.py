"""Time the binned scatter's kernels at config 2 on ray-coherent sample points."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd
from scanerf_amd._capi import check, lib, stream, workspace
from scanerf_amd.hashgrid import level_resolutions
dev = "cuda:0"
B, S, T, L = 65536, 128, 2 ** 19, 16
torch.manual_seed(0)
o = torch.rand(B, 3, device=dev) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
z = torch.sort(torch.rand(B, S, device=dev) * 6, dim=-1)[0]
pts = ((o[:, None] + z[..., None] * d[:, None]).reshape(-1, 3) + 8) / 16 * 4 - 2
pts = pts.clamp(-2, 2).contiguous()
N = B * S
res = level_resolutions(torch.tensor([32] * 3), torch.tensor([2048] * 3), 16).to(dev)
gin = torch.randn(N, L, 2, device=dev)
gf = torch.zeros(L, T, 2, device=dev)
need = lib().scanerf_embedding_bwd_workspace_bytes(N, L, T)
ws = workspace(dev, need)
def run():
    check(lib().scanerf_embedding_bg_backward_binned(ctypes.c_void_p(pts.data_ptr()), ctypes.c_void_p(gin.data_ptr()),
          ctypes.c_void_p(gf.data_ptr()), ctypes.c_void_p(res.data_ptr()), N, L, T, 0, ctypes.c_void_p(ws.data_ptr()),
          ctypes.c_size_t(ws.numel()), stream()), "b")
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
print(f"mode={os.environ.get('SCANERF_DEBUG_ACC_MODE','0')} binned scatter total: {e0.elapsed_time(e1)/5:.2f} ms")
