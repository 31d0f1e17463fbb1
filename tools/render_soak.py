"""The same 1920x1080 view of a 4-tile scene rendered N times (2.7e8 foreground + 2.7e8 background sample slots per frame): every
frame must equal the first bit for bit (no atomics on the render-time path; a difference is a hazard the generated code does not
cover).  A scratch buffer is rewritten between frames so that the tables are not always warm."""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd  # noqa
from scanerf_amd import _capi
from scanerf_amd import renderer as R
from scanerf_amd.tile_model import TileModel
dev = "cuda:0"
H, W, N = int(os.environ.get("H", 1080)), int(os.environ.get("W", 1920)), int(os.environ.get("N", 20))
torch.manual_seed(0)
tiles = []
for ix in range(2):
    for iz in range(2):
        m = TileModel([-7 + 6 * ix, -4, -7 + 6 * iz], [8, 8, 8], dev, log2_T=19, seed=ix * 2 + iz, sampler_log2dim=6)
        with torch.no_grad():
            m.features.mul_(100.0)
            m.decoder.sigma_layer_mlp_0_bias.add_(2.0)
        g = torch.rand(tuple(m.occupied_grid.shape), device=dev)
        yy = torch.arange(g.shape[1], device=dev)[None, :, None]
        m.occupied_grid = (g < 0.15) & (yy < g.shape[1] // 2)
        tiles.append({"features": m.features.detach().cpu().numpy().astype(np.float16), "occupied_grid": m.occupied_grid.cpu().numpy(),
                      "block_corner": m.min_bbox.numpy(), "block_size": m.bbox_size.numpy(), "grid_log2dim": m.log2dim.cpu().numpy(),
                      "resolution": m.resolution.cpu().numpy(), "blob": m.decoder.blob().detach().cpu().numpy()})
        del m
rnd = R.TileSetRenderer(dev, tiles)
K = np.float32([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]])
c2w = np.float32([[1, 0, 0, 0.5], [0, 0.94, -0.34, 3.0], [0, 0.34, 0.94, -14.0]])
scratch = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
ref, bad = None, 0
for i in range(N):
    scratch.fill_(i & 255)
    _capi.SWEEP_ICACHE = os.environ.get("SWEEP", "0") == "1" and i > 0   # instruction caches swept after every library call
    out = rnd.render(H, W, K, c2w)
    _capi.SWEEP_ICACHE = False
    torch.cuda.synchronize()
    if ref is None:
        ref = [t.clone() for t in out]
        continue
    d = [int((a != b).sum()) for a, b in zip(out, ref)]
    if any(d):
        bad += 1
        print(f"  frame {i}: differing elements (diffuse, specular, depth, transparency) = {d}", flush=True)
print(f"{bad} of {N - 1} frames differ from frame 0")
