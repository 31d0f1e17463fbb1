"""Novel-view render time of the multi-tile renderer (secondary metric: rendering.py:270 prints ms/frame
at 1280x720 on the reference's V100; no published number).  Synthetic scene: 2x2 tiles of 8 m, 2 m overlap."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd
from scanerf_amd import renderer as R
from scanerf_amd.tile_model import TileModel

dev = "cuda:0"
H, W = int(os.environ.get("H", 720)), int(os.environ.get("W", 1280))
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
        m.occupied_grid = (g < 0.15) & (yy < g.shape[1] // 2)      # a sparse "ground" layer
        tiles.append({"features": m.features.detach().cpu().numpy().astype(np.float16), "occupied_grid": m.occupied_grid.cpu().numpy(),
                      "block_corner": m.min_bbox.numpy(), "block_size": m.bbox_size.numpy(), "grid_log2dim": m.log2dim.cpu().numpy(),
                      "resolution": m.resolution.cpu().numpy(), "blob": m.decoder.blob().detach().cpu().numpy()})
        del m
rnd = R.TileSetRenderer(dev, tiles)
K = np.float32([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]])
c2w = np.float32([[1, 0, 0, 0.5], [0, 0.94, -0.34, 3.0], [0, 0.34, 0.94, -14.0]])  # above the ground, looking along +z, tilted down
for _ in range(2):
    out = rnd.render(H, W, K, c2w)
torch.cuda.synchronize()
t0 = time.time(); n = 5
for _ in range(n):
    out = rnd.render(H, W, K, c2w)
torch.cuda.synchronize()
ms = (time.time() - t0) / n * 1e3
T = out[3]
print(f"render {W}x{H}, {len(tiles)} tiles, 128+128 samples: {ms:.1f} ms/frame  ({H*W/ms*1e3:.3e} rays/s)  "
      f"opaque pixels {(T < 0.5).float().mean().item():.2f}")
