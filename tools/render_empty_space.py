"""What the zero-opacity gate of the render-time decoder is worth on empty space: configs[4]'s frame with every tile's density head
forced to -300 (softplus -> 0: every sample's opacity is exactly zero, every tile skips its directional layers) beside the ordinary
frame.  Usage: python tools/render_empty_space.py"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scanerf_amd  # noqa
from scanerf_amd import renderer as R, tile_model as tm

dev = torch.device("cuda:0")
H, W, ntile = 1080, 1920, 4
for empty in (False, True):
    tiles = []
    with tempfile.TemporaryDirectory() as tmp:
        for t in range(ntile):
            m = tm.TileModel([-4.0 * ntile + 8.0 * t, -4, -4], [8, 8, 8], dev, log2_T=19, seed=t, sampler_log2dim=7)
            m.set_occupancy(tm.sphere_shell_occupancy(m, 3.0, 0.5))
            with torch.no_grad():
                m.features.mul_(300.0)
                if empty:
                    m.decoder.sigma_layer_mlp_0_weight.zero_()
                    m.decoder.sigma_layer_mlp_0_bias.fill_(-300.0)
            R.export_tile(os.path.join(tmp, f"tile{t}"), m)
            tiles.append(R.load_tile(os.path.join(tmp, f"tile{t}")))
            del m
    rend = R.TileSetRenderer(dev, tiles)
    K = [1600.0, 0, W / 2, 0, 1600.0, H / 2, 0, 0, 1]
    c2w = torch.tensor([[1.0, 0, 0, 0.0], [0, 1, 0, 0.5], [0, 0, 1, -14.0]])
    for _ in range(2):
        out = rend.render(H, W, K, c2w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = rend.render(H, W, K, c2w)
    torch.cuda.synchronize()
    print("zero opacity everywhere" if empty else "ordinary scene", f"{(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per frame; max |colour| {float(out[0].abs().max()):.3g}")
    del rend, tiles
    torch.cuda.empty_cache()
