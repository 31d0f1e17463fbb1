"""End-to-end learning check on a procedural scene (no dataset ships with the reference): a shaded sphere in an empty tile.
Trains one tile with the fused kernels (TileTrainer: schedulers, sparse Adam, pruning) on random rays and reports the PSNR
(tools/utils.py:53-55 definition) of held-out rays before and after.

    python tools/train_demo.py [--steps 400] [--rays 16384] [--log2-T 19]
"""
import argparse
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def sphere_scene(o, d, radius=2.5):
    """RGB of rays against a sphere at the origin: 0.5 + 0.5 * normal where hit, black elsewhere."""
    dn = torch.nn.functional.normalize(d, dim=-1)
    b = (o * dn).sum(-1)
    c = (o * o).sum(-1) - radius * radius
    disc = b * b - c
    hit = (disc > 0) & (c > 0) & (-b - torch.sqrt(disc.clamp(min=0)) > 0)  # outside the sphere, looking at it
    t = -b - torch.sqrt(disc.clamp(min=0))
    n = torch.nn.functional.normalize(o + t[:, None] * dn, dim=-1)
    return torch.where(hit[:, None], 0.5 + 0.5 * n, torch.zeros_like(n))


def random_rays(B, dev, gen):
    """origins on a shell outside the sphere but inside the tile, directions towards a jittered point near the centre"""
    o = torch.nn.functional.normalize(torch.randn(B, 3, device=dev, generator=gen), dim=-1) * (3.2 + 0.6 * torch.rand(B, 1, device=dev, generator=gen))
    look = (torch.rand(B, 3, device=dev, generator=gen) - 0.5) * 6.0
    return o.contiguous(), torch.nn.functional.normalize(look - o, dim=-1).contiguous()


def psnr(a, b):
    mse = float(((a * 255.0 - b * 255.0) ** 2).mean())
    return 10.0 * math.log10(255.0 ** 2 / (mse + 1e-8))


def run(steps=400, rays=16384, log2_T=19, samples=128, dev="cuda:0", verbose=True):
    import scanerf_amd  # noqa: F401
    from scanerf_amd import trainer
    from scanerf_amd.tile_model import TileModel
    gen = torch.Generator(device=dev).manual_seed(0)
    model = TileModel([-4, -4, -4], [8, 8, 8], dev, log2_T=log2_T, seed=0)
    test_o, test_d = random_rays(8192, dev, gen)
    test_rgb = sphere_scene(test_o, test_d)

    def evaluate(step):
        out, _, valid = model.render_fore_fused(test_o, test_d, samples, step)
        return psnr(out[:, 0:3], test_rgb)

    def batch(step):
        o, d = random_rays(rays, dev, gen)
        return o, d, sphere_scene(o, d)

    tr = trainer.TileTrainer(model, batch, total_step=steps, eta_hash=1e-2, eta_decoder=1e-3, grid_log2dim=(4, 5, 6),
                             pruning_th=(0.01, 0.02), adjust_step=max(steps // 3, 1), num_sample=samples)
    p0 = evaluate(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = []
    tr.train(steps, on_step=lambda s, l: losses.append(l))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p1 = evaluate(steps)
    losses = [float(l) for l in losses]
    if verbose:
        print(f"procedural sphere, {steps} steps x {rays} rays x {samples} samples, T=2^{log2_T}: "
              f"held-out PSNR {p0:.2f} -> {p1:.2f} dB, loss {losses[0]:.4f} -> {losses[-1]:.4f}, "
              f"{dt / steps * 1e3:.2f} ms/step ({rays * steps / dt:.3e} rays/s incl. ray generation), "
              f"occupied cells {int(model.occupied_grid.sum())}/{model.occupied_grid.numel()} at log2dim {model.log2dim.tolist()}")
    return p0, p1, losses


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--rays", type=int, default=16384)
    ap.add_argument("--log2-T", type=int, default=19)
    a = ap.parse_args()
    run(a.steps, a.rays, a.log2_T)
