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


def run(steps=400, rays=16384, log2_T=19, samples=128, dev="cuda:0", verbose=True, return_model=False, arith=None):
    """arith: None = the library's default; "f32" / "h3" / "t16" / ... = train under that arithmetic (render.set_arith)."""
    import scanerf_amd  # noqa: F401
    from scanerf_amd import render, trainer
    from scanerf_amd.tile_model import TileModel
    if arith is not None:
        render.set_arith(arith)
        try:
            return run(steps, rays, log2_T, samples, dev, verbose, return_model)
        finally:
            render.set_arith(render.DEFAULT_ARITH)
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
        print(f"[{render.arith_name()}] procedural sphere, {steps} steps x {rays} rays x {samples} samples, T=2^{log2_T}: "
              f"held-out PSNR {p0:.2f} -> {p1:.2f} dB, loss {losses[0]:.4f} -> {losses[-1]:.4f}, "
              f"{dt / steps * 1e3:.2f} ms/step ({rays * steps / dt:.3e} rays/s incl. ray generation), "
              f"occupied cells {int(model.occupied_grid.sum())}/{model.occupied_grid.numel()} at log2dim {model.log2dim.tolist()}")
    return (p0, p1, losses, model) if return_model else (p0, p1, losses)


def novel_view_check(model, H=120, W=160, samples=128, step=40000, dev="cuda:0"):
    """Train-time renderer vs export -> multi-tile render-time renderer on one novel view of the trained tile, and both
    against the procedural ground truth: -> dict of PSNR / SSIM figures."""
    import tempfile

    from scanerf_amd import cameras as CM
    from scanerf_amd import metrics as MT
    from scanerf_amd import renderer as R
    c2w = torch.tensor([[1.0, 0, 0, 0.3], [0, 1, 0, -0.2], [0, 0, 1, -3.7]])
    K = torch.tensor([[110.0, 0, W / 2], [0, 110.0, H / 2], [0, 0, 1]])
    cams = CM.CameraSet(K[None], c2w[None], dev)
    locs = CM.pixel_locs(1, torch.arange(H * W), W, dev)
    with torch.no_grad():
        o, d = cams.get_rays(locs)
    gt = sphere_scene(o, d).reshape(H, W, 3)
    with torch.no_grad():
        train_img = model.render_rays_fused(o.contiguous(), d.contiguous(), samples, samples, step)["pred_color"].reshape(H, W, 3).clamp(0, 1)
    with tempfile.TemporaryDirectory() as tmp:
        R.export_tile(os.path.join(tmp, "tile-0"), model)
        rend = R.TileSetRenderer(dev, [R.load_tile(os.path.join(tmp, "tile-0"))])
    dif, spec, depth, transp = rend.render(H, W, K.reshape(-1).tolist(), c2w, num_sample=samples, num_bg_sample=samples)
    rt_img = (dif + spec).clamp(0, 1)
    nchw = lambda im: im[None].permute(0, 3, 1, 2)
    return {"psnr_train_vs_gt": MT.psnr(train_img * 255, gt * 255), "psnr_render_vs_gt": MT.psnr(rt_img * 255, gt * 255),
            "psnr_render_vs_train": MT.psnr(rt_img * 255, train_img * 255), "ssim_render_vs_gt": float(MT.ssim(nchw(rt_img), nchw(gt))),
            "ssim_render_vs_train": float(MT.ssim(nchw(rt_img), nchw(train_img)))}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--rays", type=int, default=16384)
    ap.add_argument("--log2-T", type=int, default=19)
    ap.add_argument("--arith", default=None, help="f32 | h3 | t16 | ...: training arithmetic (default: the library's)")
    ap.add_argument("--render", action="store_true", help="also export the tile and render a novel view through the render-time path")
    a = ap.parse_args()
    res = run(a.steps, a.rays, a.log2_T, return_model=a.render, arith=a.arith)
    if a.render:
        print("novel view 160x120:", {k: round(v, 3) for k, v in novel_view_check(res[3], step=max(a.steps, 10000)).items()})
