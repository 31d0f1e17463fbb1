"""Image metrics of rendered views: PSNR (tools/utils.py:53-55) and SSIM (tools/ssim.py: 11 x 11 Gaussian window, sigma 1.5,
C1 = 0.01^2, C2 = 0.03^2), and the evaluation log RenderingHashGrid writes (rendering.py:205-267).  Torch ops on whatever device
the images live on (the Gaussian window is applied as two separable passes: same result, 11 + 11 taps instead of 121)."""
import math

import torch
import torch.nn.functional as F


def psnr(img1, img2):
    """10 log10(255^2 / (mse + 1e-8)) on 0..255 values."""
    mse = float(((img1.float() - img2.float()) ** 2).mean())
    return 10.0 * math.log10(255.0 ** 2 / (mse + 1e-8))


def _window(window_size, sigma, device, dtype):
    x = torch.arange(window_size, device=device, dtype=torch.float64)
    g = torch.exp(-((x - window_size // 2) ** 2) / (2.0 * sigma ** 2))
    return (g / g.sum()).to(dtype)


def _blur(x, w, channel):
    k = w.numel()
    x = F.conv2d(x, w.view(1, 1, k, 1).expand(channel, 1, k, 1), padding=(k // 2, 0), groups=channel)
    return F.conv2d(x, w.view(1, 1, 1, k).expand(channel, 1, 1, k), padding=(0, k // 2), groups=channel)


def ssim(img1, img2, window_size=11, size_average=True):
    """img1, img2: [N,C,H,W] in 0..1 -> mean SSIM (or one value per image)."""
    channel = img1.shape[1]
    w = _window(window_size, 1.5, img1.device, img1.dtype)
    mu1, mu2 = _blur(img1, w, channel), _blur(img2, w, channel)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _blur(img1 * img1, w, channel) - mu1_sq
    s2 = _blur(img2 * img2, w, channel) - mu2_sq
    s12 = _blur(img1 * img2, w, channel) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


def evaluate_views(render_fn, views, log_path=None):
    """rendering.py:205-267: render every (K, c2w, gt image [H,W,3] in 0..255) of `views`, log 'img i psnr .. ssim ..' lines and
    the means.  render_fn(H, W, K, c2w) -> rgb [H,W,3] in 0..1.  Returns (mean psnr, mean ssim, per-view list)."""
    rows = []
    for i, (K, c2w, gt) in enumerate(views):
        gt = torch.as_tensor(gt, dtype=torch.float32)
        H, W = gt.shape[:2]
        pred = render_fn(H, W, K, c2w).clamp(0, 1)
        gt = gt.to(pred.device)
        p = psnr(pred * 255.0, gt)
        s = float(ssim((gt / 255.0)[None].permute(0, 3, 1, 2), pred[None].permute(0, 3, 1, 2)))
        rows.append((i, p, s))
    mp = sum(r[1] for r in rows) / max(len(rows), 1)
    ms = sum(r[2] for r in rows) / max(len(rows), 1)
    if log_path is not None:
        with open(log_path, "w") as f:
            for i, p, s in rows:
                f.write(f"img {i} psnr {p:.2f}\tssim {s:.3f}\n")
            f.write(f"mean psnr {mp:.2f}\tmean ssim {ms:.3f}\n")
    return mp, ms, rows
