"""Golden vectors for the image metrics (PSNR / SSIM of rendered views).

Run ONCE in the build container (where /root/reference exists):

    python tests/golden/make_golden_metrics.py

Loads the reference's tools/ssim.py (pure torch) by path, feeds it seeded image pairs and writes inputs + outputs to
tests/golden/g11_ssim.npz.  Only DATA is written.  Nothing here runs on the GPU box.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def main():
    spec = importlib.util.spec_from_file_location("ref_ssim", "/root/reference/tools/ssim.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    torch.manual_seed(0)
    a = torch.rand(2, 3, 40, 56)
    b = (a + 0.1 * torch.randn_like(a)).clamp(0, 1)
    smooth = torch.nn.functional.avg_pool2d(a, 5, 1, 2)
    m = ref.SSIM(window_size=11)
    out = {"a": a, "b": b, "smooth": smooth, "ssim_ab": m(a, b), "ssim_aa": m(a, a), "ssim_asmooth": m(a, smooth),
           "ssim_ab_per_image": ref.ssim(a, b, 11, size_average=False)}
    # tools/utils.py:53-55: psnr on 0..255 values
    i1, i2 = a[0].permute(1, 2, 0).numpy() * 255.0, b[0].permute(1, 2, 0).numpy() * 255.0
    out["psnr_ab0"] = np.array(10 * float(np.log10(255.0 ** 2 / (np.mean((i1 - i2) ** 2) + 1e-8))))
    np.savez_compressed(os.path.join(HERE, "g11_ssim.npz"),
                        **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()})
    print({k: (float(v) if np.asarray(v).size == 1 else np.asarray(v).shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
