"""Golden vector G17 from the reference's camera module (round 5).

    python tests/golden/make_golden_cameras.py          (build container only: /root/reference must exist)

G17  camera_utils.CAM (camera_utils.py:39-118) with pose noise and non-zero corrections: getRays(H, W, ray_idx) -> rays_o, rays_d
     [num_camera, len(ray_idx), 3] (camera.get_center_and_ray_v2, +0.5 pixel centre, directions not normalised), get_poses(), and
     -- through the reference's own torch autograd -- d(loss)/d(se3_refine) of loss = sum(w_o * rays_o + w_d * rays_d): pins the
     adjoint the HIP ray kernels implement (compute_ray_backward, cuda/compute_ray_kernel.cu:46-92, whose CUDA body has an indexing
     bug this repo documents and does not copy) against the graph the reference's training actually differentiates.
Only DATA is written (inputs + the reference's outputs); nothing here runs on the GPU box."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, _stub_modules  # noqa: E402

sys.dont_write_bytecode = True


def main():
    _stub_modules()
    cv2 = types.ModuleType("cv2")   # camera_utils.py:1 imports one unused name from it
    cv2.norm = lambda *a, **k: None
    sys.modules["cv2"] = cv2
    sys.path.insert(0, REF)
    import camera_utils  # noqa
    torch.manual_seed(17)
    C, H, W = 4, 12, 16
    ks = torch.tensor([[90.0, 0, W / 2 + 0.3, 0, 95.0, H / 2 - 0.2, 0, 0, 1]]).repeat(C, 1).reshape(C, 3, 3)
    ks[1, 0, 0], ks[2, 1, 2] = 70.0, 4.4
    c2ws = torch.cat([torch.linalg.qr(torch.randn(C, 3, 3))[0], torch.randn(C, 3, 1) * 2], -1)
    noise = torch.randn(C, 6) * 0.05
    cam = camera_utils.CAM(ks, c2ws, "cpu", noise)
    with torch.no_grad():
        cam.se3_refine.copy_(torch.randn(C, 6) * 0.03)
    ray_idx = torch.tensor([0, 5, 17, 63, 100, 191])
    w_o, w_d = torch.randn(C, ray_idx.numel(), 3), torch.randn(C, ray_idx.numel(), 3)
    ro, rd = cam.getRays(H, W, ray_idx)
    loss = (ro * w_o).sum() + (rd * w_d).sum()
    loss.backward()
    out = {"ks": ks, "c2ws": c2ws, "noise": noise, "se3_refine": cam.se3_refine.detach(), "ray_idx": ray_idx, "H": np.array(H),
           "W": np.array(W), "rays_o": ro.detach(), "rays_d": rd.detach(), "poses": cam.get_poses().detach(), "w_o": w_o, "w_d": w_d,
           "loss": loss.detach(), "grad_se3_refine": cam.se3_refine.grad}
    out = {k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()}
    np.savez_compressed(os.path.join(HERE, "g17_cam_rays.npz"), **out)
    print("wrote g17_cam_rays", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
