"""Cameras with learnable pose corrections -- the bundle-adjusting half of the method (camera.py:11-142 Pose / Lie algebra,
camera_utils.py:39-118 CAM) around the HIP ray kernels.

The reference generates rays in torch (camera.get_center_and_ray_v2) and lets autograd carry dL/drays back to
`se3_refine`; its CUDA_EXT also ships compute_ray_forward / compute_ray_backward (cuda/compute_ray_kernel.cu) for the same
job.  Here the per-ray part runs in those kernels' HIP counterparts: forward = one launch over (view, px, py) triples,
backward = a reduction of the ray gradients to dL/dC2W [C,3,4] (lanes of a view combined before 12 atomics per view);
only the tiny [C,6] -> [C,3,4] pose algebra stays in torch autograd.
"""
import torch
import torch.nn as nn

from .cuda import compute_ray_backward, compute_ray_forward


# ---- Lie algebra / pose composition (camera.py:37-59, :84-95, :110-142) ----------------------------------------------
def _series(x, first_denominator_pair):
    """sum_{i=0..10} (-1)^i x^(2i) / d_i with d_0 = prod(first pair) and d_i = d_{i-1} * (2i+a)(2i+b): the three
    truncated Taylor series the reference evaluates (sin x / x, (1-cos x)/x^2, (x-sin x)/x^3)."""
    a, b = first_denominator_pair
    ans = torch.zeros_like(x)
    denom = 1.0
    for i in range(11):
        if not (a == 0 and i == 0):
            denom *= (2 * i + a) * (2 * i + b)
        ans = ans + (-1) ** i * x ** (2 * i) / denom
    return ans


def taylor_A(x):
    return _series(x, (0, 1))   # 1, 2*3, 2*3*4*5, ...


def taylor_B(x):
    return _series(x, (1, 2))   # 1*2, 1*2*3*4, ...


def taylor_C(x):
    return _series(x, (2, 3))   # 2*3, 2*3*4*5, ...


# generators of so(3): hat(w) = w_0 E_0 + w_1 E_1 + w_2 E_2
_SO3_GENERATORS = torch.tensor([[[0.0, 0, 0], [0, 0, -1], [0, 1, 0]],
                                [[0, 0, 1], [0, 0, 0], [-1, 0, 0]],
                                [[0, -1, 0], [1, 0, 0], [0, 0, 0]]])


def skew_symmetric(w):
    """hat operator [...,3] -> [...,3,3] (camera.py:110-116) as a contraction with the so(3) generators."""
    return torch.einsum("...k,kij->...ij", w, _SO3_GENERATORS.to(device=w.device, dtype=w.dtype))


def se3_to_SE3(wu):
    """Exponential map se(3) -> SE(3), [...,6] = (rotation vector, translation part) -> [...,3,4] = [R | V u] with
    R = I + A K + B K^2 and V = I + B K + C K^2 for K = hat(w) (Rodrigues; camera.py:84-95), A, B, C the reference's
    truncated series in the rotation angle."""
    rot, trans = wu[..., :3], wu[..., 3:]
    K = skew_symmetric(rot)
    K2 = K @ K
    angle = rot.norm(dim=-1)[..., None, None]
    ident = torch.eye(3, device=wu.device, dtype=torch.float32)
    b = taylor_B(angle)
    R = ident + taylor_A(angle) * K + b * K2
    V = ident + b * K + taylor_C(angle) * K2
    return torch.cat([R, V @ trans[..., None]], dim=-1)


def pose_invert(pose):
    R, t = pose[..., :3], pose[..., 3:]
    R_inv = R.transpose(-1, -2)
    return torch.cat([R_inv, -R_inv @ t], dim=-1)


def pose_compose_pair(pose_a, pose_b):
    """x -> pose_b(pose_a(x))"""
    R_a, t_a = pose_a[..., :3], pose_a[..., 3:]
    R_b, t_b = pose_b[..., :3], pose_b[..., 3:]
    return torch.cat([R_b @ R_a, R_b @ t_a + t_b], dim=-1)


def pose_compose(pose_list):
    out = pose_list[0]
    for p in pose_list[1:]:
        out = pose_compose_pair(out, p)
    return out


# ---- rays through the HIP kernels ------------------------------------------------------------------------------------
class _Rays(torch.autograd.Function):
    """(C2Ws [C,3,4], Ks [C,3,3], locs [N,3] int32 (view, px, py)) -> rays_o, rays_d [N,3]; d is not normalised and uses the
    +0.5 pixel centre (cuda_utils.h:143-155 = camera.py:259-281)."""

    @staticmethod
    def forward(ctx, c2ws, ks, locs):
        C = c2ws.shape[0]
        c = c2ws.detach().reshape(C, 12).contiguous()
        k = ks.detach().reshape(C, 9).contiguous()
        o = torch.empty(locs.shape[0], 3, device=c.device)
        d = torch.empty(locs.shape[0], 3, device=c.device)
        compute_ray_forward(o, d, k, c, locs)
        ctx.save_for_backward(k, locs)
        ctx.C = C
        return o, d

    @staticmethod
    def backward(ctx, g_o, g_d):
        k, locs = ctx.saved_tensors
        g = torch.zeros(ctx.C, 12, device=k.device)
        compute_ray_backward(g_o.contiguous(), g_d.contiguous(), k, g, locs)
        return g.reshape(ctx.C, 3, 4), None, None


def rays_from_cameras(c2ws, ks, locs):
    return _Rays.apply(c2ws, ks, locs)


def pixel_locs(num_camera, ray_idx, W, device):
    """(view, px, py) triples of the same pixel set in every view -- the reference's batch layout [num_camera, len(ray_idx)]
    (tile.py:902-921): view-major."""
    ray_idx = ray_idx.to(device)
    v = torch.arange(num_camera, device=device)[:, None].expand(num_camera, ray_idx.numel())
    px = (ray_idx % W)[None, :].expand_as(v)
    py = torch.div(ray_idx, W, rounding_mode="floor")[None, :].expand_as(v)
    return torch.stack([v, px, py], -1).reshape(-1, 3).int().contiguous()


class CameraSet(nn.Module):
    """camera_utils.CAM: fixed intrinsics, world-to-camera start poses (optionally perturbed by `noise` [C,6]) and the
    learnable corrections se3_refine [C,6]; rts = se3_to_SE3(se3_refine) o rts0, c2w = rts^-1."""

    def __init__(self, ks, c2ws, device, noise=None):
        super().__init__()
        self.device = device
        c2ws = torch.as_tensor(c2ws, dtype=torch.float32)[..., :3, :4].to(device)
        self.num_camera = c2ws.shape[0]
        self.ori_rts = pose_invert(c2ws)
        self.se3_refine = nn.Parameter(torch.zeros(self.num_camera, 6, device=device))
        self.rts = self.ori_rts.clone() if noise is None else pose_compose([se3_to_SE3(noise.to(device)), self.ori_rts.clone()])
        self.ks = torch.as_tensor(ks, dtype=torch.float32).reshape(self.num_camera, 3, 3).to(device)

    def get_rts(self, se3_refine=None):
        return pose_compose([se3_to_SE3(self.se3_refine if se3_refine is None else se3_refine), self.rts])

    def get_poses(self):
        return pose_invert(self.get_rts())

    def get_rays(self, locs):
        """locs [N,3] int32 (view, px, py) -> rays_o, rays_d [N,3], differentiable w.r.t. se3_refine."""
        return rays_from_cameras(self.get_poses(), self.ks, locs)

    def get_rays_idx(self, W, ray_idx):
        return self.get_rays(pixel_locs(self.num_camera, ray_idx, W, self.device))
