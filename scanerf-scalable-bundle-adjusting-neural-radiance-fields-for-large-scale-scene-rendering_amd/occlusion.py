"""Shared-depth occlusion exchange between overlapping tiles (tile.py:366-475).

The reference's second exchange, next to the camera consensus: a tile that CONTAINS an overlap camera renders a
half-resolution depth map of that view (render_shared_depth, tile.py:436-471) and publishes it through the master
process's shared dictionary; every other tile that sees the camera from OUTSIDE its box masks the pixels whose shared
depth lies beyond the entry of its own box -- something nearer tiles already explain -- and dilates the kept region
with a 91 x 91 box filter (update_occlusion_mask, tile.py:366-400).

Here the depth maps live on the GPU, the renders go through the fused forward kernels, box entry distances through
the HIP ray_aabb_intersection, and the exchange between ranks is a collective: each camera's map is published by at
most one tile (the one containing the camera), so a dense [N_cam, H/2, W/2] buffer initialised to +inf and reduced with
all_reduce(MIN) (RCCL over xGMI) delivers every map to every rank -- no master process, no pickling.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F

from .cuda import ray_aabb_intersection
from .consensus import collective_active

NO_DEPTH = float("inf")


def camera_inside(cam_center, bbox_center, box_size):
    """tile.py:388 / :450: |o - centre| < size/2 on every axis, size = featureGrid.bbox_size / 2 (the tile itself)."""
    return bool(torch.all(torch.abs(cam_center - bbox_center) < (box_size / 2.0)))


@torch.no_grad()
def render_depth_rays(model, rays_o, rays_d, S_fg, S_bg, global_step, batch_size=2 ** 16):
    """tile.py:714-722 on the fused kernels: merged fg + T_left * bg depth per ray, zeros where nothing renders."""
    depth = torch.zeros_like(rays_o[..., :1])
    for i in range(0, rays_o.shape[0], batch_size):
        out = model.render_rays_fused(rays_o[i:i + batch_size].contiguous(), rays_d[i:i + batch_size].contiguous(), S_fg, S_bg,
                                      global_step)
        depth[i:i + batch_size] = out["pred_depth"]
    return depth


@torch.no_grad()
def render_shared_depth(model, get_rays, H, W, visible_poses, overlap_idxs, shared_depth, S_fg=128, S_bg=128, global_step=40000):
    """tile.py:436-471.  get_rays(local_view) -> rays_o, rays_d [H*W,3]; visible_poses: global camera id per local view;
    overlap_idxs: local views shared with other tiles; shared_depth: [N_cam, H/2, W/2] device buffer (NO_DEPTH = not
    published).  Only views whose camera lies inside this tile are rendered, at every second pixel."""
    center, size = model._center_dev, model._half_dev
    overlap = set(int(i) for i in (overlap_idxs.tolist() if torch.is_tensor(overlap_idxs) else overlap_idxs))
    published = []
    for idx, ori_idx in enumerate(visible_poses):
        if idx not in overlap:
            continue
        rays_o, rays_d = get_rays(idx)
        o = rays_o.reshape(H, W, 3)[::2, ::2].reshape(-1, 3).contiguous()
        d = rays_d.reshape(H, W, 3)[::2, ::2].reshape(-1, 3).contiguous()
        if not camera_inside(o[0], center, size):
            continue
        shared_depth[ori_idx] = render_depth_rays(model, o, d, S_fg, S_bg, global_step).reshape((H + 1) // 2, (W + 1) // 2)
        published.append(int(ori_idx))
    return published


@torch.no_grad()
def exchange_shared_depth(shared_depth, published=None, group=None):
    """One round of the exchange: every rank ends up with every map published THIS round; entries nobody re-published keep
    what an earlier round delivered (the reference replaces a camera's entry in the master's dictionary when a new map
    arrives and leaves the others alone, tile.py:436-471 / admm_trainer.py shared dict).

    published: global camera ids this rank wrote into `shared_depth` since the last exchange (None = every finite entry,
    for a first or only round).  Only those rows enter the collective -- a fresh +inf buffer reduced with MIN (one
    publisher per camera, so MIN is a gather).  Reducing the persistent buffer itself would mix rounds: ranks that hold a
    camera's OLD map would contribute it again and the result would be min(old, new), so depths could only ever decrease.
    Every rank must call this the same number of times (it is a collective): AdmmDriver does, once per stretch."""
    multi = collective_active(group)   # (world size 1 included: the same collective path everywhere; members of `group` only)
    if published is None and not multi:
        return shared_depth
    fresh = torch.full_like(shared_depth, NO_DEPTH)
    if published is None:
        fresh.copy_(shared_depth)
    elif len(published):
        idx = torch.as_tensor(sorted(set(int(i) for i in published)), dtype=torch.long, device=shared_depth.device)
        fresh[idx] = shared_depth[idx]
    if multi:
        dist.all_reduce(fresh, op=dist.ReduceOp.MIN, group=group)
    arrived = torch.isfinite(fresh).reshape(fresh.shape[0], -1).any(1)
    shared_depth[arrived] = fresh[arrived]
    return shared_depth


@torch.no_grad()
def occlusion_mask_view(rays_o, rays_d, depth_half, bbox_center, box_size, H, W, kernel_size=91):
    """tile.py:391-400 for one view -> bool [H,W,1], True = the pixel takes part in training."""
    depth = depth_half.repeat_interleave(2, 0).repeat_interleave(2, 1)[:H, :W].reshape(-1, 1)
    bounds = torch.full((rays_o.shape[0], 2), -1.0, device=rays_o.device)
    ray_aabb_intersection(rays_o.contiguous(), rays_d.contiguous(), bbox_center, box_size, bounds)
    occ = ((depth > bounds[..., :1]) & (bounds[..., :1] != -1)).reshape(1, 1, H, W)
    kernel = torch.ones((1, 1, kernel_size, kernel_size), dtype=torch.float32, device=rays_o.device)
    occ = 1.0 - F.conv2d(1.0 - occ.float(), kernel, padding=(kernel_size // 2, kernel_size // 2)).clamp(0, 1)
    return occ.bool().reshape(H, W, 1)


@torch.no_grad()
def update_occlusion_mask(model, get_rays, H, W, visible_poses, shared_depth, kernel_size=91):
    """tile.py:366-409 -> occlusions bool [num_camera, H, W, 1] (True where no shared depth exists or the camera is
    inside this tile)."""
    center, size = model._center_dev, model._half_dev
    occlusions = torch.ones((len(visible_poses), H, W, 1), dtype=torch.bool, device=model.device)
    for idx, ori_idx in enumerate(visible_poses):
        depth = shared_depth[ori_idx]
        if bool(torch.isinf(depth).all()):  # shared_depth[ori_idx] == None in the reference
            continue
        rays_o, rays_d = get_rays(idx)
        rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
        if camera_inside(rays_o[0], center, size):
            continue
        occlusions[idx] = occlusion_mask_view(rays_o, rays_d, depth, center, size, H, W, kernel_size)
    return occlusions
