"""Drop-in for the reference's pybind module CUDA_EXT (cuda/binding.cpp:10-54), hot-path ops only.

Same names, positional argument order and in-place output convention as the reference's
C++ prototypes (cuda/include/{compute_ray,sample,helper,adam}.h); every function returns
None.  Each call forwards to one C-ABI entry point of libscanerf_hip.so on the current
torch stream.  Ops of the module that are off the hot path (warp-loss, view selection,
BlockBuilder: SURVEY.md section 2.2) are not provided and raise on access.
"""
import ctypes

import torch

from ..._capi import check, dev_ptr, lib, stream

_f32, _i32 = torch.float32, torch.int32


def compute_ray_forward(rays_o, rays_d, Ks, C2Ws, locs):
    """compute_ray.h:9-14"""
    B = rays_o.shape[0]
    check(lib().scanerf_compute_ray_forward(dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(rays_d, _f32, "rays_d"),
                                            dev_ptr(Ks, _f32, "Ks"), dev_ptr(C2Ws, _f32, "C2Ws"),
                                            dev_ptr(locs, _i32, "locs"), ctypes.c_int(B), stream()),
          "compute_ray_forward")


def compute_ray_backward(grad_rays_o, grad_rays_d, Ks, grad_C2Ws, locs):
    """compute_ray.h:16-21.  Per-ray adjoint (the reference's kernel indexes the incoming
    gradients by view, compute_ray_kernel.cu:71-72)."""
    B = grad_rays_o.shape[0]
    check(lib().scanerf_compute_ray_backward(dev_ptr(grad_rays_o, _f32, "grad_rays_o"),
                                             dev_ptr(grad_rays_d, _f32, "grad_rays_d"), dev_ptr(Ks, _f32, "Ks"),
                                             dev_ptr(grad_C2Ws, _f32, "grad_C2Ws"), dev_ptr(locs, _i32, "locs"),
                                             ctypes.c_int(B), ctypes.c_int(grad_C2Ws.shape[0]), stream()),
          "compute_ray_backward")


def ray_aabb_intersection(rays_o, rays_d, aabb_center, aabb_size, bounds):
    """helper.h:10-15: bounds [B,2] pre-filled by the caller (-1)."""
    check(lib().scanerf_ray_aabb_intersection(dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(rays_d, _f32, "rays_d"),
                                              dev_ptr(aabb_center, _f32, "aabb_center"),
                                              dev_ptr(aabb_size, _f32, "aabb_size"), dev_ptr(bounds, _f32, "bounds"),
                                              ctypes.c_int(rays_o.shape[0]), ctypes.c_int(1), stream()),
          "ray_aabb_intersection")


def ray_aabb_intersection_v2(rays_o, rays_d, aabb_center, aabb_size, bounds):
    """helper.h:17-22: K boxes, bounds [B,K,2]."""
    check(lib().scanerf_ray_aabb_intersection(dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(rays_d, _f32, "rays_d"),
                                              dev_ptr(aabb_center, _f32, "aabb_center"),
                                              dev_ptr(aabb_size, _f32, "aabb_size"), dev_ptr(bounds, _f32, "bounds"),
                                              ctypes.c_int(rays_o.shape[0]), ctypes.c_int(aabb_center.shape[0]),
                                              stream()),
          "ray_aabb_intersection_v2")


def sample_points_grid(rays_o, rays_d, z_vals, dists, block_corner, block_size, occupied_gird, log2dim):
    """helper.h:42-50: the live training sampler; z_vals/dists [B,S] pre-filled with -1."""
    if log2dim.dtype != _i32:
        raise RuntimeError(f"scanerf: log2dim must be int32 (the reference reinterprets it as int*), got {log2dim.dtype}")
    check(lib().scanerf_sample_points_grid(dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(rays_d, _f32, "rays_d"),
                                           dev_ptr(z_vals, _f32, "z_vals"), dev_ptr(dists, _f32, "dists"),
                                           dev_ptr(block_corner, _f32, "block_corner"),
                                           dev_ptr(block_size, _f32, "block_size"),
                                           dev_ptr(occupied_gird, (torch.bool, torch.uint8), "occupied_gird"),
                                           dev_ptr(log2dim, _i32, "log2dim"), ctypes.c_int(rays_o.shape[0]),
                                           ctypes.c_int(z_vals.shape[1]), stream()),
          "sample_points_grid")


def sample_points_contract(rays_o, rays_d, z_vals, block_corner, block_size, occupied_gird):
    """helper.h:33-39.  Ill-formed in the reference (declared to return a Tensor, returns
    nothing: helper_kernel.cu:511-536) and has no caller; the surface exists, the op does not."""
    raise NotImplementedError("sample_points_contract is dead code in the reference (no caller, UB return); "
                              "use sample_points_grid")


def sample_insideout_block(rays_o, rays_d, num_sample, num_sample_bg, block_center, block_size, far, z_vals,
                           z_vals_bg):
    """sample.h:9-17"""
    check(lib().scanerf_sample_insideout_block(dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(rays_d, _f32, "rays_d"),
                                               ctypes.c_int(num_sample), ctypes.c_int(num_sample_bg),
                                               dev_ptr(block_center, _f32, "block_center"),
                                               dev_ptr(block_size, _f32, "block_size"), ctypes.c_float(far),
                                               dev_ptr(z_vals, _f32, "z_vals"), dev_ptr(z_vals_bg, _f32, "z_vals_bg"),
                                               ctypes.c_void_p(0), ctypes.c_int(rays_o.shape[0]), stream()),
          "sample_insideout_block")


def background_sampling_cuda(rays_o, rays_d, starts, bg_depth, z_vals, num_sample, sample_range):
    """sample.h:19-26 (rays are unused by the reference kernel too)."""
    check(lib().scanerf_background_sampling(dev_ptr(starts, _f32, "starts"), dev_ptr(bg_depth, _f32, "bg_depth"),
                                            dev_ptr(z_vals, _f32, "z_vals"), ctypes.c_int(num_sample),
                                            ctypes.c_float(sample_range), ctypes.c_int(rays_o.shape[0]), stream()),
          "background_sampling_cuda")


def _adam(fn, name, params, grad_params, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, mdt):
    check(fn(dev_ptr(params, _f32, "params"), dev_ptr(grad_params, _f32, "grad_params"),
             dev_ptr(exp_avg, mdt, "exp_avg"), dev_ptr(exp_avg_sq, mdt, "exp_avg_sq"), ctypes.c_float(lr),
             ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_float(eps), ctypes.c_int(step),
             ctypes.c_int64(params.shape[0]), ctypes.c_int(params.shape[1]), stream()), name)


def adam_step_cuda(params, grad_params, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
    """adam.h:10-16.  `step` is the previous step count: the reference's `int &step` increment is
    invisible to Python, its kernel runs with step+1 (adam_kernel.cu:83)."""
    _adam(lib().scanerf_adam_step, "adam_step_cuda", params, grad_params, exp_avg, exp_avg_sq, lr, beta1, beta2, eps,
          step, _f32)


def adam_step_cuda_fp16(params, grad_params, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
    """adam.h:18-24: moments stored as half, loss scale 128."""
    _adam(lib().scanerf_adam_step_fp16, "adam_step_cuda_fp16", params, grad_params, exp_avg, exp_avg_sq, lr, beta1,
          beta2, eps, step, torch.float16)


def voxelize_mesh(_log2dim, block_corner, block_size, model_path, vis, init_out, outside):
    """cuda/include/voxelize.h:12-119: initialise the sampler's occupancy grid `vis` (and `outside`) from a PLY mesh;
    model_path == "" marks every cell occupied.  The reference runs this on the host over CPU tensors
    (hashgrid/__init__.py:71-80); here the faces are marked by a HIP kernel.  `vis` / `outside` are bool grids
    [2^lx,2^ly,2^lz], updated in place: GPU tensors directly, CPU tensors (what the reference's caller passes) through
    a device copy.  _log2dim [3] int32, block_corner / block_size [3] float32 (any device)."""
    from ... import formats
    if vis.dtype != torch.bool or outside.dtype != torch.bool:
        raise RuntimeError("scanerf: voxelize_mesh needs bool grids")
    l2d = [int(v) for v in _log2dim.detach().cpu().reshape(-1).tolist()]
    if tuple(vis.shape) != tuple(1 << k for k in l2d) or tuple(outside.shape) != tuple(vis.shape):
        raise RuntimeError(f"scanerf: voxelize_mesh grids must be {tuple(1 << k for k in l2d)}, got {tuple(vis.shape)}")
    if model_path == "":
        vis.fill_(True)
        return
    dev = vis.device if vis.is_cuda else torch.device("cuda", torch.cuda.current_device())
    verts, faces = formats.read_ply(model_path)
    v = torch.from_numpy(verts).to(dev).contiguous()
    f = torch.from_numpy(faces).to(dev).contiguous()
    gv = vis if vis.is_cuda else vis.to(dev)
    go = outside if outside.is_cuda else outside.to(dev)
    if not (gv.is_contiguous() and go.is_contiguous()):
        raise RuntimeError("scanerf: voxelize_mesh grids must be contiguous")
    scratch = torch.empty(6, dtype=torch.int32, device=dev)
    corner = (ctypes.c_float * 3)(*[float(x) for x in block_corner.detach().cpu().reshape(-1).tolist()])
    size = (ctypes.c_float * 3)(*[float(x) for x in block_size.detach().cpu().reshape(-1).tolist()])
    check(lib().scanerf_voxelize_mesh(dev_ptr(v, _f32, "vertices"), dev_ptr(f, _i32, "faces"), ctypes.c_int(v.shape[0]),
                                      ctypes.c_int(f.shape[0]), (ctypes.c_int32 * 3)(*l2d), corner, size,
                                      dev_ptr(gv, torch.bool, "vis"), ctypes.c_int(int(bool(init_out))),
                                      dev_ptr(go, torch.bool, "outside"), dev_ptr(scratch, _i32, "scratch"), stream()),
          "voxelize_mesh")
    if gv is not vis:
        vis.copy_(gv)
        outside.copy_(go)


_OFF_PATH = ("proj2pixel_and_fetch_color", "computeViewcost", "grid_sample_forward_cuda",
             "grid_sample_backward_cuda", "gaussian_grid_sample_forward_cuda", "gaussian_grid_sample_backward_cuda",
             "grid_sample_bool_cuda", "proj2neighbor_forward", "proj2neighbor_backward", "BlockBuilder")


def __getattr__(name):
    if name in _OFF_PATH:
        raise AttributeError(f"CUDA_EXT.{name} is outside the per-tile rendering hot path and is not part of this "
                             "build (SURVEY.md section 2.2: loss-side / preprocessing op)")
    raise AttributeError(name)
