"""Drop-in for the reference's pybind module HASHGRID (hashgrid/binding.cpp:9-44), training-path ops.

Same names / positional order / in-place outputs as hashgrid/include/hashgrid.h:19-51 (encoder ops) and
hashgrid/include/rendering.h:20-182 (the render-time ops of rendering.py's novel-view loop, second half of this file).
"""
import ctypes

import torch

from ..._capi import check, dev_ptr, feat_dtype_code, lib, stream, workspace

_f32, _i32 = torch.float32, torch.int32
_feat = (torch.float32, torch.float16, torch.bfloat16)


def _res(resolutions):
    if resolutions.dtype != _i32:
        raise RuntimeError(f"scanerf: resolutions must be int32 [L,3] (reinterpreted as int3*), got {resolutions.dtype}")
    return dev_ptr(resolutions, _i32, "resolutions")


def embedding_bg_forward_cuda(points, outputs, features, resolutions):
    """hashgrid.h:38-42: points [N,3] in [-2,2], outputs [N,L,2] (zero-filled by the caller),
    features [L,T,2].  fp32 tables as in the reference; f16/bf16 tables are also accepted."""
    N, (L, T) = points.shape[0], features.shape[:2]
    check(lib().scanerf_embedding_bg_forward(dev_ptr(points, _f32, "points"), dev_ptr(outputs, _f32, "outputs"),
                                             dev_ptr(features, _feat, "features"), _res(resolutions),
                                             ctypes.c_int(N), ctypes.c_int(L), ctypes.c_int(T),
                                             ctypes.c_int(feat_dtype_code(features)), stream()),
          "embedding_bg_forward_cuda")


TABLE_GRAD_ROUTE = "binned"   # or "atomics"
RECORD_FORMAT = -1
# Decoder arithmetic of the render-time inference ops: "t16" (default: 16-sample tiles, four waves per SIMD), "h3" (round 4's
# 32-sample-tile kernel) or "f32" (single-pass f32 MFMA: exact f32, [B,S] arrays only) -- flag bits OR-ed into the ops' `sample_major`
INFER_ARITH = "t16"
_INFER_FLAGS = {"t16": 0, "h3": 8, "f32": 16}   # include/scanerf_hip.h SCANERF_INFER_H3 / SCANERF_INFER_F32


def embedding_bg_backward_cuda(points, grad_in, grad_points, grad_features, features, resolutions):
    """hashgrid.h:45-51: accumulates into grad_points [N,3] and grad_features [L,T,2].
    The table gradient goes through the atomic-free binned scatter (csrc/scatter.hip) when the
    shape allows; TABLE_GRAD_ROUTE = "atomics" (module attribute) forces the reference-style atomic kernel; RECORD_FORMAT
    (-1 = default: 12-byte records for 16-level point-major rows, 0 = 16-byte, 2 = 12-byte) is the op's `compact_records`."""
    N, (L, T) = points.shape[0], features.shape[:2]
    need = 0
    if grad_features is not None and N >= 4096 and TABLE_GRAD_ROUTE != "atomics":
        need = lib().scanerf_embedding_bwd_workspace_bytes(ctypes.c_int(N), ctypes.c_int(L), ctypes.c_int(T))
    if need:
        ws = workspace(points.device, need)
        check(lib().scanerf_embedding_bg_backward_binned(
            dev_ptr(points, _f32, "points"), dev_ptr(grad_in, _f32, "grad_in"),
            dev_ptr(grad_features, _f32, "grad_features"), _res(resolutions), ctypes.c_int(N), ctypes.c_int(L),
            ctypes.c_int(T), ctypes.c_int(0), ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), ctypes.c_int(RECORD_FORMAT),
            stream()), "embedding_bg_backward_cuda(binned)")
        if grad_points is None:
            return
        grad_features = None  # the point gradient still needs the corner features: gather kernel
    check(lib().scanerf_embedding_bg_backward(dev_ptr(points, _f32, "points"), dev_ptr(grad_in, _f32, "grad_in"),
                                              dev_ptr(grad_points, _f32, "grad_points", allow_none=True),
                                              dev_ptr(grad_features, _f32, "grad_features", allow_none=True),
                                              dev_ptr(features, _f32, "features"), _res(resolutions),
                                              ctypes.c_int(N), ctypes.c_int(L), ctypes.c_int(T), stream()),
          "embedding_bg_backward_cuda")


def embedding_forward_cuda(points, outputs, features, block_corner, block_size, resolutions):
    """hashgrid.h:19-25 (world-space box variant)."""
    N, (L, T) = points.shape[0], features.shape[:2]
    check(lib().scanerf_embedding_forward(dev_ptr(points, _f32, "points"), dev_ptr(outputs, _f32, "outputs"),
                                          dev_ptr(features, _f32, "features"),
                                          dev_ptr(block_corner, _f32, "block_corner"),
                                          dev_ptr(block_size, _f32, "block_size"), _res(resolutions),
                                          ctypes.c_int(N), ctypes.c_int(L), ctypes.c_int(T), stream()),
          "embedding_forward_cuda")


def embedding_backward_cuda(points, grad_in, grad_points, grad_features, features, block_corner, block_size,
                            resolutions):
    """hashgrid.h:27-35"""
    N, (L, T) = points.shape[0], features.shape[:2]
    check(lib().scanerf_embedding_backward(dev_ptr(points, _f32, "points"), dev_ptr(grad_in, _f32, "grad_in"),
                                           dev_ptr(grad_points, _f32, "grad_points", allow_none=True),
                                           dev_ptr(grad_features, _f32, "grad_features"),
                                           dev_ptr(features, _f32, "features"),
                                           dev_ptr(block_corner, _f32, "block_corner"),
                                           dev_ptr(block_size, _f32, "block_size"), _res(resolutions),
                                           ctypes.c_int(N), ctypes.c_int(L), ctypes.c_int(T), stream()),
          "embedding_backward_cuda")


# ------------------------------------------------------------------ render-time ops (rendering.h:20-182)
_i16, _i64, _bool = torch.int16, torch.int64, (torch.bool, torch.uint8)
_I = ctypes.c_int


# The 16-sample-tile inference kernel takes images whose three Gaussian-activated layers (blob floats [0, 2112): Spatial_MLP.mlp.0,
# [6503, 9639): Directional_MLP.mlp.0, [9639, 13799): Directional_MLP.mlp.2 -- bias + weights each) carry the activation's constant
# sqrt(50 log2 e), so that exp(-u^2 / 0.02) = exp2(-(u')^2) (include/scanerf_hip.h SCANERF_INFER_FOLDED): one vector instruction
# less per activation.  The other kernels (INFER_ARITH "h3" / "f32") read the plain images.
FOLD_ACTIVATION = True
_FOLD_C = 8.493218  # sqrt(72.13475204444817)
_FOLD_RANGES = ((0, 2112), (6503, 9639), (9639, 13799))
_INFER_FOLDED = 32


def _pack_images(params, images, folded=False):
    p = params.contiguous()
    if folded:
        p = p.clone()
        for lo, hi in _FOLD_RANGES:
            p[:, lo:hi] *= _FOLD_C
    ones = torch.ones(32, dtype=_f32, device=p.device)
    for b in range(p.shape[0]):
        check(lib().scanerf_pack_decoder(ctypes.c_void_p(p[b].data_ptr()), dev_ptr(ones, _f32, "wf"),
                                         ctypes.c_void_p(images[b].data_ptr()), stream()), "pack_decoder")
    return images


def _new_images(params):
    if params.dim() != 2 or params.shape[1] != 13994:
        raise RuntimeError(f"scanerf: decoder blobs must be [nb, 13994], got {tuple(params.shape)}")
    return torch.empty((params.shape[0], lib().scanerf_render_workspace_floats()), dtype=_f32, device=params.device)


class PackedDecoders:
    """[nb, 13994] render-time decoder blobs packed once into the LDS images of the MFMA decoder (weight_feature == 1:
    the render-time decoder has no coarse-to-fine mask, decoder.h:169-218).  Owned by whoever owns the blobs
    (renderer.TileSetRenderer packs once in its constructor) and accepted by pts_inference / bg_pts_inference_v2 in
    place of the raw `params` tensor; call repack() after changing the blobs."""

    def __init__(self, params):
        self.params = params
        self.images = _new_images(params)
        self.images_folded = _new_images(params)   # (for the 16-sample-tile kernel: FOLD_ACTIVATION)
        self.repack()

    def repack(self):
        _pack_images(self.params, self.images)
        _pack_images(self.params, self.images_folded, folded=True)
        return self


# Raw `params` tensors (the reference's calling convention) are packed on first use and remembered PER TENSOR OBJECT: the
# entry holds only a WEAK reference to the tensor, is dropped when the tensor dies, and is trusted only while that
# reference still points at the very same object and its version counter is unchanged -- an address recycled by the
# caching allocator for a new tensor can never hit an old entry.  Writes that bypass the version counter
# (params.data[...] = ...) are not seen: own a PackedDecoders and repack() instead.
_images = {}


def _infer_folded(nb, z_vals):
    """The call will run the 16-sample-tile kernel (not its single-pass fallback for > 64 tiles / >= 2^31 samples)."""
    return FOLD_ACTIVATION and INFER_ARITH == "t16" and nb <= 64 and z_vals.numel() < 2 ** 31


def _infer_flags(folded):
    return _INFER_FLAGS[INFER_ARITH] | (_INFER_FOLDED if folded else 0)


def _packed_images(params, folded=False):
    """The packed images a call reads: folded for the 16-sample-tile kernel, plain otherwise."""
    if isinstance(params, PackedDecoders):
        return params.images_folded if folded else params.images
    import weakref
    key = (id(params), folded)
    hit = _images.get(key)
    if hit is not None and hit[0]() is params and hit[1] == params._version:
        return hit[2]
    images = _pack_images(params, _new_images(params), folded)
    _images[key] = (weakref.ref(params, lambda _r, key=key: _images.pop(key, None)), params._version, images)
    return images


def ray_block_intersection(rays_o, rays_d, block_corners, block_sizes, intersections):
    """rendering.h: intersections [B,nb,2] pre-filled with 1e7 (rendering.py:299)."""
    check(lib().scanerf_ray_block_intersection(dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(rays_d, _f32, "rays_d"),
                                               dev_ptr(block_corners, _f32, "block_corners"),
                                               dev_ptr(block_sizes, _f32, "block_sizes"),
                                               dev_ptr(intersections, _f32, "intersections"), _I(rays_d.shape[0]),
                                               _I(block_corners.shape[0]), stream()), "ray_block_intersection")


def _S(z_vals, sample_major):
    """per-sample arrays: [B,S] (0, the reference's), [S,B] (1) or [B/32,S,32] (2) -- scanerf_hip.h `sample_major`"""
    return _I(z_vals.shape[0] if (int(sample_major) & 3) == 1 else z_vals.shape[1])


def sort_tracing_blocks(intersections):
    """torch.argsort(intersections[..., 0], dim=-1, stable=True).int() (rendering.py:301) in one launch: [B,nb] i32."""
    B, nb = intersections.shape[:2]
    if nb > 64:
        return torch.argsort(intersections[..., 0], dim=-1, stable=True).int().contiguous()
    order = torch.empty((B, nb), dtype=torch.int32, device=intersections.device)
    check(lib().scanerf_sort_tracing_blocks(dev_ptr(intersections, _f32, "intersections"), dev_ptr(order, _i32, "order"), _I(B), _I(nb),
                                            stream()), "sort_tracing_blocks")
    return order


def sample_points(rays_o, rays_d, block_corners, block_sizes, grid_occupied, grid_starts, grid_log2dim, tracing_blocks,
                  intersections, tracing_idx, z_start, z_vals, dists, sample_major=False):
    check(lib().scanerf_render_sample_points(
        dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(rays_d, _f32, "rays_d"), dev_ptr(block_corners, _f32, "block_corners"),
        dev_ptr(block_sizes, _f32, "block_sizes"), dev_ptr(grid_occupied, _bool, "grid_occupied"),
        dev_ptr(grid_starts, _i64, "grid_starts"), dev_ptr(grid_log2dim, _i32, "grid_log2dim"),
        dev_ptr(tracing_blocks, _i32, "tracing_blocks"), dev_ptr(intersections, _f32, "intersections"),
        dev_ptr(tracing_idx, _i32, "tracing_idx"), dev_ptr(z_start, _f32, "z_start"), dev_ptr(z_vals, _f32, "z_vals"),
        dev_ptr(dists, _f32, "dists"), _I(rays_d.shape[0]), _S(z_vals, sample_major), _I(block_corners.shape[0]),
        _I(int(sample_major)), stream()), "sample_points")


def prepare_points(z_vals, runing_mask, intersections, block_idxs, sample_major=False):
    check(lib().scanerf_prepare_points(dev_ptr(z_vals, _f32, "z_vals"), dev_ptr(runing_mask, _bool, "runing_mask"),
                                       dev_ptr(intersections, _f32, "intersections"),
                                       dev_ptr(block_idxs, _i16, "block_idxs"), _I(intersections.shape[0]), _S(z_vals, sample_major),
                                       _I(intersections.shape[1]), _I(int(sample_major)), stream()), "prepare_points")


def pts_inference(rays_o, rays_d, z_vals, dists, block_idxs, features_tables, params, resolution, grid_occupied,
                  grid_starts, grid_log2dim, block_corners, block_sizes, diffuse, specular, alpha, sample_major=False):
    folded = _infer_folded(block_corners.shape[0], z_vals)
    img = _packed_images(params, folded)
    check(lib().scanerf_pts_inference(
        dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(rays_d, _f32, "rays_d"), dev_ptr(z_vals, _f32, "z_vals"),
        dev_ptr(dists, _f32, "dists"), dev_ptr(block_idxs, _i16, "block_idxs"),
        dev_ptr(features_tables, torch.float16, "features_tables"), dev_ptr(img, _f32, "images"),
        dev_ptr(resolution, _i32, "resolution"), dev_ptr(grid_occupied, _bool, "grid_occupied"),
        dev_ptr(grid_starts, _i64, "grid_starts"), dev_ptr(grid_log2dim, _i32, "grid_log2dim"),
        dev_ptr(block_corners, _f32, "block_corners"), dev_ptr(block_sizes, _f32, "block_sizes"),
        dev_ptr(diffuse, _f32, "diffuse"), dev_ptr(specular, _f32, "specular"), dev_ptr(alpha, _f32, "alpha"),
        _I(rays_d.shape[0]), _S(z_vals, sample_major), _I(features_tables.shape[2]), _I(block_corners.shape[0]),
        _I(int(sample_major) | _infer_flags(folded)), stream()), "pts_inference")


SKIP_UNSAMPLED = 4   # include/scanerf_hip.h SCANERF_SKIP_UNSAMPLED: OR into `sample_major` of pts_inference_tracing / accumulate_color


def tracing_fusable(nb):
    """pts_inference_tracing serves up to 8 tiles on the 16-sample-tile kernel (include/scanerf_hip.h)"""
    return nb <= 8 and INFER_ARITH == "t16"


def pts_inference_tracing(rays_o, rays_d, z_vals, dists, running_mask, intersections, features_tables, params, resolution,
                          grid_occupied, grid_starts, grid_log2dim, block_corners, block_sizes, diffuse, specular, alpha,
                          sample_major=False):
    """prepare_points + pts_inference in one launch (no reference counterpart): same outputs as
    `prepare_points(z_vals, running_mask, intersections, block_idxs); pts_inference(..., block_idxs, ...)` without the
    block_idxs array."""
    folded = _infer_folded(block_corners.shape[0], z_vals)
    img = _packed_images(params, folded)
    check(lib().scanerf_pts_inference_tracing(
        dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(rays_d, _f32, "rays_d"), dev_ptr(z_vals, _f32, "z_vals"),
        dev_ptr(dists, _f32, "dists"), dev_ptr(running_mask, _bool, "running_mask"), dev_ptr(intersections, _f32, "intersections"),
        dev_ptr(features_tables, torch.float16, "features_tables"), dev_ptr(img, _f32, "images"),
        dev_ptr(resolution, _i32, "resolution"), dev_ptr(grid_occupied, _bool, "grid_occupied"),
        dev_ptr(grid_starts, _i64, "grid_starts"), dev_ptr(grid_log2dim, _i32, "grid_log2dim"),
        dev_ptr(block_corners, _f32, "block_corners"), dev_ptr(block_sizes, _f32, "block_sizes"),
        dev_ptr(diffuse, _f32, "diffuse"), dev_ptr(specular, _f32, "specular"), dev_ptr(alpha, _f32, "alpha"),
        _I(rays_d.shape[0]), _S(z_vals, sample_major), _I(features_tables.shape[2]), _I(block_corners.shape[0]),
        _I(int(sample_major) | _infer_flags(folded)), stream()), "pts_inference_tracing")


def accumulate_color(pts_diffuse, pts_specular, pts_alpha, transparency, z_vals, diffuse, specular, depth, sample_major=False):
    check(lib().scanerf_accumulate_color(
        dev_ptr(pts_diffuse, _f32, "pts_diffuse"), dev_ptr(pts_specular, _f32, "pts_specular"),
        dev_ptr(pts_alpha, _f32, "pts_alpha"), dev_ptr(transparency, _f32, "transparency"), dev_ptr(z_vals, _f32, "z_vals"),
        dev_ptr(diffuse, _f32, "diffuse"), dev_ptr(specular, _f32, "specular"), dev_ptr(depth, _f32, "depth"),
        _I(transparency.shape[0]), _S(z_vals, sample_major), _I(int(sample_major)), stream()), "accumulate_color")


def inverse_z_sampling(intersections, related_bidx, z_vals, sample_range, sample_major=False):
    check(lib().scanerf_render_inverse_z_sampling(dev_ptr(intersections, _f32, "intersections"),
                                                  dev_ptr(related_bidx, _i16, "related_bidx"),
                                                  dev_ptr(z_vals, _f32, "z_vals"), ctypes.c_float(sample_range),
                                                  _I(intersections.shape[0]), _S(z_vals, sample_major),
                                                  _I(intersections.shape[1]), _I(int(sample_major)), stream()), "inverse_z_sampling")


def bg_pts_inference_v2(rays_o, rays_d, z_vals, bg_idxs, step, block_corners, block_sizes, resolution, features_tables,
                        params, diffuse, specular, alpha, sample_major=False):
    folded = _infer_folded(block_corners.shape[0], z_vals)
    img = _packed_images(params, folded)
    check(lib().scanerf_bg_pts_inference_v2(
        dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(rays_d, _f32, "rays_d"), dev_ptr(z_vals, _f32, "z_vals"),
        dev_ptr(bg_idxs, _i16, "bg_idxs"), _I(step), dev_ptr(block_corners, _f32, "block_corners"),
        dev_ptr(block_sizes, _f32, "block_sizes"), dev_ptr(resolution, _i32, "resolution"),
        dev_ptr(features_tables, torch.float16, "features_tables"), dev_ptr(img, _f32, "images"),
        dev_ptr(diffuse, _f32, "diffuse"), dev_ptr(specular, _f32, "specular"), dev_ptr(alpha, _f32, "alpha"),
        _I(rays_d.shape[0]), _S(z_vals, sample_major), _I(features_tables.shape[2]), _I(block_corners.shape[0]),
        _I(int(sample_major) | _infer_flags(folded)), stream()), "bg_pts_inference_v2")


def bg_pts_inference(rays_o, rays_d, z_vals, outgoing_bidxs, blend_weights, block_corners, block_sizes, resolution,
                     features_tables, params, diffuse, specular, alpha):
    """hashgrid/binding.cpp:31 (v1; rendering_kernel.cu:872-1008, :1176-1208; superseded by bg_pts_inference_v2 in
    rendering.py:497 and without a caller there): every sample blends the inference of ALL of its ray's outgoing tiles --
    slots of outgoing_bidxs [B,4] up to the first -1 -- on the same z_vals: out = sum_i w_i x_i / sum_i w_i with x_i what
    bg_pts_inference_v2(step = i) writes and w_i = blend_weights [B,4]; rays without an outgoing tile get zeros.
    One v2 launch per slot in use (the chunk-major HIP kernel), the blend in torch."""
    B, S = z_vals.shape[0], z_vals.shape[1]
    live = torch.cumprod((outgoing_bidxs != -1).to(torch.int32), dim=1).bool()   # the reference's loop breaks at the first -1
    idx = torch.where(live, outgoing_bidxs, torch.full_like(outgoing_bidxs, -1)).contiguous()
    acc_d, acc_s, acc_a = torch.zeros_like(diffuse), torch.zeros_like(specular), torch.zeros_like(alpha)
    wsum = torch.zeros((B, 1, 1), dtype=torch.float32, device=z_vals.device)
    for i in range(idx.shape[1]):
        if not bool(live[:, i].any()):
            break
        td, ts, ta = torch.zeros_like(diffuse), torch.zeros_like(specular), torch.zeros_like(alpha)
        bg_pts_inference_v2(rays_o, rays_d, z_vals, idx, i, block_corners, block_sizes, resolution, features_tables, params, td, ts, ta)
        w = (blend_weights[:, i] * live[:, i]).reshape(B, 1, 1)
        acc_d += w * td.reshape(B, S, 3)
        acc_s += w * ts.reshape(B, S, 3)
        acc_a += w * ta.reshape(B, S, 1)
        wsum += w
    norm = torch.where(wsum > 0, wsum, torch.ones_like(wsum))
    diffuse.copy_((acc_d.reshape(B, S, 3) / norm).reshape(diffuse.shape))
    specular.copy_((acc_s.reshape(B, S, 3) / norm).reshape(specular.shape))
    alpha.copy_((acc_a.reshape(B, S, 1) / norm).reshape(alpha.shape))


def update_outgoing_bidx(rays_o, rays_d, block_corners, block_sizes, tracing_blocks, intersections, outgoing_bidxs,
                         blend_weights, ratio, skip):
    check(lib().scanerf_update_outgoing_bidx(
        dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(rays_d, _f32, "rays_d"), dev_ptr(block_corners, _f32, "block_corners"),
        dev_ptr(block_sizes, _f32, "block_sizes"), dev_ptr(tracing_blocks, _i32, "tracing_blocks"),
        dev_ptr(intersections, _f32, "intersections"), dev_ptr(outgoing_bidxs, _i16, "outgoing_bidxs"),
        dev_ptr(blend_weights, _f32, "blend_weights"), ctypes.c_float(ratio), _I(int(bool(skip))),
        _I(tracing_blocks.shape[0]), _I(tracing_blocks.shape[1]), stream()), "update_outgoing_bidx")


def update_outgoing_bidx_v2(rays_o, rays_d, block_corners, block_sizes, tracing_blocks, intersections, inside_bidxs,
                            blend_weights):
    check(lib().scanerf_update_outgoing_bidx_v2(
        dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(block_corners, _f32, "block_corners"),
        dev_ptr(block_sizes, _f32, "block_sizes"), dev_ptr(inside_bidxs, _i16, "inside_bidxs"),
        dev_ptr(blend_weights, _f32, "blend_weights"), _I(tracing_blocks.shape[0]), _I(tracing_blocks.shape[1]), stream()),
        "update_outgoing_bidx_v2")


def get_last_block(tracing_blocks, bidxs, intersections):
    check(lib().scanerf_get_last_block(dev_ptr(tracing_blocks, _i32, "tracing_blocks"), dev_ptr(bidxs, _i32, "bidxs"),
                                       dev_ptr(intersections, _f32, "intersections"), _I(intersections.shape[0]),
                                       _I(intersections.shape[1]), stream()), "get_last_block")


def ray_firsthit_block(rays_o, rays_d, block_corners, block_sizes, grid_occupied, grid_starts, grid_log2dim,
                       tracing_blocks, intersections, hit_blockIdxs):
    check(lib().scanerf_ray_firsthit_block(
        dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(rays_d, _f32, "rays_d"), dev_ptr(block_corners, _f32, "block_corners"),
        dev_ptr(block_sizes, _f32, "block_sizes"), dev_ptr(grid_occupied, _bool, "grid_occupied"),
        dev_ptr(grid_starts, _i64, "grid_starts"), dev_ptr(grid_log2dim, _i32, "grid_log2dim"),
        dev_ptr(tracing_blocks, _i32, "tracing_blocks"), dev_ptr(intersections, _f32, "intersections"),
        dev_ptr(hit_blockIdxs, _i16, "hit_blockIdxs"), _I(rays_d.shape[0]), _I(block_corners.shape[0]), stream()),
        "ray_firsthit_block")


def process_occupied_grid(bidx, total_grid, block_corners, block_sizes, grid_occupied, grid_starts, grid_log2dim,
                          tgt_grid_occupied):
    check(lib().scanerf_process_occupied_grid(
        _I(bidx), _I(total_grid), dev_ptr(block_corners, _f32, "block_corners"), dev_ptr(block_sizes, _f32, "block_sizes"),
        dev_ptr(grid_occupied, _bool, "grid_occupied"), dev_ptr(grid_starts, _i64, "grid_starts"),
        dev_ptr(grid_log2dim, _i32, "grid_log2dim"), dev_ptr(tgt_grid_occupied, _bool, "tgt_grid_occupied"),
        _I(block_corners.shape[0]), stream()), "process_occupied_grid")


def sort_by_key(keys_tensor, values_tensor, starts_tensor):
    """rendering_kernel.cu:452-463 (thrust sort_by_key + unique_by_key; no caller in the reference).
    In-place on keys / values; unique keys and the first `starts` of each run are compacted to the
    front; returns the number of unique keys.  Host-side torch ops: not on the hot path."""
    k, order = torch.sort(keys_tensor, stable=True)
    keys_tensor.copy_(k)
    values_tensor.copy_(values_tensor[order])
    first = torch.ones_like(k, dtype=torch.bool)
    first[1:] = k[1:] != k[:-1]
    n = int(first.sum())
    keys_tensor[:n] = k[first]
    starts_tensor[:n] = starts_tensor[first]
    return n


def rendering_cuda(*args, **kwargs):
    """hashgrid/binding.cpp:22.  The reference's device body is commented out
    (hashgrid/src/rendering/renderbase_kernel.cu:72-177): dead code, exported as a stub."""
    raise NotImplementedError("rendering_cuda is dead code in the reference (device body commented out)")


class Sampler:
    """hashgrid/binding.cpp:39-43.  Constructed by HashGrid (hashgrid/__init__.py:68) but never
    built or used (its build call is commented out at :81-82): constructor only."""

    def build(self, *a, **k):
        raise NotImplementedError("Sampler.build is unused by the reference (hashgrid/__init__.py:81-82)")

    rebuild = samplePoints = build
