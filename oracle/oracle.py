"""CPU oracle for the ScaNeRF per-tile volume-rendering hot path.

THIS MODULE IS TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product package
never does (it fails loudly when its HIP library is missing instead).

Two halves:

* ``C`` -- ctypes view of ``oracle/scanerf_oracle.c`` (restatement of the
  reference's CUDA sources; parity unpinned, KAT-pinned only -- see that file).
* torch-CPU restatements of the reference's *Python* half of the path
  (``network.py``, ``hashgrid/__init__.py``, ``consensus.py``,
  ``admm_trainer.py``).  These ARE pinned: ``tests/golden/*.npz`` were captured
  by importing the reference (``tests/golden/make_golden.py``).

Citations are file:line under /root/reference.
"""
import ctypes
import math
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libscanerf_oracle.so")

PARAMSIZE = 13994  # hashgrid/include/decoder.h:20
TRAIN, INFERENCE = 0, 1  # cfg.py:1-2


def build(force=False):
    """Compile the C oracle (gcc).  Building the checker is not using it."""
    src = os.path.join(_HERE, "scanerf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def _load():
    if not os.path.exists(_SO):
        build()
    lib = ctypes.CDLL(_SO)
    lib.orc_hash.restype = ctypes.c_uint32
    lib.orc_hash.argtypes = [ctypes.c_int] * 4
    lib.orc_float2half.restype = ctypes.c_uint16
    lib.orc_float2half.argtypes = [ctypes.c_float]
    lib.orc_half2float.restype = ctypes.c_float
    lib.orc_half2float.argtypes = [ctypes.c_uint16]
    lib.orc_sample_insideout_block.restype = ctypes.c_int
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _np(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


_cf = ctypes.c_float
_ci = ctypes.c_int


# --------------------------------------------------------------------------- C half
def hash_index(x, y, z, T):
    return int(lib().orc_hash(int(x), int(y), int(z), int(T)))


def compute_ray_forward(locs, Ks, C2Ws):
    locs, Ks, C2Ws = _i32(_np(locs)), _f32(_np(Ks)), _f32(_np(C2Ws))
    B = locs.shape[0]
    o = np.zeros((B, 3), np.float32)
    d = np.zeros((B, 3), np.float32)
    lib().orc_compute_ray_forward(_p(locs), _p(Ks), _p(C2Ws), _p(o), _p(d), _ci(B))
    return o, d


def compute_ray_backward(g_o, g_d, Ks, locs, num_cam, ref_bug=False):
    g_o, g_d, Ks, locs = _f32(_np(g_o)), _f32(_np(g_d)), _f32(_np(Ks)), _i32(_np(locs))
    g = np.zeros((num_cam, 12), np.float32)
    lib().orc_compute_ray_backward(_p(g_o), _p(g_d), _p(Ks), _p(locs), _p(g), _ci(locs.shape[0]),
                                   _ci(1 if ref_bug else 0))
    return g


def ray_aabb_intersection(rays_o, rays_d, center, size):
    """center/size [3] -> bounds [B,2]; center/size [K,3] -> bounds [B,K,2] (v2)."""
    o, d = _f32(_np(rays_o)), _f32(_np(rays_d))
    c, s = _f32(_np(center)), _f32(_np(size))
    K = 1 if c.ndim == 1 else c.shape[0]
    B = o.shape[0]
    out = np.full((B, K, 2), -1, np.float32)
    lib().orc_ray_aabb_intersection(_p(o), _p(d), _p(c), _p(s), _p(out), _ci(B), _ci(K))
    return out[:, 0] if c.ndim == 1 else out


def sample_points_grid(rays_o, rays_d, corner, size, occ, log2dim, S):
    o, d = _f32(_np(rays_o)), _f32(_np(rays_d))
    occ8 = np.ascontiguousarray(_np(occ).astype(np.uint8))
    B = o.shape[0]
    z = np.full((B, S), -1, np.float32)
    dist = np.full((B, S), -1, np.float32)
    lib().orc_sample_points_grid(_p(o), _p(d), _p(z), _p(dist), _p(_f32(_np(corner))),
                                 _p(_f32(_np(size))), _p(occ8), _p(_i32(_np(log2dim))), _ci(B), _ci(S))
    return z, dist


def sample_insideout_block(rays_o, rays_d, S, S_bg, center, size, far):
    o, d = _f32(_np(rays_o)), _f32(_np(rays_d))
    B = o.shape[0]
    z = np.full((B, S), -1, np.float32)
    zb = np.full((B, S_bg), -1, np.float32)
    missed = lib().orc_sample_insideout_block(_p(o), _p(d), _ci(S), _ci(S_bg), _p(_f32(_np(center))),
                                              _p(_f32(_np(size))), _cf(far), _p(z), _p(zb), _ci(B))
    return z, zb, missed


def background_sampling(starts, bg_depth, S, sample_range):
    st, bd = _f32(_np(starts)), _f32(_np(bg_depth))
    B = st.shape[0]
    z = np.zeros((B, S), np.float32)
    lib().orc_background_sampling(_p(st), _p(bd), _p(z), _ci(S), _cf(sample_range), _ci(B))
    return z


def embedding_forward(points, features, res, corner=None, size=None):
    pts, feat, res = _f32(_np(points)), _f32(_np(features)), _i32(_np(res))
    N, (L, T) = pts.shape[0], feat.shape[:2]
    out = np.zeros((N, L, 2), np.float32)
    if corner is None:
        lib().orc_embedding_bg_forward(_p(pts), _p(out), _p(feat), _p(res), _ci(N), _ci(L), _ci(T))
    else:
        lib().orc_embedding_forward(_p(pts), _p(out), _p(feat), _p(_f32(_np(corner))), _p(_f32(_np(size))),
                                    _p(res), _ci(N), _ci(L), _ci(T))
    return out


def embedding_backward(points, grad_in, features, res, corner=None, size=None):
    pts, g, feat, res = _f32(_np(points)), _f32(_np(grad_in)), _f32(_np(features)), _i32(_np(res))
    N, (L, T) = pts.shape[0], feat.shape[:2]
    gp = np.zeros((N, 3), np.float32)
    gf = np.zeros_like(feat)
    if corner is None:
        lib().orc_embedding_bg_backward(_p(pts), _p(g), _p(gp), _p(gf), _p(feat), _p(res), _ci(N), _ci(L), _ci(T))
    else:
        lib().orc_embedding_backward(_p(pts), _p(g), _p(gp), _p(gf), _p(feat), _p(_f32(_np(corner))),
                                     _p(_f32(_np(size))), _p(res), _ci(N), _ci(L), _ci(T))
    return gp, gf


def adam_step(params, grad, m, v, lr, b1, b2, eps, step, fp16=False):
    """In place on numpy arrays shaped [K, 8]; `step` = previous step count (adam.h:16 quirk)."""
    K, dim = params.shape
    assert params.flags.c_contiguous and m.flags.c_contiguous and v.flags.c_contiguous
    g = _f32(grad)
    fn = lib().orc_adam_step_fp16 if fp16 else lib().orc_adam_step
    fn(_p(params), _p(g), _p(m), _p(v), _cf(lr), _cf(b1), _cf(b2), _cf(eps), _ci(step),
       ctypes.c_int64(K), _ci(dim))


def decoder_inference(params, feat, dirs):
    """decoder.h:169-218 over a flat 13994-float blob -> sigma, diffuse, specular(tinted), tint."""
    p, f, d = _f32(_np(params)), _f32(_np(feat)), _f32(_np(dirs))
    assert p.size == PARAMSIZE
    n = f.shape[0]
    sigma = np.zeros((n,), np.float32)
    diff = np.zeros((n, 3), np.float32)
    spec = np.zeros((n, 3), np.float32)
    tint = np.zeros((n, 3), np.float32)
    lib().orc_decoder_inference(_p(p), _p(f), _p(d), _p(sigma), _p(diff), _p(spec), _p(tint), _ci(n))
    return sigma, diff, spec, tint


# ---- a16: render-time kernels (hashgrid/src/rendering_kernel.cu) -------------------------------
def _u8(a):
    return np.ascontiguousarray(_np(a).astype(np.uint8))


def _i16(a):
    return np.ascontiguousarray(np.asarray(_np(a), dtype=np.int16))


def _i64(a):
    return np.ascontiguousarray(np.asarray(_np(a), dtype=np.int64))


def ray_block_intersection(rays_o, rays_d, corners, sizes):
    o, d, c, s = _f32(_np(rays_o)), _f32(_np(rays_d)), _f32(_np(corners)), _f32(_np(sizes))
    out = np.full((o.shape[0], c.shape[0], 2), 1e7, np.float32)
    lib().orc_ray_block_intersection(_p(o), _p(d), _p(c), _p(s), _p(out), _ci(o.shape[0]), _ci(c.shape[0]))
    return out


def render_sample_points(rays_o, rays_d, corners, sizes, occ, grid_starts, log2dim, S, tracing_blocks, inter,
                         tracing_idx, z_start):
    """One tracing step; tracing_idx / z_start are updated in place (numpy int32 / float32 arrays)."""
    o, d = _f32(_np(rays_o)), _f32(_np(rays_d))
    B, nb = o.shape[0], np.asarray(corners).shape[0]
    z = np.full((B, S), -1, np.float32)
    dd = np.full((B, S), -1, np.float32)
    lib().orc_render_sample_points(_p(o), _p(d), _p(_f32(_np(corners))), _p(_f32(_np(sizes))), _p(_u8(occ)),
                                   _p(_i64(grid_starts)), _p(_i32(_np(log2dim))), _ci(S), _ci(nb),
                                   _p(_i32(_np(tracing_blocks))), _p(_f32(_np(inter))), _p(tracing_idx), _p(z_start),
                                   _p(z), _p(dd), _ci(B))
    return z, dd


def prepare_points(z_vals, running, inter):
    z = _f32(_np(z_vals))
    B, S = z.shape
    nb = np.asarray(inter).shape[1]
    out = np.full((B, S, 4), -1, np.int16)
    lib().orc_prepare_points(_p(z), _p(_u8(running)), _p(out), _p(_f32(_np(inter))), _ci(S), _ci(nb), _ci(B))
    return out


def pts_inference(rays_o, rays_d, z_vals, dists, block_idxs, tables_f16, params, res, occ, grid_starts, log2dim,
                  corners, sizes):
    o, d, z, dd = _f32(_np(rays_o)), _f32(_np(rays_d)), _f32(_np(z_vals)), _f32(_np(dists))
    B, S = z.shape
    tb = np.ascontiguousarray(_np(tables_f16).view(np.uint16)) if _np(tables_f16).dtype == np.float16 else _np(tables_f16)
    T = tb.shape[2]
    dif, spec, al = np.zeros((B, S, 3), np.float32), np.zeros((B, S, 3), np.float32), np.zeros((B, S, 1), np.float32)
    lib().orc_pts_inference(_p(o), _p(d), _p(z), _p(dd), _p(_i16(block_idxs)), _p(tb), _p(_f32(_np(params))),
                            _p(_i32(_np(res))), _p(_u8(occ)), _p(_i64(grid_starts)), _p(_i32(_np(log2dim))), _ci(T),
                            _p(_f32(_np(corners))), _p(_f32(_np(sizes))), _p(dif), _p(spec), _p(al), _ci(B), _ci(S))
    return dif, spec, al


def accumulate_color(pts_dif, pts_spec, pts_alpha, transp, z_vals, dif, spec, depth):
    """In place on transp [B,1], dif/spec [B,3], depth [B,1] (numpy float32)."""
    z = _f32(_np(z_vals))
    lib().orc_accumulate_color(_p(_f32(pts_dif)), _p(_f32(pts_spec)), _p(_f32(pts_alpha)), _p(transp), _p(z), _p(dif),
                               _p(spec), _p(depth), _ci(z.shape[0]), _ci(z.shape[1]))


def render_inverse_z_sampling(inter, related, S, sample_range):
    it = _f32(_np(inter))
    B, nb = it.shape[:2]
    z = np.full((B, S), -1, np.float32)
    lib().orc_render_inverse_z_sampling(_p(it), _p(_i16(related)), _ci(S), _ci(nb), _cf(sample_range), _p(z), _ci(B))
    return z


def bg_pts_inference_v2(rays_o, rays_d, z_vals, bg_idxs, step, corners, sizes, res, tables_f16, params):
    o, d, z = _f32(_np(rays_o)), _f32(_np(rays_d)), _f32(_np(z_vals))
    B, S = z.shape
    tb = np.ascontiguousarray(_np(tables_f16).view(np.uint16))
    T = tb.shape[2]
    dif, spec, al = np.zeros((B, S, 3), np.float32), np.zeros((B, S, 3), np.float32), np.zeros((B, S, 1), np.float32)
    lib().orc_bg_pts_inference_v2(_p(o), _p(d), _p(z), _p(tb), _p(_f32(_np(params))), _p(_f32(_np(corners))),
                                  _p(_f32(_np(sizes))), _p(_i32(_np(res))), _p(_i16(bg_idxs)), _ci(step), _p(dif), _p(spec),
                                  _p(al), _ci(T), _ci(B), _ci(S))
    return dif, spec, al


def bg_pts_inference(rays_o, rays_d, z_vals, outgoing_bidxs, blend_weights, corners, sizes, res, tables_f16, params):
    """rendering_kernel.cu:872-1008, :1176-1208 (v1: all of a ray's outgoing blocks blended per sample)."""
    o, d, z = _f32(_np(rays_o)), _f32(_np(rays_d)), _f32(_np(z_vals))
    B, S = z.shape
    tb = np.ascontiguousarray(_np(tables_f16).view(np.uint16))
    T = tb.shape[2]
    dif, spec, al = np.zeros((B, S, 3), np.float32), np.zeros((B, S, 3), np.float32), np.zeros((B, S, 1), np.float32)
    lib().orc_bg_pts_inference(_p(o), _p(d), _p(z), _p(tb), _p(_f32(_np(params))), _p(_f32(_np(corners))), _p(_f32(_np(sizes))),
                               _p(_i32(_np(res))), _p(_i16(outgoing_bidxs)), _p(_f32(_np(blend_weights))), _p(dif), _p(spec), _p(al),
                               _ci(T), _ci(B), _ci(S))
    return dif, spec, al


def update_outgoing_bidx(rays_o, rays_d, corners, sizes, tracing_blocks, inter, ratio, skip):
    o, d = _f32(_np(rays_o)), _f32(_np(rays_d))
    B, nb = o.shape[0], np.asarray(corners).shape[0]
    ob = np.full((B, 4), -1, np.int16)
    bw = np.zeros((B, 4), np.float32)
    lib().orc_update_outgoing_bidx(_p(o), _p(d), _p(_f32(_np(corners))), _p(_f32(_np(sizes))), _p(_i32(_np(tracing_blocks))),
                                   _p(_f32(_np(inter))), _p(ob), _p(bw), _cf(ratio), _ci(int(skip)), _ci(nb), _ci(B))
    return ob, bw


def update_outgoing_bidx_v2(rays_o, corners, sizes):
    o = _f32(_np(rays_o))
    B, nb = o.shape[0], np.asarray(corners).shape[0]
    ob = np.full((B, 4), -1, np.int16)
    bw = np.zeros((B, 4), np.float32)
    lib().orc_update_outgoing_bidx_v2(_p(o), _p(_f32(_np(corners))), _p(_f32(_np(sizes))), _p(ob), _p(bw), _ci(nb), _ci(B))
    return ob, bw


def get_last_block(tracing_blocks, inter):
    tb = _i32(_np(tracing_blocks))
    out = np.full((tb.shape[0],), -1, np.int32)
    lib().orc_get_last_block(_p(tb), _p(out), _p(_f32(_np(inter))), _ci(tb.shape[1]), _ci(tb.shape[0]))
    return out


def ray_firsthit_block(rays_o, rays_d, corners, sizes, occ, grid_starts, log2dim, tracing_blocks, inter):
    """rendering_kernel.cu:705-813 -> hit_blockIdxs [B] int16 (pre-filled with -1 as the kernel expects)."""
    tb = _i32(_np(tracing_blocks))
    hit = np.full((tb.shape[0],), -1, np.int16)
    lib().orc_ray_firsthit_block(_p(_f32(_np(rays_o))), _p(_f32(_np(rays_d))), _p(_f32(_np(corners))), _p(_f32(_np(sizes))),
                                 _p(_u8(occ)), _p(_i64(grid_starts)), _p(_i32(_np(log2dim))), _p(tb), _p(_f32(_np(inter))),
                                 _p(hit), _ci(tb.shape[1]), _ci(tb.shape[0]))
    return hit


def process_occupied_grid(bidx, total_grid, corners, sizes, occ, grid_starts, log2dim, tgt):
    """In place on tgt (numpy uint8, concatenated grids)."""
    lib().orc_process_occupied_grid(_ci(bidx), _ci(total_grid), _p(_f32(_np(corners))), _p(_f32(_np(sizes))), _p(_u8(occ)),
                                    _p(_i64(grid_starts)), _p(_i32(_np(log2dim))), _p(tgt), _ci(np.asarray(corners).shape[0]))


class _EncodeBG(torch.autograd.Function):
    """hashgrid/PyHashGridBG.py:9-30 with the C oracle underneath (CPU tensors)."""

    @staticmethod
    def forward(ctx, points, features, res):
        ctx.save_for_backward(points, features, res)
        return torch.from_numpy(embedding_forward(points, features, res))

    @staticmethod
    def backward(ctx, g):
        points, features, res = ctx.saved_tensors
        gp, gf = embedding_backward(points, g.contiguous(), features, res)
        return torch.from_numpy(gp), torch.from_numpy(gf), None


def encode_bg(points, features, res):
    return _EncodeBG.apply(points, features, res)


# ---------------------------------------------------------------- torch (Python) half
def level_resolutions(base_res, finest_res, n_levels=16):
    """hashgrid/PyHashGridBG.py:53-62 (torch float32 arithmetic, .int() truncation)."""
    base = torch.as_tensor(base_res).float()
    fin = torch.as_tensor(finest_res).float()
    b = torch.exp((torch.log(fin) - torch.log(base)) / (n_levels - 1))
    return torch.stack([(base * b ** i).int() for i in range(n_levels)], 0)


_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435]


def sh_deg3(v):
    """network.py:38-77 at deg=3 -> [...,16]."""
    x, y, z = v[..., 0:1], v[..., 1:2], v[..., 2:3]
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    cols = [torch.ones_like(x) * _C0, _C1 * y, _C1 * z, _C1 * x,
            _C2[0] * xy, _C2[1] * yz, _C2[2] * (2.0 * zz - xx - yy), _C2[3] * xz, _C2[4] * (xx - yy),
            _C3[0] * y * (3 * xx - yy), _C3[1] * xy * z, _C3[2] * y * (4 * zz - xx - yy),
            _C3[3] * z * (2 * zz - 3 * xx - 3 * yy), _C3[4] * x * (4 * zz - xx - yy),
            _C3[5] * z * (xx - yy), _C3[6] * x * (xx - 3 * yy)]
    return torch.cat(cols, -1)


def gaussian_act(x, sigma=0.1):
    """network.py:79-84"""
    return torch.exp((x ** 2) * (1.0 / (-2 * sigma ** 2)))


# state-dict order of network.ShallowMLP (network.py:155-163): (name, out, in)
MLP_LAYERS = [("Spatial_MLP.mlp.0", 64, 32), ("Spatial_MLP.mlp.2", 64, 64), ("sigma_layer.mlp.0", 1, 32),
              ("diffuse_layer.mlp.0", 3, 32), ("tint_layer.mlp.0", 3, 32), ("Directional_MLP.mlp.0", 64, 48),
              ("Directional_MLP.mlp.2", 64, 64), ("Directional_MLP.mlp.4", 3, 64)]


def init_mlp(seed=0, in_channel=32, bias_scale=0.0):
    """Xavier-normal weights (network.py:202-205); optional non-zero bias for tests."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, o, i in MLP_LAYERS:
        if name == "Spatial_MLP.mlp.0":
            i = in_channel
        std = math.sqrt(2.0 / (i + o))
        sd[name + ".weight"] = torch.randn(o, i, generator=g) * std
        sd[name + ".bias"] = torch.randn(o, generator=g) * bias_scale
    return sd


def pack_blob(sd):
    """rendering.py:101-112 / tools/utils.py:399-410: per layer [bias, W^T flattened]."""
    parts = []
    for name, _, _ in MLP_LAYERS:
        parts += [sd[name + ".bias"].reshape(-1), sd[name + ".weight"].transpose(1, 0).reshape(-1)]
    return torch.cat(parts, 0).float()


def unpack_blob(blob, in_channel=32):
    sd, k = {}, 0
    blob = torch.as_tensor(blob)
    for name, o, i in MLP_LAYERS:
        if name == "Spatial_MLP.mlp.0":
            i = in_channel
        sd[name + ".bias"] = blob[k:k + o]
        k += o
        sd[name + ".weight"] = blob[k:k + i * o].reshape(i, o).transpose(1, 0)
        k += i * o
    return sd


def _lin(sd, name, x):
    return x @ sd[name + ".weight"].t() + sd[name + ".bias"]


def mlp_forward(sd, x, weight_feature):
    """network.ShallowMLP.forward, network.py:172-190.  x [...,32+3]."""
    feats, dirs = x[..., :-3], x[..., -3:]
    dirs = dirs / (dirs.norm(2, dim=-1, keepdim=True) + 1e-8)
    H = _lin(sd, "Spatial_MLP.mlp.2", gaussian_act(_lin(sd, "Spatial_MLP.mlp.0", feats * weight_feature)))
    sigma = torch.nn.functional.softplus(_lin(sd, "sigma_layer.mlp.0", H[..., :32]))
    tint = torch.sigmoid(_lin(sd, "tint_layer.mlp.0", H[..., :32]))
    c_d = torch.sigmoid(_lin(sd, "diffuse_layer.mlp.0", H[..., :32]))
    h = torch.cat([H[..., 32:], sh_deg3(dirs)], -1)
    h = gaussian_act(_lin(sd, "Directional_MLP.mlp.0", h))
    h = gaussian_act(_lin(sd, "Directional_MLP.mlp.2", h))
    c_s = torch.sigmoid(_lin(sd, "Directional_MLP.mlp.4", h))
    return {"diffuse": c_d, "specular": c_s, "sigma": sigma, "tint": tint}


def mlp_sigma(sd, feats):
    """network.py:168-170"""
    H = _lin(sd, "Spatial_MLP.mlp.2", gaussian_act(_lin(sd, "Spatial_MLP.mlp.0", feats)))
    return torch.nn.functional.softplus(_lin(sd, "sigma_layer.mlp.0", H[..., :32]))


def weight_feature(global_step):
    """hashgrid/__init__.py:228-235 -> [16]"""
    alpha = max(min(global_step / 10000 * 8 + 8, 16), 0)
    k = torch.arange(16, dtype=torch.float32)
    return (1 - (alpha - k).clamp_(min=0, max=1).mul_(np.pi).cos_()) / 2


def contract_fore(x, min_bbox, bbox_size):
    """hashgrid/__init__.py:394-395"""
    return (x - min_bbox) / bbox_size * 4.0 - 2.0


def contract_bg(x, min_bbox, bbox_size):
    """hashgrid/__init__.py:397-411"""
    x = (x - min_bbox) / bbox_size * 4.0 - 2.0
    linf, _ = torch.max(torch.abs(x), dim=-1, keepdim=True)
    temp = 2 - 1.0 / linf
    return x * (temp / linf)


def cal_integrate_weight(sigma, dists, rays_d, infinity=True):
    """hashgrid/__init__.py:344-360.  sigma [B,S,1], dists [B,S] -> weights [B,S,1], T_left [B]
    (T_left is the transmittance BEFORE the last sample: the reference quirk at :358-360)."""
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    if infinity:
        dists = dists.clone()
        dists[:, -1] = 1e10
    alpha = 1.0 - torch.exp(-sigma * dists[..., None])
    T = torch.cumprod(torch.cat([torch.ones((alpha.shape[0], 1, 1)), 1.0 - alpha + 1e-6], 1), 1)[:, :-1]
    return alpha * T, T[:, -1, 0]


def render_batch_rays(rays_o, rays_d, z_vals, dists, features, res, sd, mode, contract, global_step,
                      infinity=False):
    """hashgrid/__init__.py:512-596 (out_normal=False).  `contract` is a callable on [N,3]."""
    B, S = z_vals.shape
    L = int(res.shape[0])  # 16 in the reference (hard-coded, hashgrid/__init__.py:62,233); BASELINE configs[0] uses 8
    samples = rays_o[:, None, :] + z_vals[..., None] * rays_d[:, None, :]
    cx = contract(samples.reshape(-1, 3))
    feats = encode_bg(cx, features, res).reshape(B, S, 2 * L)
    wf = weight_feature(global_step)[:L][None, None, :].repeat_interleave(2, dim=-1)
    inputs = torch.cat([feats, rays_d[:, None, :].repeat(1, S, 1)], -1)
    o = mlp_forward(sd, inputs, wf)
    weights, T_left = cal_integrate_weight(o["sigma"], dists, rays_d, infinity=infinity)
    acc = lambda a: torch.sum(weights * a, 1)
    out = {"depth": acc(z_vals[..., None]), "tint": acc(o["tint"]), "diffuse": acc(o["diffuse"]),
           "specular": acc(o["tint"] * o["specular"]), "T_left": T_left, "weights": weights}
    out["rgb"] = torch.clamp(out["diffuse"] + out["specular"], 0, 1)
    if mode == TRAIN:
        out["l2_reg_specular"] = torch.mean(torch.sum(weights.detach() * (o["specular"] - 0) ** 2, 1))
    return out


def inverse_z_sampling(rays_o, rays_d, bbox_center, bbox_size, S, invalid_underground=True):
    """hashgrid/__init__.py:306-337 (+ :287-293).  bbox_* are the HashGrid (2x) box."""
    bounds = torch.from_numpy(ray_aabb_intersection(rays_o, rays_d, bbox_center, bbox_size / 2.0).copy())
    if invalid_underground:
        outgoing = rays_o + bounds[:, 1:] * rays_d
        corner = bbox_center - bbox_size / 4.0
        valid = ~(torch.abs(outgoing[:, 1] - corner[1]) < 0.0001)
    else:
        valid = torch.ones_like(rays_d[..., 0]).bool()
    bounds[torch.any(bounds == -1, dim=-1), 1:] = 0.1
    t = torch.linspace(0.0, 1.0, steps=S)[None, :]
    z = 1.0 / (1.0 / (bounds[:, 1:] + 1e-6) * (1.0 - t) + 1.0 / 1e6 * t)
    z = z.expand([rays_o.shape[0], S])
    d = z[:, 1:] - z[:, :-1]
    d = torch.cat([d, 1e-6 * torch.ones(d[..., :1].shape)], -1)
    return z, d, valid


class Tile:
    """The geometric state of hashgrid.HashGrid (hashgrid/__init__.py:33-92) that the path needs."""

    def __init__(self, corner, size, log2_T=19, grid_resolution=(32, 2048), sampler_log2dim=4, n_levels=16):
        corner = torch.as_tensor(corner, dtype=torch.float32)
        size = torch.as_tensor(size, dtype=torch.float32)
        self.bbox_center = corner + size / 2.0
        self.bbox_size = size * 2  # 2x for the background shell (:50)
        self.min_bbox = self.bbox_center - self.bbox_size / 2.0
        self.finest = (self.bbox_size / self.bbox_size.min() * grid_resolution[1]).int()
        self.base = (self.bbox_size / self.bbox_size.min() * grid_resolution[0]).int()
        self.res = level_resolutions(self.base, self.finest, n_levels)
        self.T = 2 ** log2_T
        self.log2dim = (sampler_log2dim - torch.log2(self.bbox_size.max() / self.bbox_size).int()).int()
        self.occ = torch.ones(tuple(int(2 ** k) for k in self.log2dim), dtype=torch.bool)  # voxelize.h:111-117
        self.occ_corner = self.min_bbox + self.bbox_size / 4.0  # :281-282
        self.occ_size = self.bbox_size / 2.0


def render_rays(tile, features, sd, rays_o, rays_d, S_fg, S_bg, mode, global_step, invalid_underground=False,
                occlusion_mask=None):
    """tile.py:639-692 with hashgrid/__init__.py:413-509 (BG_MODE 'IZ'; occlusion_mask [B,1] bool as tile.py:655,661 passes
    it: both branches' valid sets are ANDed with it, hashgrid/__init__.py:420-421,479-480).  Besides the merged prediction the
    result carries the two branches' own dictionaries (out["fg"], out["bg"]: what render_fore_rays / render_bg_rays return)."""
    B = rays_o.shape[0]
    z, d = sample_points_grid(rays_o, rays_d, tile.occ_corner, tile.occ_size, tile.occ, tile.log2dim, S_fg)
    z, d = torch.from_numpy(z), torch.from_numpy(d)
    valid = torch.all(z != -1, dim=-1)
    if occlusion_mask is not None:
        valid = valid & occlusion_mask[..., 0]
    out = {"fore_valid": valid}
    zeros3, ones1 = torch.zeros_like(rays_o), torch.ones_like(rays_d[..., :1])
    fg = {"rgb": zeros3.clone(), "depth": torch.zeros_like(ones1), "T_left": ones1.clone(),
          "specular": zeros3.clone(), "diffuse": zeros3.clone()}
    l2 = 0.0
    if valid.any():
        f = render_batch_rays(rays_o[valid], rays_d[valid], z[valid], d[valid], features, tile.res, sd, mode,
                              lambda x: contract_fore(x, tile.min_bbox, tile.bbox_size), global_step,
                              infinity=False)
        for k in ("rgb", "depth", "specular", "diffuse"):
            fg[k] = fg[k].index_put((valid,), f[k])
        fg["T_left"] = fg["T_left"].index_put((valid,), f["T_left"][:, None])
        out["fg_weights"] = f["weights"]
        if mode == TRAIN:
            l2 = l2 + f["l2_reg_specular"]
            out["fg_l2_reg_specular"] = f["l2_reg_specular"]
    zb, db, vb = inverse_z_sampling(rays_o, rays_d, tile.bbox_center, tile.bbox_size, S_bg, invalid_underground)
    if occlusion_mask is not None:
        vb = vb & occlusion_mask[..., 0]
    bg = {"rgb": zeros3.clone(), "depth": torch.zeros_like(ones1), "specular": zeros3.clone(),
          "diffuse": zeros3.clone()}
    bg_T = ones1.clone()
    if vb.any():
        g = render_batch_rays(rays_o[vb], rays_d[vb], zb[vb], db[vb], features, tile.res, sd, mode,
                              lambda x: contract_bg(x, tile.min_bbox, tile.bbox_size), global_step,
                              infinity=True)
        for k in bg:
            bg[k] = bg[k].index_put((vb,), g[k])
        bg_T = bg_T.index_put((vb,), g["T_left"][:, None])
        out["bg_weights"] = g["weights"]
        if mode == TRAIN:
            l2 = l2 + g["l2_reg_specular"]
            out["bg_l2_reg_specular"] = g["l2_reg_specular"]
    out["bg_valid"] = vb
    out["fg"], out["bg"] = dict(fg), dict(bg, T_left=bg_T)
    out["pred_color"] = fg["rgb"] + fg["T_left"] * bg["rgb"]
    out["pred_depth"] = fg["depth"] + fg["T_left"] * bg["depth"]
    out["pred_specular"] = fg["specular"] + fg["T_left"] * bg["specular"]
    out["pred_diffuse"] = fg["diffuse"] + fg["T_left"] * bg["diffuse"]
    out["T_left"] = fg["T_left"]
    out["l2_reg_specular"] = l2
    return out


# ------------------------------------------------------------------- a17: consensus
def consensus_reduce(tiles, num_camera, prev_shared=None):
    """admm_trainer.py:137-170.  tiles: list of dicts {pose [M,6], idx list[int], confidence [M]}.
    Returns shared [N,6], overlap mask [N] (count>=2), dual residual, primal residual."""
    acc = torch.zeros((num_camera, 6))
    cnt = torch.zeros((num_camera,), dtype=torch.int32)
    w = torch.zeros((num_camera,))
    for t in tiles:
        idx = torch.as_tensor(t["idx"], dtype=torch.long)
        cnt[idx] += 1
        w[idx] += t["confidence"]
        acc[idx] += t["confidence"][..., None] * t["pose"]
    w[w == 0] = 1
    shared = acc / w[..., None]
    prev = torch.zeros_like(shared) if prev_shared is None else prev_shared
    dual = torch.mean(torch.abs(prev - shared))
    primal = sum(torch.mean(torch.abs(t["pose"] - shared[torch.as_tensor(t["idx"], dtype=torch.long)]))
                 for t in tiles) / len(tiles)
    return shared, cnt >= 2, dual, primal


def consensus_update(se3_refine, shared_se3, delta_se3):
    """consensus.py:40-45: over-relaxed dual update."""
    return delta_se3 + (1 + 0.5) * (se3_refine - shared_se3)


def camera_loss(se3_refine, shared_se3, delta_se3, overlap_flags, rho):
    """consensus.py:70-76"""
    c = (se3_refine - shared_se3 + delta_se3) ** 2
    return torch.mean(rho[None, :] * c[overlap_flags])


def psnr(a, b):
    """tools/utils.py:53-55 on 0..255 images"""
    mse = np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)
    return 10.0 * math.log10(255.0 ** 2 / (mse + 1e-8))


# --------------------------------------------------------------------------- f1 / f4 restatements
def scheduler_eta(step, start_eta, end_eta, iterations, decay_rate=0.1, start_itr=0, end_itr=100000000):
    """scheduler.py:15-52 with decay_func2: eta = start * rate^(step / (iterations / log_rate(end/start)))."""
    if step < start_itr or step >= end_itr:
        return 0.0
    decay_steps = iterations / math.log(end_eta / start_eta, decay_rate)
    return start_eta * decay_rate ** (step / decay_steps)


def voxelize_mesh(vertices, faces, log2dim, block_corner, block_size, init_out):
    """cuda/include/voxelize.h:12-119 on arrays -> (vis, outside) bool grids."""
    l2d = _i32(log2dim)
    shape = tuple(1 << int(k) for k in l2d)
    vis, out = np.zeros(shape, np.uint8), np.zeros(shape, np.uint8)
    v, f = _f32(vertices).reshape(-1, 3), _i32(faces).reshape(-1, 3)
    lib().orc_voxelize_mesh(_p(v), _p(f), _ci(f.shape[0]), _p(l2d), _p(_f32(block_corner)), _p(_f32(block_size)), _p(vis),
                            _ci(int(bool(init_out))), _p(out))
    return vis.astype(bool), out.astype(bool)


def pruning_tile_grid(occupied_grid, log2dim, features, res, sd, bbox_size, global_step, sub_split, pruning_th,
                      finest_resolution=2048, batch_size=92 ** 3):
    """hashgrid/__init__.py:138-213 with the CPU encoder: -> (new_grid bool, new_log2dim)."""
    occ = torch.as_tensor(occupied_grid)
    log2dim = torch.as_tensor(log2dim).int() + (1 if sub_split else 0)
    scale = 2 if sub_split else 1
    grid_resolution = 2 ** log2dim
    bbox_size = torch.as_tensor(bbox_size, dtype=torch.float32)
    fin = (bbox_size / bbox_size.min() * finest_resolution).int()
    total_res = fin / 4.0 if global_step < 10000 else fin / 2.0
    sample_resolution = ((total_res / 2.0) / grid_resolution).int()
    xs, ys, zs = torch.where(occ.repeat_interleave(scale, 0).repeat_interleave(scale, 1).repeat_interleave(scale, 2))
    locs = torch.stack([xs, ys, zs], -1).long()
    grid_corner = locs / grid_resolution
    X, Y, Z = torch.meshgrid(torch.arange(0, int(sample_resolution[0])), torch.arange(0, int(sample_resolution[1])),
                             torch.arange(0, int(sample_resolution[2])), indexing="ij")
    grid_point = torch.stack([X, Y, Z], -1).reshape(-1, 3) / (sample_resolution * grid_resolution)
    run = max(int(batch_size / int(torch.prod(sample_resolution))), 1)
    alpha_res = torch.zeros(locs.shape[0])
    wf = weight_feature(global_step)[None, :].repeat_interleave(2, dim=-1)
    for i in range(0, locs.shape[0], run):
        pts = (grid_corner[i:i + run, None, :] + grid_point[None, ...]) * 2 - 1
        n = pts.shape[0]
        feats = encode_bg(pts.reshape(-1, 3).float(), features, res).reshape(-1, 32) * wf
        alpha = 1 - torch.exp(-1.0 * mlp_sigma(sd, feats))
        alpha_res[i:i + n] = alpha.reshape(n, -1).max(dim=-1)[0]
    new = torch.zeros(tuple(int(r) for r in grid_resolution), dtype=torch.bool)
    keep = locs[alpha_res > pruning_th]
    new[keep[:, 0], keep[:, 1], keep[:, 2]] = True
    return new, log2dim


def occlusion_mask(rays_o, rays_d, shared_depth_half, bbox_center, bbox_size_half, H, W, kernel_size=91):
    """tile.py:366-400 for one view: half-resolution shared depth [H/2,W/2,1] upsampled 2x (nearest), compared with
    the entry depth of the tile box (ray_aabb_intersection with size = bbox_size/2, i.e. half extents bbox_size/4
    ... as the reference passes it), the un-occluded set dilated by a kernel_size box filter.
    -> bool [H,W,1]: True = keep the pixel."""
    depth = torch.as_tensor(shared_depth_half).repeat_interleave(2, 0).repeat_interleave(2, 1).reshape(-1, 1)
    bounds = torch.from_numpy(ray_aabb_intersection(_np(rays_o), _np(rays_d), _np(bbox_center), _np(bbox_size_half)))
    occ = ((depth > bounds[..., :1]) & (bounds[..., :1] != -1)).reshape(1, 1, H, W)
    kernel = torch.ones((1, 1, kernel_size, kernel_size), dtype=torch.float32)
    occ = 1.0 - torch.nn.functional.conv2d(1.0 - occ.float(), kernel, padding=(kernel_size // 2, kernel_size // 2)).clamp(0, 1)
    return occ.bool().reshape(H, W, 1)
