"""Fused fast path: one launch per HashGrid.render_batch_rays (hashgrid/__init__.py:512-596)."""
import ctypes

import torch

from . import _capi
from ._capi import RAY_OUT, RenderCfg, check, dev_ptr, feat_dtype_code, lib, stream

_f32 = torch.float32
FORE, BG = 0, 1

# Decoder arithmetic of the fused kernels (module-level switch: set_arith(); it becomes scanerf_render_cfg.arith of every call):
#   "f32"  = the f32-input MFMA (exact f32);
#   "h3"   = f16 matrix cores on hi/lo-split operands, three products per term, f32 accumulate: results as close to fp64 as
#            the f32 evaluation (csrc/render_h3.h), forward and backward (32-sample tiles, one wave per SIMD);
#   "t16s" = (default) forward as "h3"; backward on 16-sample tiles at two waves per SIMD with EVERY product split like h3's
#            (csrc/render_t16.h): f32-equivalent gradients (4e-6 relative L2 against the oracle) at 5.5 instead of 6.2 ms;
#   "t16"  = the same backward with the gradient products on ONE f16 MFMA per term and 8-byte table-gradient records
#            (7e-4 relative L2): the fast, reduced-precision option (3.9 ms); converges like the others on the procedural
#            scene (tests/test_gpu_harness.py: 27.1 dB each) but is not what the reference computes.
#   Calls the 16-sample-tile kernels cannot serve (no x-stash) run the "h3" backward: see backward_arith().
_ARITH_CODES = {"f32": _capi.ARITH_F32, "h3": _capi.ARITH_H3, "t16": _capi.ARITH_T16, "t16s": _capi.ARITH_T16S}
ARITH_NAMES = tuple(_ARITH_CODES)
FP32_EQUIV_ARITH = "t16s"   # the fastest arithmetic whose gradients are f32-equivalent: what bench.py's headline runs
DEFAULT_ARITH = FP32_EQUIV_ARITH
ARITH = _ARITH_CODES[DEFAULT_ARITH]
# what each arithmetic computes in, for bench.py's `dtype`
ARITH_DTYPE = {
    "f32": "f32 throughout (f32-input MFMA; f32 table-gradient records summed in 64-bit fixed point)",
    "h3": "f32 tables / compositing / accumulate; every decoder product (forward, gradient chains, weight gradients) on f16 MFMA "
          "with hi+lo split operands (22-bit), three products per term, f32 accumulate; f32 table-gradient records summed in "
          "64-bit fixed point",
    "t16s": "f32 tables / compositing / accumulate; every decoder product (forward, gradient chains, weight gradients) on f16 MFMA "
            "with hi+lo split operands (22-bit), three products per term, f32 accumulate; G' in f32; f32 table-gradient records "
            "summed in 64-bit fixed point",
    "t16": "f32 tables / compositing; forward + backward recompute split-f16 x3 MFMA (22-bit operands); gradient products ONE f16 "
           "MFMA per term (11-bit operands); 13-bit table-gradient records summed in 64-bit fixed point -- REDUCED precision",
}


def set_arith(name):
    global ARITH
    ARITH = _ARITH_CODES[name]


def arith_name():
    return next(k for k, v in _ARITH_CODES.items() if v == ARITH)


def backward_arith(have_xstash=True, pose_grads=False):
    """The arithmetic code one training step's scatter_plan / render_backward / scatter_accumulate must agree on
    (t16 needs the forward's x-stash; it produces the pose-gradient sums as well)."""
    if ARITH in _capi.T16_FAMILY and not have_xstash:
        return _capi.ARITH_H3
    return ARITH


def compact_record_format(arith_code):
    """`compact_records` of scatter_table_grad_adam for feature gradients out of a backward run under `arith_code`:
    8-byte records behind t16 (its gradients are f16 products anyway), 12-byte ones behind t16s (f32-grade), else the
    16-byte ones."""
    return {_capi.ARITH_T16: 1, _capi.ARITH_T16S: 2}.get(arith_code, 0)


def tile_T_columns(S):
    """Columns of the forward's tile_T output: transmittance entering each 16-sample tile."""
    return (S + 15) // 16


# columns of out_ray [B,16]
RGB, DEPTH, T_LEFT, DIFFUSE, SPECULAR, TINT, W_SPEC2 = slice(0, 3), 3, 4, slice(5, 8), slice(8, 11), slice(11, 14), 14


class PackedDecoder:
    """Device workspace holding the decoder in the layout the fused kernels stage into LDS.
    Re-pack after every optimiser step on the decoder or when weight_feature changes."""

    def __init__(self, device):
        self.workspace = torch.empty(lib().scanerf_render_workspace_floats(), dtype=_f32, device=device)

    skip_levels = 0

    def pack(self, blob, weight_feature, skip_levels=0):
        """skip_levels: bit l = weight_feature is exactly zero on level l (network.skip_levels): render_forward then leaves
        that level's table alone."""
        if blob.numel() != _capi.PARAMSIZE or weight_feature.numel() != 32:
            raise RuntimeError(f"scanerf: blob must hold {_capi.PARAMSIZE} floats and weight_feature 32")
        self.skip_levels = int(skip_levels)
        check(lib().scanerf_pack_decoder(dev_ptr(blob, _f32, "mlp_blob"), dev_ptr(weight_feature, _f32, "weight_feature"),
                                         dev_ptr(self.workspace, _f32, "workspace"), stream()), "pack_decoder")
        return self


def _cfg(min_bbox, bbox_size, contract_mode, infinity, arith=None, skip_levels=0):
    c = RenderCfg()
    c.contract_mode, c.infinity = int(contract_mode), int(bool(infinity))
    c.arith = ARITH if arith is None else arith
    c.skip_levels = int(skip_levels)
    for k in range(3):
        c.min_bbox[k] = float(min_bbox[k])
        c.bbox_size[k] = float(bbox_size[k])
    return c


JSTASH_DTYPE = torch.int32


def jstash_shape(B, S):
    """Shape of render_forward's jstash output: per ray, 32-sample tile, level of the half-wave (8), word (4), forward lane (64):
    the six position Jacobians d(feature)/d(p) of a (sample, level) as 20-bit significands under one exponent, 16 bytes
    (csrc/render_device.h jst_pack)."""
    return (B, (S + 31) // 32, 8, 4, 64)


def forward_plan_supported(B, S, T):
    """render_forward(plan=True) can do scatter_plan's work for the t16 backward of the same rays (equal kernel grids)."""
    return bool(lib().scanerf_render_forward_plan_supported(ctypes.c_int(B), ctypes.c_int(S), ctypes.c_int(T)))


def render_forward(rays_o, rays_d, z_vals, dists, features, resolutions, packed, min_bbox, bbox_size, contract_mode,
                   infinity, ray_valid=None, want_weights=True, out_ray=None, weights=None, tile_T=None, xstash=None,
                   plan=False, plan_workspace=None, jstash=None):
    """-> out_ray [B,16] (see column constants), weights [B,S] or None.
    min_bbox / bbox_size: host sequences of 3 floats (the HashGrid 2x box).
    plan=True (only where forward_plan_supported): the launch also reserves the t16 backward's scatter-record ranges for these
    rays -- call it INSTEAD of scatter_plan; returns (out_ray, weights, workspace).
    jstash (fp32 tables): jstash_shape(B, S) JSTASH_DTYPE (packed words) that receives the encoder's position Jacobians, for
    render_backward(ray_pos_grad=...) (pose refinement without a second pass over the table)."""
    B, S = z_vals.shape
    if features.shape[0] != 16 or features.shape[2] != 2:
        raise RuntimeError("scanerf: fused path needs a [16,T,2] table (the reference hard-codes 16 levels)")
    if out_ray is None:
        out_ray = torch.empty((B, RAY_OUT), dtype=_f32, device=z_vals.device)
    if weights is None and want_weights:
        weights = torch.empty((B, S), dtype=_f32, device=z_vals.device)
    T = features.shape[1]
    args = (dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(rays_d, _f32, "rays_d"), dev_ptr(z_vals, _f32, "z_vals"),
            dev_ptr(dists, _f32, "dists"), dev_ptr(features, (torch.float32, torch.float16, torch.bfloat16), "features"),
            ctypes.c_int(feat_dtype_code(features)), dev_ptr(resolutions, torch.int32, "resolutions"),
            dev_ptr(packed.workspace, _f32, "workspace"))
    tail = (dev_ptr(ray_valid, (torch.bool, torch.uint8), "ray_valid", allow_none=True), dev_ptr(out_ray, _f32, "out_ray"),
            dev_ptr(weights, _f32, "weights", allow_none=True), dev_ptr(tile_T, _f32, "tile_T", allow_none=True),
            dev_ptr(xstash, _f32, "xstash", allow_none=True), ctypes.c_int(B), ctypes.c_int(S), ctypes.c_int(T))
    if plan or jstash is not None:
        ws = None
        if plan:
            need = lib().scanerf_render_scatter_workspace_bytes(ctypes.c_int(B), ctypes.c_int(S), ctypes.c_int(T))
            if not need or not forward_plan_supported(B, S, T):
                raise RuntimeError(f"scanerf: render_forward(plan=True) does not support B={B} S={S} T={T}")
            ws = _capi.workspace(z_vals.device, need) if plan_workspace is None else plan_workspace
        cfg = _cfg(min_bbox, bbox_size, contract_mode, infinity, ARITH if ARITH in _capi.T16_FAMILY else _capi.ARITH_T16,
                   getattr(packed, "skip_levels", 0))
        tail = tail[:5] + (dev_ptr(jstash, JSTASH_DTYPE, "jstash", allow_none=True),) + tail[5:]
        check(lib().scanerf_render_forward_packed_plan(*args, ctypes.byref(cfg), *tail,
                                                       ctypes.c_void_p(ws.data_ptr() if ws is not None else None),
                                                       ctypes.c_size_t(ws.numel() if ws is not None else 0), stream()),
              "render_forward(plan)")
        return (out_ray, weights, ws) if plan else (out_ray, weights)
    cfg = _cfg(min_bbox, bbox_size, contract_mode, infinity, skip_levels=getattr(packed, "skip_levels", 0))
    check(lib().scanerf_render_forward_packed(*args, ctypes.byref(cfg), *tail, stream()), "render_forward")
    return out_ray, weights


_KEEP_DW_PARTIAL = None


def render_backward(rays_o, rays_d, z_vals, dists, features, resolutions, packed, weight_feature, min_bbox, bbox_size,
                    contract_mode, infinity, out_ray, tile_T, grad_out, ray_valid=None, grad_blob=None, xstash=None,
                    ray_grad_buffers=None, scatter=None, want_dfeat=True, arith=None, jstash=None, ray_pos_grad=None):
    """Adjoint of render_forward -> (dfeat [16, B*S, 2] level-major, grad_blob [13994]).
    scatter = (workspace, grad_features) from scatter_plan(): the kernel emits the table-gradient records
    itself; finish with scatter_accumulate().  dfeat is then only produced if want_dfeat.
    arith: as given to scatter_plan (default: backward_arith() of this call's inputs).
    jstash + ray_pos_grad ([B,6] zeros; t16 kernel, with ray_grad_buffers): the kernel also writes dL/d(rays_o), dL/d(rays_d)
    through the sample positions, from the forward's position Jacobians (see ray_gradients_fused)."""
    B, S = z_vals.shape
    dev = z_vals.device
    if arith is None:
        arith = backward_arith(xstash is not None, ray_grad_buffers is not None)
    if scatter is None and not want_dfeat:
        raise ValueError("render_backward: nothing would receive the feature gradients")
    dfeat = torch.empty((16, B * S, 2), dtype=_f32, device=dev) if want_dfeat else None
    nblk = lib().scanerf_render_backward_grid(ctypes.c_int(B))
    dw_partial = torch.empty((4 * nblk, _capi.PARAMSIZE), dtype=_f32, device=dev)
    if _KEEP_DW_PARTIAL is not None:   # investigation hook (tools/bwd_stamps.py: the -DT16_STAMPS build parks its cycle sums there)
        _KEEP_DW_PARTIAL[:] = [dw_partial, nblk]
    if grad_blob is None:
        grad_blob = torch.zeros(_capi.PARAMSIZE, dtype=_f32, device=dev)
    cfg = _cfg(min_bbox, bbox_size, contract_mode, infinity, arith)
    check(lib().scanerf_render_backward(
        dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(rays_d, _f32, "rays_d"), dev_ptr(z_vals, _f32, "z_vals"),
        dev_ptr(dists, _f32, "dists"), dev_ptr(features, (torch.float32, torch.float16, torch.bfloat16), "features"),
        ctypes.c_int(feat_dtype_code(features)), dev_ptr(resolutions, torch.int32, "resolutions"),
        dev_ptr(packed.workspace, _f32, "workspace"), dev_ptr(weight_feature, _f32, "weight_feature"),
        ctypes.byref(cfg), dev_ptr(ray_valid, (torch.bool, torch.uint8), "ray_valid", allow_none=True),
        dev_ptr(out_ray, _f32, "out_ray"), dev_ptr(tile_T, _f32, "tile_T"), dev_ptr(grad_out, _f32, "grad_out"),
        dev_ptr(xstash, _f32, "xstash", allow_none=True), dev_ptr(dfeat, _f32, "dfeat", allow_none=True), dev_ptr(dw_partial, _f32, "dw_partial"), dev_ptr(grad_blob, _f32, "grad_blob"),
        dev_ptr(ray_grad_buffers[0] if ray_grad_buffers else None, _f32, "g_dnorm", allow_none=True),
        dev_ptr(ray_grad_buffers[1] if ray_grad_buffers else None, _f32, "g_rowsum", allow_none=True),
        dev_ptr(jstash, JSTASH_DTYPE, "jstash", allow_none=True), dev_ptr(ray_pos_grad, _f32, "ray_pos_grad", allow_none=True),
        ctypes.c_void_p(scatter[0].data_ptr() if scatter else None), ctypes.c_size_t(scatter[0].numel() if scatter else 0),
        dev_ptr(scatter[1] if scatter else None, _f32, "grad_features", allow_none=True),
        ctypes.c_int(B), ctypes.c_int(S), ctypes.c_int(features.shape[1]), stream()), "render_backward")
    return dfeat, grad_blob


def photometric_loss_grad(out_ray, target, ray_valid=None, reg_weight=0.01, grad_out=None):
    """MSE over the valid rays' rgb + reg_weight * l2_reg_specular (criterions.py:142-144, tile.py:999) and its gradient
    w.r.t. out_ray, without a torch graph -> (loss [1] device tensor, grad_out [B,16])."""
    B = out_ray.shape[0]
    dev = out_ray.device
    if grad_out is None:
        grad_out = torch.empty((B, RAY_OUT), dtype=_f32, device=dev)
    loss = torch.empty(1, dtype=_f32, device=dev)
    scratch = torch.empty(lib().scanerf_photometric_loss_scratch_floats(), dtype=_f32, device=dev)
    check(lib().scanerf_photometric_loss_grad(
        dev_ptr(out_ray, _f32, "out_ray"), dev_ptr(target, _f32, "target"),
        dev_ptr(ray_valid, (torch.bool, torch.uint8), "ray_valid", allow_none=True), ctypes.c_float(reg_weight),
        dev_ptr(grad_out, _f32, "grad_out"), dev_ptr(loss, _f32, "loss"), dev_ptr(scratch, _f32, "scratch"), ctypes.c_int(B),
        stream()), "photometric_loss_grad")
    return loss, grad_out


def photometric_loss_grad_fgbg(out_fg, out_bg, target, valid_fg=None, valid_bg=None, reg_weight=0.01):
    """Loss of the complete per-tile render (tile.py:666-690: pred = fg.rgb + fg.T_left * bg.rgb; MSE over the rays valid in either branch, criterions.py:121-138, +
    reg_weight * both branches' l2_reg_specular, tile.py:999) and its gradients w.r.t. the two branches' out_ray, without a
    torch graph -> (loss [1], grad_fg [B,16], grad_bg [B,16])."""
    B = out_fg.shape[0]
    dev = out_fg.device
    gfg, gbg = torch.empty((B, RAY_OUT), dtype=_f32, device=dev), torch.empty((B, RAY_OUT), dtype=_f32, device=dev)
    loss = torch.empty(1, dtype=_f32, device=dev)
    scratch = torch.empty(lib().scanerf_photometric_loss_scratch_floats(), dtype=_f32, device=dev)
    vt = (torch.bool, torch.uint8)
    check(lib().scanerf_photometric_loss_grad_fgbg(
        dev_ptr(out_fg, _f32, "out_fg"), dev_ptr(out_bg, _f32, "out_bg"), dev_ptr(target, _f32, "target"),
        dev_ptr(valid_fg, vt, "valid_fg", allow_none=True), dev_ptr(valid_bg, vt, "valid_bg", allow_none=True),
        ctypes.c_float(reg_weight), dev_ptr(gfg, _f32, "grad_fg"), dev_ptr(gbg, _f32, "grad_bg"), dev_ptr(loss, _f32, "loss"),
        dev_ptr(scratch, _f32, "scratch"), ctypes.c_int(B), stream()), "photometric_loss_grad_fgbg")
    return loss, gfg, gbg


def scatter_accumulate_adam2(ws1, S1, ws2, S2, params, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, B, half_table=None,
                             overflow_grad=None):
    """scatter_accumulate_adam over the record sets of TWO fused backward launches on the same rays and table (a tile's
    foreground and background branches): both gradients meet in one Adam step."""
    check(lib().scanerf_render_scatter_accumulate_adam2(
        dev_ptr(params, _f32, "params"), dev_ptr(exp_avg, _f32, "exp_avg"), dev_ptr(exp_avg_sq, _f32, "exp_avg_sq"),
        dev_ptr(half_table, (torch.float16, torch.bfloat16), "half_table", allow_none=True),
        ctypes.c_int(feat_dtype_code(half_table) if half_table is not None else 0),
        dev_ptr(overflow_grad, _f32, "overflow_grad", allow_none=True), ctypes.c_float(lr), ctypes.c_float(beta1),
        ctypes.c_float(beta2), ctypes.c_float(eps), ctypes.c_int(step), ctypes.c_int(B), ctypes.c_int(params.shape[1]),
        ctypes.c_int(S1), ctypes.c_void_p(ws1.data_ptr()), ctypes.c_size_t(ws1.numel()), ctypes.c_int(S2),
        ctypes.c_void_p(ws2.data_ptr()), ctypes.c_size_t(ws2.numel()), stream()), "scatter_accumulate_adam2")


def ray_valid(z_vals):
    """[B] uint8: every sample of the ray's row != -1 (hashgrid/__init__.py:419 torch.all(z_vals != -1, dim=-1))."""
    B, S = z_vals.shape
    n = (B + 15) // 16 * 16  # scanerf_compact_rays reads the flags 16 bytes at a time
    buf = torch.zeros(n, dtype=torch.uint8, device=z_vals.device)
    check(lib().scanerf_ray_valid(dev_ptr(z_vals, _f32, "z_vals"), dev_ptr(buf, torch.uint8, "valid"), ctypes.c_int(B),
                                  ctypes.c_int(S), stream()), "ray_valid")
    return buf[:B]


def compact_rays(valid, rays_o, rays_d, target, z_vals, dists, want_index=False):
    """Order-preserving compaction of the valid rays (hashgrid/__init__.py:419-434) in one launch (wave ballot / popcount
    prefix sums): -> (count, rays_o, rays_d, target, z_vals, dists[, index]) with the valid rays' rows first; the returned
    tensors are views of length `count` (one host read of the count, as torch's boolean-mask indexing needs too).
    valid: from ray_valid()."""
    B, S = z_vals.shape
    dev = z_vals.device
    if valid.dtype not in (torch.uint8, torch.bool) or valid.data_ptr() % 16:
        raise RuntimeError("scanerf: compact_rays needs the 16-byte aligned uint8 flags ray_valid() returns")
    o, d = torch.empty_like(rays_o), torch.empty_like(rays_d)
    t = torch.empty_like(target) if target is not None else None
    z, di = torch.empty_like(z_vals), torch.empty_like(dists)
    idx = torch.empty(B, dtype=torch.int32, device=dev) if want_index else None
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib().scanerf_compact_rays(
        dev_ptr(valid, (torch.uint8, torch.bool), "valid"), ctypes.c_int(B), ctypes.c_int(S), dev_ptr(rays_o, _f32, "rays_o"),
        dev_ptr(rays_d, _f32, "rays_d"), dev_ptr(target, _f32, "target", allow_none=True), dev_ptr(z_vals, _f32, "z_vals"),
        dev_ptr(dists, _f32, "dists"), dev_ptr(o, _f32, "out_o"), dev_ptr(d, _f32, "out_d"), dev_ptr(t, _f32, "out_t", allow_none=True),
        dev_ptr(z, _f32, "out_z"), dev_ptr(di, _f32, "out_dist"), dev_ptr(idx, torch.int32, "out_index", allow_none=True),
        dev_ptr(count, torch.int32, "count"), stream()), "compact_rays")
    n = int(count.item())
    res = (n, o[:n], d[:n], t[:n] if t is not None else None, z[:n], di[:n])
    return res + (idx[:n],) if want_index else res


def scatter_supported(B, S, T):
    return lib().scanerf_render_scatter_workspace_bytes(ctypes.c_int(B), ctypes.c_int(S), ctypes.c_int(T)) != 0


def scatter_plan(rays_o, rays_d, z_vals, resolutions, T, min_bbox, bbox_size, contract_mode, infinity, ray_valid=None,
                 arith=None, workspace=None, skip_levels=0):
    """Reserve the record ranges of the fused table-gradient path for this batch (count + scan).
    Returns the workspace tensor to hand to render_backward(scatter=(ws, grad_features)) and scatter_accumulate.
    The workspace is a per-(device, stream) cache: one plan/backward/accumulate sequence at a time.
    arith: the backward kernel that will emit the records (its ray -> workgroup map is reserved here); default
    backward_arith() = what render_backward picks when it is given an x-stash and no pose-gradient buffers."""
    B, S = z_vals.shape
    need = lib().scanerf_render_scatter_workspace_bytes(ctypes.c_int(B), ctypes.c_int(S), ctypes.c_int(T))
    if not need:
        raise RuntimeError(f"scanerf: fused scatter does not support B={B} S={S} T={T}")
    # (workspace: a caller-owned uint8 tensor instead of the cached one; smaller than `need` = overflow records take the
    # atomic path -- tests)
    ws = _capi.workspace(z_vals.device, need) if workspace is None else workspace
    # skip_levels (PackedDecoder.skip_levels): masked levels get no records; the plan notes it in the workspace for the backward
    cfg = _cfg(min_bbox, bbox_size, contract_mode, infinity, backward_arith() if arith is None else arith, skip_levels)
    check(lib().scanerf_render_scatter_plan(
        dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(rays_d, _f32, "rays_d"), dev_ptr(z_vals, _f32, "z_vals"),
        dev_ptr(resolutions, torch.int32, "resolutions"), ctypes.byref(cfg),
        dev_ptr(ray_valid, (torch.bool, torch.uint8), "ray_valid", allow_none=True), ctypes.c_int(B), ctypes.c_int(S),
        ctypes.c_int(T), ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), stream()), "scatter_plan")
    return ws


def scatter_accumulate(ws, grad_features, B, S):
    """grad_features [16,T,2] += the records the fused backward emitted into ws."""
    check(lib().scanerf_render_scatter_accumulate(
        dev_ptr(grad_features, _f32, "grad_features"), ctypes.c_int(B), ctypes.c_int(S),
        ctypes.c_int(grad_features.shape[1]), ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), stream()),
        "scatter_accumulate")
    return grad_features


def scatter_accumulate_adam(ws, params, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, B, S, half_table=None,
                            overflow_grad=None):
    """The records the fused backward emitted into ws applied straight to the table: accumulate + fused sparse Adam in one
    pass (no gradient table, no zero-fill, no scan of it).  `step` = the previous step count (cuda/adam_kernel.cu:83).
    half_table: optional f16 / bf16 gather copy of params, refreshed where params change; overflow_grad: the zero table
    handed to render_backward as grad_features (touched only if the record workspace overflows)."""
    check(lib().scanerf_render_scatter_accumulate_adam(
        dev_ptr(params, _f32, "params"), dev_ptr(exp_avg, _f32, "exp_avg"), dev_ptr(exp_avg_sq, _f32, "exp_avg_sq"),
        dev_ptr(half_table, (torch.float16, torch.bfloat16), "half_table", allow_none=True),
        ctypes.c_int(feat_dtype_code(half_table) if half_table is not None else 0),
        dev_ptr(overflow_grad, _f32, "overflow_grad", allow_none=True), ctypes.c_float(lr), ctypes.c_float(beta1),
        ctypes.c_float(beta2), ctypes.c_float(eps), ctypes.c_int(step), ctypes.c_int(B), ctypes.c_int(S),
        ctypes.c_int(params.shape[1]), ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), stream()),
        "scatter_accumulate_adam")


def scatter_table_grad(points, dfeat, grad_features, resolutions, compact_records=-1):
    """grad_features [16,T,2] += binned scatter of level-major dfeat at contracted `points` [N,3].
    compact_records: -1 = the layout's default (16-byte records for level-major gradients), 0 / 1 / 2 = 16- / 8- / 12-byte records."""
    N, (L, T) = points.shape[0], grad_features.shape[:2]
    need = lib().scanerf_embedding_bwd_workspace_bytes(ctypes.c_int(N), ctypes.c_int(L), ctypes.c_int(T))
    if not need:
        raise RuntimeError("scanerf: shape not supported by the binned scatter")
    ws = _capi.workspace(points.device, need, "scatter")
    check(lib().scanerf_embedding_bg_backward_binned(
        dev_ptr(points, _f32, "points"), dev_ptr(dfeat, _f32, "dfeat"), dev_ptr(grad_features, _f32, "grad_features"),
        dev_ptr(resolutions, torch.int32, "resolutions"), ctypes.c_int(N), ctypes.c_int(L), ctypes.c_int(T),
        ctypes.c_int(1), ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), ctypes.c_int(int(compact_records)), stream()),
        "scatter_table_grad")
    return grad_features


def scatter_table_grad_adam(points, dfeat, resolutions, params, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step,
                            half_table=None, overflow_grad=None, compact_records=0):
    """scatter_table_grad ending in the fused sparse Adam (no gradient table): the table path of tables too large for the
    backward kernel's own record emission.  overflow_grad: zero table like params (required by the C ABI).
    compact_records: 1 = 8-byte records (for dfeat out of the t16 backward; scatter_common.h Rec8), 2 = 12-byte records
    (f32-grade, behind the t16s backward; Rec12), 0 = 16-byte records."""
    N, (L, T) = points.shape[0], params.shape[:2]
    need = lib().scanerf_embedding_bwd_workspace_bytes(ctypes.c_int(N), ctypes.c_int(L), ctypes.c_int(T))
    if not need:
        raise RuntimeError("scanerf: shape not supported by the binned scatter")
    ws = _capi.workspace(points.device, need, "scatter")
    check(lib().scanerf_embedding_bg_backward_binned_adam(
        dev_ptr(points, _f32, "points"), dev_ptr(dfeat, _f32, "dfeat"), dev_ptr(resolutions, torch.int32, "resolutions"),
        ctypes.c_int(N), ctypes.c_int(L), ctypes.c_int(T), ctypes.c_int(1), ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()),
        dev_ptr(params, _f32, "params"), dev_ptr(exp_avg, _f32, "exp_avg"), dev_ptr(exp_avg_sq, _f32, "exp_avg_sq"),
        dev_ptr(half_table, (torch.float16, torch.bfloat16), "half_table", allow_none=True),
        ctypes.c_int(feat_dtype_code(half_table) if half_table is not None else 0), dev_ptr(overflow_grad, _f32, "overflow_grad"),
        ctypes.c_float(lr), ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_float(eps), ctypes.c_int(step),
        ctypes.c_int(int(compact_records)), stream()), "scatter_table_grad_adam")


RAYS_SCATTER = True   # (False: tile_model builds contracted points in torch and concatenates the branches, as rounds 1-5 did)


def scatter_rays_supported(T, arith_code):
    """scatter_table_grad_adam_rays applies: tables of at least 2^22 entries per level (one level's counters in the producer's LDS
    at a time) behind the f32-grade backward (12-byte records)."""
    return T >= (1 << 22) and compact_record_format(arith_code) == 2 and RAYS_SCATTER


def scatter_table_grad_adam_rays(rays_o, rays_d, branches, min_bbox, bbox_size, resolutions, params, exp_avg, exp_avg_sq, lr, beta1,
                                 beta2, eps, step, half_table=None, overflow_grad=None, fp16_moments=False):
    """The table gradient of one or two render branches over the same rays, scattered and applied by ONE sparse Adam step
    (scanerf_table_grad_scatter_adam_rays; tile.py:639-692, :1010).  branches: [(z [B,S], dfeat [16,B*S,2], ray_valid [B] or None,
    contract mode FORE / BG), ...] -- no contracted-point tensors, no concatenation.
    fp16_moments (opt-in): exp_avg / exp_avg_sq are float16 tensors, the update is adam_step_cuda_fp16's (cuda/adam_kernel.cu:98-144)."""
    mdt = torch.float16 if fp16_moments else _f32
    B, T = rays_o.shape[0], params.shape[1]
    if not 1 <= len(branches) <= 2 or params.shape[0] != 16:
        raise RuntimeError("scanerf: scatter_table_grad_adam_rays takes one or two branches and 16 levels")
    N = sum(B * z.shape[1] for z, _, _, _ in branches)
    need = lib().scanerf_embedding_bwd_workspace_bytes(ctypes.c_int(N), ctypes.c_int(16), ctypes.c_int(T))
    if not need:
        raise RuntimeError("scanerf: shape not supported by the binned scatter")
    ws = _capi.workspace(rays_o.device, need, "scatter")
    args = []
    for k in range(2):
        if k < len(branches):
            z, dfeat, valid, mode = branches[k]
            if tuple(dfeat.shape) != (16, B * z.shape[1], 2) or z.shape[0] != B:
                raise RuntimeError(f"scanerf: branch {k}: dfeat {tuple(dfeat.shape)} does not match z {tuple(z.shape)}")
            args += [dev_ptr(z, _f32, f"z{k}"), dev_ptr(dfeat, _f32, f"dfeat{k}"),
                     dev_ptr(valid, (torch.bool, torch.uint8), f"valid{k}", allow_none=True), ctypes.c_int(z.shape[1]), ctypes.c_int(int(mode))]
        else:
            args += [None, None, None, ctypes.c_int(0), ctypes.c_int(0)]
    mn, sz = (ctypes.c_float * 3)(*[float(v) for v in min_bbox]), (ctypes.c_float * 3)(*[float(v) for v in bbox_size])
    check(lib().scanerf_table_grad_scatter_adam_rays(
        dev_ptr(rays_o, _f32, "rays_o"), dev_ptr(rays_d, _f32, "rays_d"), ctypes.c_int(B), *args, mn, sz,
        dev_ptr(resolutions, torch.int32, "resolutions"), ctypes.c_int(T), ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()),
        dev_ptr(params, _f32, "params"), dev_ptr(exp_avg, mdt, "exp_avg"), dev_ptr(exp_avg_sq, mdt, "exp_avg_sq"),
        dev_ptr(half_table, (torch.float16, torch.bfloat16), "half_table", allow_none=True),
        ctypes.c_int(feat_dtype_code(half_table) if half_table is not None else 0), dev_ptr(overflow_grad, _f32, "overflow_grad"),
        ctypes.c_float(lr), ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_float(eps), ctypes.c_int(step),
        ctypes.c_int(1 if fp16_moments else 0), stream()), "scatter_table_grad_adam_rays")


def ray_gradients_fused(rays_o, rays_d, blob, ray_pos_grad, g_dnorm, g_rowsum, ray_valid=None):
    """dL/d(rays_o), dL/d(rays_d) when the backward kernel produced the position path itself (render_backward(jstash=...,
    ray_pos_grad=...)): adds the two per-ray paths -- |d| through delta = dist * |d| and SH(d / |d|) of the decoder -- in one
    launch (scanerf_ray_grad_epilogue).  Same result as ray_gradients() / ray_gradients_fused_autograd()."""
    B = rays_d.shape[0]
    g_o, g_d = torch.empty_like(rays_o, dtype=_f32), torch.empty_like(rays_d, dtype=_f32)
    S = g_dnorm.shape[1] * 32
    check(lib().scanerf_ray_grad_epilogue(
        dev_ptr(rays_d, _f32, "rays_d"), dev_ptr(blob.detach(), _f32, "mlp_blob"), dev_ptr(ray_pos_grad, _f32, "ray_pos_grad"),
        dev_ptr(g_dnorm, _f32, "g_dnorm"), dev_ptr(g_rowsum, _f32, "g_rowsum"),
        dev_ptr(ray_valid, (torch.bool, torch.uint8), "ray_valid", allow_none=True), dev_ptr(g_o, _f32, "g_o"), dev_ptr(g_d, _f32, "g_d"),
        ctypes.c_int(B), ctypes.c_int(S), stream()), "ray_grad_epilogue")
    return g_o, g_d


def ray_gradients_fused_autograd(rays_o, rays_d, blob, ray_pos_grad, g_dnorm, g_rowsum, ray_valid=None):
    """ray_gradients_fused by torch autograd on [B]-sized tensors (what it was before the epilogue kernel: ~150 small launches);
    kept as the reference the tests compare the kernel with."""
    from . import tile_model
    d = rays_d.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        dn = d.norm(2, dim=-1, keepdim=True)
        sh = tile_model.sh3(d / (dn + 1e-8))
    w_sh = blob[6503 + 64 + 32 * 64: 6503 + 64 + 48 * 64].reshape(16, 64)  # Directional_MLP.mlp.0 weight^T rows 32..47
    g_sh = g_rowsum.sum(1) @ w_sh.t()
    g_dn = g_dnorm.sum(1, keepdim=True)
    g_o, g_dpos = ray_pos_grad[:, 0:3], ray_pos_grad[:, 3:6]
    if ray_valid is not None:
        keep = ray_valid[:, None].to(_f32)
        g_sh, g_dn, g_o, g_dpos = g_sh * keep, g_dn * keep, g_o * keep, g_dpos * keep
    torch.autograd.backward([sh, dn], [g_sh, g_dn])
    return g_o.contiguous(), d.grad + g_dpos


def ray_gradients(rays_o, rays_d, z_vals, features, resolutions, blob, min_bbox, bbox_size, contract_mode, dfeat,
                  g_dnorm, g_rowsum, ray_valid=None):
    """dL/d(rays_o), dL/d(rays_d) of a fused render (for pose refinement: tile.py trains se3_refine through
    the rays).  Three paths: the sample positions (hash-encoder point gradient kernel on dfeat, then the
    contraction's Jacobian by torch autograd), |d| through delta = dist*|d|, and SH(d/|d|) of the decoder."""
    from . import tile_model
    B, S = z_vals.shape
    dev = z_vals.device
    mn = torch.as_tensor(min_bbox, dtype=_f32, device=dev)
    sz = torch.as_tensor(bbox_size, dtype=_f32, device=dev)
    o = rays_o.detach().clone().requires_grad_(True)
    d = rays_d.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        x = ((o[:, None, :] + z_vals[..., None] * d[:, None, :]).reshape(-1, 3) - mn) / sz * 4.0 - 2.0
        if contract_mode == BG:
            linf = torch.max(torch.abs(x), dim=-1, keepdim=True)[0]
            x = x * ((2 - 1.0 / linf) / linf)
        dn = d.norm(2, dim=-1, keepdim=True)
        sh = tile_model.sh3(d / (dn + 1e-8))
    gp = torch.zeros(B * S, 3, dtype=_f32, device=dev)
    xd = x.detach().contiguous()
    check(lib().scanerf_embedding_bg_point_grad(dev_ptr(xd, _f32, "points"), dev_ptr(dfeat, _f32, "dfeat"),
                                                dev_ptr(gp, _f32, "grad_points"), dev_ptr(features, _f32, "features"),
                                                dev_ptr(resolutions, torch.int32, "resolutions"), ctypes.c_int(B * S),
                                                ctypes.c_int(16), ctypes.c_int(features.shape[1]), stream()),
          "embedding_bg_point_grad")
    w_sh = blob[6503 + 64 + 32 * 64: 6503 + 64 + 48 * 64].reshape(16, 64)  # Directional_MLP.mlp.0 weight^T rows 32..47
    g_sh = g_rowsum.sum(1) @ w_sh.t()
    g_dn = g_dnorm.sum(1, keepdim=True)
    if ray_valid is not None:
        keep = ray_valid[:, None].to(_f32)
        gp = gp * keep.repeat_interleave(S, dim=0)
        g_sh, g_dn = g_sh * keep, g_dn * keep
    torch.autograd.backward([x, sh, dn], [gp, g_sh, g_dn])
    return o.grad, d.grad


# ---------------------------------------------------------------------------------------------------------------------
# Autograd boundary of the fused path: HashGrid.render_batch_rays (hashgrid/__init__.py:512-596) as ONE differentiable op.
# A caller that keeps the reference's loss code -- depth / smoothness / ADMM penalty terms on the per-ray outputs, summed
# and sent through loss.backward() (tile.py:954-1011, criterions.py:122-196) -- reaches the fused kernels through it.
FORWARD_PLAN_IN_AUTOGRAD = True   # (False: FusedRenderRays plans in its backward, as rounds 1-5 did: A/B and tests)


class FusedRenderRays(torch.autograd.Function):
    """out_ray [B,16], weights [B,S] = render(rays, samples; table, decoder blob).

    forward  = scanerf_render_forward_packed (+ x-stash, + position-Jacobian stash when the rays need gradients);
    backward = scanerf_render_backward on an ARBITRARY dL/d(out_ray) [B,16], returning the gradient of the hash table (dense
               [16,T,2]: fused record emission + scanerf_render_scatter_accumulate up to 2^21 entries per level, dfeat + the
               binned scatter above), of the decoder blob [13994] and of rays_o / rays_d.
    Not differentiated: z_vals / dists (the sampler runs under no_grad in the reference, hashgrid/__init__.py:278-285) and the
    `weights` output (returned for inspection: gradients flowing into it are ignored, as for l2_reg_specular's detached
    weights, hashgrid/__init__.py:593).  The out_ray column W_SPEC2 is differentiated with the weights detached (ibid.)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, z_vals, dists, features, blob, resolutions, weight_feature, min_bbox, bbox_size,
                contract_mode, infinity, ray_valid, skip_levels, want_weights):
        B, S = z_vals.shape
        dev = z_vals.device
        rays_o, rays_d = rays_o.contiguous(), rays_d.contiguous()
        z_vals, dists = z_vals.contiguous(), dists.contiguous()
        packed = PackedDecoder(dev).pack(blob.detach().contiguous(), weight_feature.reshape(-1).contiguous(), skip_levels)
        need_rays = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        need_grad = need_rays or ctx.needs_input_grad[4] or ctx.needs_input_grad[5]
        table = features.detach()
        tile_T = torch.empty((B, tile_T_columns(S)), dtype=_f32, device=dev) if need_grad else None
        xstash = torch.empty((B * S, 32), dtype=_f32, device=dev) if need_grad else None
        jstash = None
        if need_rays and table.dtype == torch.float32 and ARITH in _capi.T16_FAMILY:
            jstash = torch.empty(jstash_shape(B, S), dtype=JSTASH_DTYPE, device=dev)
        if ray_valid is not None and ray_valid.dtype not in (torch.bool, torch.uint8):
            raise RuntimeError("scanerf: ray_valid must be bool / uint8")
        # The forward launch counts the backward's record ranges (its hash indices are the plan's: no separate count launch, 0.27 ms at
        # configs[1]) where the table gradient will go through the fused records -- into a workspace this call OWNS until its
        # backward has run (other render calls may come between the two and use the per-stream one).
        T = table.shape[1]
        own_ws = None
        if (FORWARD_PLAN_IN_AUTOGRAD and ctx.needs_input_grad[4] and T <= (1 << 21) and scatter_supported(B, S, T)
                and forward_plan_supported(B, S, T) and backward_arith(True, need_rays) in _capi.T16_FAMILY):
            own_ws = torch.empty(lib().scanerf_render_scatter_workspace_bytes(ctypes.c_int(B), ctypes.c_int(S), ctypes.c_int(T)),
                                 dtype=torch.uint8, device=dev)
        r = render_forward(rays_o, rays_d, z_vals, dists, table, resolutions, packed, min_bbox, bbox_size, contract_mode,
                           infinity, ray_valid=ray_valid, want_weights=want_weights, tile_T=tile_T, xstash=xstash, jstash=jstash,
                           plan=own_ws is not None, plan_workspace=own_ws)
        ctx.plan_ws, ctx.plan_arith = own_ws, (backward_arith(True, need_rays) if own_ws is not None else None)
        out, weights = r[0], r[1]
        ctx.save_for_backward(rays_o, rays_d, z_vals, dists, table, resolutions, weight_feature.reshape(-1).contiguous(), out,
                              tile_T, xstash, jstash, ray_valid, blob.detach())
        ctx.packed, ctx.geom = packed, (list(min_bbox), list(bbox_size), int(contract_mode), bool(infinity))
        if weights is None:
            weights = out.new_empty(0)
        ctx.mark_non_differentiable(weights)
        return out, weights

    @staticmethod
    def backward(ctx, grad_out, _grad_weights):
        (rays_o, rays_d, z_vals, dists, table, resolutions, wf, out, tile_T, xstash, jstash, ray_valid, blob) = ctx.saved_tensors
        B, S = z_vals.shape
        dev = z_vals.device
        T = table.shape[1]
        box = ctx.geom
        need_rays = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        need_table = ctx.needs_input_grad[4]
        grad_out = grad_out.contiguous().to(_f32)
        arith = backward_arith(True, need_rays)
        fused = need_table and T <= (1 << 21) and scatter_supported(B, S, T)
        gtab = torch.zeros((16, T, 2), dtype=_f32, device=dev) if need_table else None
        gblob = torch.zeros(_capi.PARAMSIZE, dtype=_f32, device=dev)
        ws = None
        if fused and ctx.plan_ws is not None and ctx.plan_arith == arith:
            ws = ctx.plan_ws   # (counted by the forward launch)
        elif fused:   # (count + scan here, on the per-stream workspace: plan, backward and accumulate run back to back)
            ws = scatter_plan(rays_o, rays_d, z_vals, resolutions, T, *box, ray_valid=ray_valid, arith=arith,
                              skip_levels=ctx.packed.skip_levels)
        ntile = (S + 31) // 32
        bufs = (torch.zeros(B, ntile, device=dev), torch.zeros(B, 2, 64, device=dev)) if need_rays else None
        rp = torch.zeros(B, 6, device=dev) if (need_rays and jstash is not None) else None
        want_dfeat = (need_table and not fused) or (need_rays and jstash is None)
        if not fused and not want_dfeat:
            want_dfeat = True   # (the kernel needs somewhere to put the feature gradients)
        dfeat, _ = render_backward(rays_o, rays_d, z_vals, dists, table, resolutions, ctx.packed, wf, *box, out, tile_T,
                                   grad_out, ray_valid=ray_valid, grad_blob=gblob, xstash=xstash, ray_grad_buffers=bufs,
                                   scatter=(ws, gtab) if fused else None, want_dfeat=want_dfeat, arith=arith, jstash=jstash,
                                   ray_pos_grad=rp)
        g_o = g_d = None
        if need_rays and jstash is not None:
            g_o, g_d = ray_gradients_fused(rays_o, rays_d, blob, rp, bufs[0], bufs[1], ray_valid=ray_valid)
        elif need_rays:
            g_o, g_d = ray_gradients(rays_o, rays_d, z_vals, table.to(_f32), resolutions, blob, box[0], box[1], box[2], dfeat,
                                     bufs[0], bufs[1], ray_valid=ray_valid)
        if fused:
            scatter_accumulate(ws, gtab, B, S)
        elif need_table:
            mn = torch.as_tensor(box[0], dtype=_f32, device=dev)
            sz = torch.as_tensor(box[1], dtype=_f32, device=dev)
            pts = ((rays_o[:, None, :] + z_vals[:, :, None] * rays_d[:, None, :]).reshape(-1, 3) - mn) / sz * 4.0 - 2.0
            if box[2] == BG:
                linf = pts.abs().amax(-1, keepdim=True)
                pts = pts * ((2.0 - 1.0 / linf) / linf)
            scatter_table_grad(pts.contiguous(), dfeat, gtab, resolutions)
        if gtab is not None and table.dtype != _f32:
            gtab = gtab.to(table.dtype)
        return (g_o if ctx.needs_input_grad[0] else None, g_d if ctx.needs_input_grad[1] else None, None, None, gtab,
                gblob if ctx.needs_input_grad[5] else None, None, None, None, None, None, None, None, None, None)


def fused_render_rays(rays_o, rays_d, z_vals, dists, features, blob, resolutions, weight_feature, min_bbox, bbox_size,
                      contract_mode, infinity, ray_valid=None, skip_levels=0, want_weights=True):
    """FusedRenderRays.apply with keyword defaults -> (out_ray [B,16], weights [B,S])."""
    return FusedRenderRays.apply(rays_o, rays_d, z_vals, dists, features, blob, resolutions, weight_feature, min_bbox, bbox_size,
                                 contract_mode, infinity, ray_valid, skip_levels, want_weights)


class CompositeRays(torch.autograd.Function):
    """Alpha compositing of per-sample decoder outputs along rays as ONE op each way (csrc/composite.hip): what
    HashGrid.cal_integrate_weight + accumulate x 4 + the l2_reg_specular sum do in ~35 torch kernels (hashgrid/__init__.py:344-366,
    :564-574, :591-594).  sigma [B,S,1] or [B,S]; diffuse / specular / tint [B,S,3]; z_vals / dists [B,S]; rays_d [B,3] ->
    out_ray [B,16] (columns as the fused forward's: RGB, DEPTH, T_LEFT, DIFFUSE, SPECULAR, TINT, W_SPEC2), weights [B,S].
    Gradients flow to sigma, diffuse, specular, tint and (through delta = dists |d|) rays_d; z_vals / dists are sampler outputs."""

    @staticmethod
    def forward(ctx, sigma, diffuse, specular, tint, z_vals, dists, rays_d, infinity):
        B, S = z_vals.shape
        dev = z_vals.device
        sg = sigma.reshape(B * S).contiguous()
        cd, cs, tn = (t.reshape(B * S, 3).contiguous() for t in (diffuse, specular, tint))
        z, dd, rd = z_vals.contiguous(), dists.contiguous(), rays_d.detach().contiguous()
        out = torch.empty((B, _capi.RAY_OUT), dtype=_f32, device=dev)
        w = torch.empty((B, S), dtype=_f32, device=dev)
        check(lib().scanerf_composite_forward(
            dev_ptr(sg, _f32, "sigma"), dev_ptr(cd, _f32, "diffuse"), dev_ptr(cs, _f32, "specular"), dev_ptr(tn, _f32, "tint"),
            dev_ptr(z, _f32, "z_vals"), dev_ptr(dd, _f32, "dists"), dev_ptr(rd, _f32, "rays_d"), dev_ptr(out, _f32, "out_ray"),
            dev_ptr(w, _f32, "weights"), ctypes.c_int(B), ctypes.c_int(S), ctypes.c_int(1 if infinity else 0), stream()), "composite_forward")
        ctx.save_for_backward(sg, cd, cs, tn, z, dd, rd, out)
        ctx.infinity, ctx.shapes = bool(infinity), (sigma.shape, diffuse.shape)
        return out, w

    @staticmethod
    def backward(ctx, g_out, g_w):
        sg, cd, cs, tn, z, dd, rd, out = ctx.saved_tensors
        B, S = z.shape
        dev = z.device
        g_sigma = torch.empty_like(sg)
        g_cd, g_cs, g_tn = torch.empty_like(cd), torch.empty_like(cs), torch.empty_like(tn)
        g_dn = torch.empty(B, dtype=_f32, device=dev)
        go = g_out.contiguous().to(_f32) if g_out is not None else torch.zeros_like(out)
        gw = g_w.contiguous().to(_f32) if g_w is not None else None
        check(lib().scanerf_composite_backward(
            dev_ptr(sg, _f32, "sigma"), dev_ptr(cd, _f32, "diffuse"), dev_ptr(cs, _f32, "specular"), dev_ptr(tn, _f32, "tint"),
            dev_ptr(z, _f32, "z_vals"), dev_ptr(dd, _f32, "dists"), dev_ptr(rd, _f32, "rays_d"), dev_ptr(out, _f32, "out_ray"),
            dev_ptr(go, _f32, "grad_out"), dev_ptr(gw, _f32, "grad_weights", allow_none=True), dev_ptr(g_sigma, _f32, "g_sigma"),
            dev_ptr(g_cd, _f32, "g_diffuse"), dev_ptr(g_cs, _f32, "g_specular"), dev_ptr(g_tn, _f32, "g_tint"),
            dev_ptr(g_dn, _f32, "g_dnorm"), ctypes.c_int(B), ctypes.c_int(S), ctypes.c_int(1 if ctx.infinity else 0), stream()),
            "composite_backward")
        g_rd = g_dn[:, None] * rd / rd.norm(dim=-1, keepdim=True) if ctx.needs_input_grad[6] else None
        sshape, cshape = ctx.shapes
        return g_sigma.reshape(sshape), g_cd.reshape(cshape), g_cs.reshape(cshape), g_tn.reshape(cshape), None, None, g_rd, None


def composite_rays(sigma, diffuse, specular, tint, z_vals, dists, rays_d, infinity=False):
    """-> (out_ray [B,16], weights [B,S]); see CompositeRays."""
    return CompositeRays.apply(sigma, diffuse, specular, tint, z_vals, dists, rays_d, infinity)


def render_batch_rays_dict(out_ray, weights, train):
    """The dictionary HashGrid.render_batch_rays returns (hashgrid/__init__.py:564-596) from the fused op's outputs:
    diffuse / tint / specular / rgb [B,3], depth [B,1], T_left [B], weights [B,S,1], and in TRAIN mode l2_reg_specular =
    mean over rays and channels of sum_s w_s c_s^2 (:593: torch.mean over [B,3]; out_ray holds the channel sum)."""
    out = {"diffuse": out_ray[:, DIFFUSE], "tint": out_ray[:, TINT], "specular": out_ray[:, SPECULAR],
           "rgb": out_ray[:, RGB], "depth": out_ray[:, DEPTH, None], "T_left": out_ray[:, T_LEFT],
           "weights": weights[..., None] if weights is not None and weights.numel() else None}
    if train:
        out["l2_reg_specular"] = out_ray[:, W_SPEC2].sum() / (3.0 * out_ray.shape[0])
    return out
