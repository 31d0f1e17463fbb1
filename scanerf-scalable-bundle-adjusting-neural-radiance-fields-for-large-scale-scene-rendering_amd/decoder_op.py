"""The decoder as one differentiable op: network.ShallowMLP.forward (network.py:172-190) through csrc/decoder.hip.

    sigma [N,1], diffuse [N,3], specular [N,3], tint [N,3] = decoder_apply(x [N,35], blob [13994], weight_feature [32])

forward = scanerf_decoder_forward, backward = scanerf_decoder_backward (one launch each, split-f16 matrix cores,
f32-equivalent); x and its gradient are addressed in place with row stride 35 (features at column 0, direction at 32).
There is no CPU path: CPU tensors raise (as every op of this package).
"""
import ctypes

import torch

from . import _capi
from ._capi import check, dev_ptr, lib, stream
from .render import PackedDecoder

_f32 = torch.float32


def _off(t, floats):
    return ctypes.c_void_p(t.data_ptr() + 4 * floats)


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, blob, wf):
        if x.dim() != 2 or x.shape[1] != 35:
            raise RuntimeError(f"scanerf: decoder input must be [N, 32 + 3], got {tuple(x.shape)}")
        x = x.contiguous()
        dev_ptr(x, _f32, "decoder input")
        N = x.shape[0]
        wf = wf.detach().reshape(-1).to(_f32).contiguous()
        packed = PackedDecoder(x.device).pack(blob.detach().contiguous(), wf)
        sigma = torch.empty((N, 1), dtype=_f32, device=x.device)
        dif, spec, tint = (torch.empty((N, 3), dtype=_f32, device=x.device) for _ in range(3))
        check(lib().scanerf_decoder_forward(
            dev_ptr(x, _f32, "x"), ctypes.c_int(35), _off(x, 32), ctypes.c_int(35), dev_ptr(packed.workspace, _f32, "workspace"),
            dev_ptr(sigma, _f32, "sigma"), dev_ptr(dif, _f32, "diffuse"), dev_ptr(spec, _f32, "specular"),
            dev_ptr(tint, _f32, "tint"), ctypes.c_longlong(N), stream()), "decoder_forward")
        ctx.save_for_backward(x, wf)
        ctx.packed = packed
        return sigma, dif, spec, tint

    @staticmethod
    def backward(ctx, g_sigma, g_dif, g_spec, g_tint):
        x, wf = ctx.saved_tensors
        N = x.shape[0]
        dev = x.device
        gx = torch.empty_like(x)
        gblob = torch.zeros(_capi.PARAMSIZE, dtype=_f32, device=dev)
        nblk = lib().scanerf_decoder_backward_grid(ctypes.c_longlong(N))
        dw_partial = torch.empty((nblk, _capi.PARAMSIZE), dtype=_f32, device=dev)
        keep = [None if t is None else t.contiguous().to(_f32) for t in (g_sigma, g_dif, g_spec, g_tint)]
        check(lib().scanerf_decoder_backward(
            dev_ptr(x, _f32, "x"), ctypes.c_int(35), _off(x, 32), ctypes.c_int(35), dev_ptr(ctx.packed.workspace, _f32, "workspace"),
            dev_ptr(wf, _f32, "weight_feature"),
            *(dev_ptr(t, _f32, n, allow_none=True) for t, n in zip(keep, ("g_sigma", "g_diffuse", "g_specular", "g_tint"))),
            dev_ptr(gx, _f32, "gx"), ctypes.c_int(35), _off(gx, 32), ctypes.c_int(35), dev_ptr(dw_partial, _f32, "dw_partial"),
            dev_ptr(gblob, _f32, "grad_blob"), ctypes.c_longlong(N), stream()), "decoder_backward")
        return gx, gblob, None


class _DecoderPartsFn(torch.autograd.Function):
    """The same op on the encoder's output and the view directions as TWO tensors (features [N,32], dirs [N,3]): no [N,35]
    concatenation (1.2 GB written per 65 536 x 128 batch on the op-by-op route, and its split in the backward)."""

    @staticmethod
    def forward(ctx, feats, dirs, blob, wf):
        if feats.dim() != 2 or feats.shape[1] != 32 or dirs.shape != (feats.shape[0], 3):
            raise RuntimeError(f"scanerf: decoder inputs must be [N,32] and [N,3], got {tuple(feats.shape)} and {tuple(dirs.shape)}")
        feats, dirs = feats.contiguous(), dirs.contiguous()
        N, dev = feats.shape[0], feats.device
        wf = wf.detach().reshape(-1).to(_f32).contiguous()
        packed = PackedDecoder(dev).pack(blob.detach().contiguous(), wf)
        sigma = torch.empty((N, 1), dtype=_f32, device=dev)
        dif, spec, tint = (torch.empty((N, 3), dtype=_f32, device=dev) for _ in range(3))
        check(lib().scanerf_decoder_forward(
            dev_ptr(feats, _f32, "features"), ctypes.c_int(32), dev_ptr(dirs, _f32, "dirs"), ctypes.c_int(3),
            dev_ptr(packed.workspace, _f32, "workspace"), dev_ptr(sigma, _f32, "sigma"), dev_ptr(dif, _f32, "diffuse"),
            dev_ptr(spec, _f32, "specular"), dev_ptr(tint, _f32, "tint"), ctypes.c_longlong(N), stream()), "decoder_forward")
        ctx.save_for_backward(feats, dirs, wf)
        ctx.packed = packed
        return sigma, dif, spec, tint

    @staticmethod
    def backward(ctx, g_sigma, g_dif, g_spec, g_tint):
        feats, dirs, wf = ctx.saved_tensors
        N, dev = feats.shape[0], feats.device
        # (directions that carry no gradient -- no pose refinement -- skip the harmonics' adjoint: d_dirs = NULL)
        gf, gd = torch.empty_like(feats), (torch.empty_like(dirs) if ctx.needs_input_grad[1] else None)
        gblob = torch.zeros(_capi.PARAMSIZE, dtype=_f32, device=dev)
        nblk = lib().scanerf_decoder_backward_grid(ctypes.c_longlong(N))
        dw_partial = torch.empty((nblk, _capi.PARAMSIZE), dtype=_f32, device=dev)
        keep = [None if t is None else t.contiguous().to(_f32) for t in (g_sigma, g_dif, g_spec, g_tint)]
        check(lib().scanerf_decoder_backward(
            dev_ptr(feats, _f32, "features"), ctypes.c_int(32), dev_ptr(dirs, _f32, "dirs"), ctypes.c_int(3),
            dev_ptr(ctx.packed.workspace, _f32, "workspace"), dev_ptr(wf, _f32, "weight_feature"),
            *(dev_ptr(t, _f32, n, allow_none=True) for t, n in zip(keep, ("g_sigma", "g_diffuse", "g_specular", "g_tint"))),
            dev_ptr(gf, _f32, "g_features"), ctypes.c_int(32), dev_ptr(gd, _f32, "g_dirs", allow_none=True), ctypes.c_int(3),
            dev_ptr(dw_partial, _f32, "dw_partial"), dev_ptr(gblob, _f32, "grad_blob"), ctypes.c_longlong(N), stream()), "decoder_backward")
        return gf, gd, gblob, None


def decoder_apply_parts(features, dirs, blob, weight_feature):
    """features [N,32] (the hash encoder's output rows), dirs [N,3] (per-sample view directions), blob [13994], weight_feature [32]
    -> sigma [N,1], diffuse / specular / tint [N,3]: decoder_apply without the concatenated input."""
    return _DecoderPartsFn.apply(features, dirs, blob, weight_feature)


def decoder_apply(x, blob, weight_feature):
    """x [N,35] (32 hash features + 3 view direction), blob [13994] (rendering.py:101-112 order), weight_feature [32]."""
    return _DecoderFn.apply(x, blob, weight_feature)
