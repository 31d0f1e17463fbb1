"""Drop-in for the reference's pybind module HASHGRID (hashgrid/binding.cpp:9-44), training-path ops.

Same names / positional order / in-place outputs as hashgrid/include/hashgrid.h:19-51.
The render-time ops of the module (rendering.h) are listed in DESIGN.md as the next rows.
"""
import ctypes
import os

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


def embedding_bg_backward_cuda(points, grad_in, grad_points, grad_features, features, resolutions):
    """hashgrid.h:45-51: accumulates into grad_points [N,3] and grad_features [L,T,2].
    The table gradient goes through the atomic-free binned scatter (csrc/scatter.hip) when the
    shape allows; SCANERF_SCATTER=atomics forces the reference-style atomic kernel."""
    N, (L, T) = points.shape[0], features.shape[:2]
    need = 0
    if grad_features is not None and N >= 4096 and os.environ.get("SCANERF_SCATTER", "binned") != "atomics":
        need = lib().scanerf_embedding_bwd_workspace_bytes(ctypes.c_int(N), ctypes.c_int(L), ctypes.c_int(T))
    if need:
        ws = workspace(points.device, need)
        check(lib().scanerf_embedding_bg_backward_binned(
            dev_ptr(points, _f32, "points"), dev_ptr(grad_in, _f32, "grad_in"),
            dev_ptr(grad_features, _f32, "grad_features"), _res(resolutions), ctypes.c_int(N), ctypes.c_int(L),
            ctypes.c_int(T), ctypes.c_int(0), ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), stream()),
            "embedding_bg_backward_cuda(binned)")
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
