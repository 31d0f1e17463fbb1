"""Autograd wrapper + module around the contracted-space hash encoder
(counterpart of the reference's hashgrid/PyHashGridBG.py:9-88, same constructor arguments)."""
import torch
import torch.nn as nn

from .lib.HASHGRID import embedding_bg_backward_cuda, embedding_bg_forward_cuda


TIMER = None  # optional tile_model.KernelTimer: per-kernel HIP-event timing for bench.py


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _sec(name, alg_bytes):
    return TIMER.section(name, alg_bytes) if TIMER is not None else _Null()


class _EncodeBG(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, features, resolution):
        N, L = points.shape[0], features.shape[0]
        out = points.new_zeros((N, L, 2))
        with _sec("embedding_bg_forward", N * L * 8 * features.shape[2] * features.element_size()):
            embedding_bg_forward_cuda(points, out, features, resolution)
        ctx.save_for_backward(points, features, resolution)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        points, features, resolution = ctx.saved_tensors
        N, L = points.shape[0], features.shape[0]
        g_pts = torch.zeros_like(points) if ctx.needs_input_grad[0] else None
        g_feat = torch.zeros_like(features)
        with _sec("embedding_bg_backward", N * L * (8 + 64 + 16 * 8)):  # SURVEY.md 8(d) bytes_bwd
            embedding_bg_backward_cuda(points, grad_out.contiguous(), g_pts, g_feat, features, resolution)
        return g_pts, g_feat, None


def HashEmbeddingBG(points, features, resolution):
    return _EncodeBG.apply(points, features, resolution)


def level_resolutions(base_resolution, finest_resolution, n_levels):
    """int(base * b**l) per axis with b = exp((ln fin - ln base)/(L-1)), in torch fp32 as the
    reference computes it (PyHashGridBG.py:53-62)."""
    base = torch.as_tensor(base_resolution).float()
    fin = torch.as_tensor(finest_resolution).float()
    b = torch.exp((torch.log(fin) - torch.log(base)) / (n_levels - 1))
    return torch.stack([(base * b ** i).int() for i in range(n_levels)], 0)


class PyHashGridBG(nn.Module):
    def __init__(self, device, bbox_corner, bbox_size, n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                 base_resolution=16, finest_resolution=512, init_mode="xavier"):
        super().__init__()
        if n_features_per_level != 2:
            raise ValueError("only 2 features per level are supported (as in the reference)")
        self.device = device
        self.bbox_corner, self.bbox_size = bbox_corner, bbox_size
        self.n_levels, self.n_features_per_level = n_levels, n_features_per_level
        self.log2_hashmap_size = log2_hashmap_size
        self.out_dim = n_levels * n_features_per_level
        self.resolution = level_resolutions(base_resolution, finest_resolution, n_levels).to(device)
        table = torch.zeros(n_levels, 2 ** log2_hashmap_size, n_features_per_level, dtype=torch.float32, device=device)
        if init_mode == "kaiming":
            nn.init.kaiming_normal_(table)
        elif init_mode == "xavier":
            nn.init.xavier_normal_(table)
        elif init_mode == "uniform":
            nn.init.uniform_(table, -1e-4, 1e-4)
        self.features = nn.Parameter(table)

    def forward(self, x):
        lead = x.shape[:-1]
        f = HashEmbeddingBG(x.reshape(-1, 3).contiguous(), self.features, self.resolution)
        return f.reshape(*lead, self.out_dim)
