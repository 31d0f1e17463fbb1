"""World-space box variant of the encoder (counterpart of hashgrid/PyHashGrid.py:9-86)."""
import torch
import torch.nn as nn

from .lib.HASHGRID import embedding_backward_cuda, embedding_forward_cuda
from .PyHashGridBG import level_resolutions


class _EncodeBox(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, features, corner, size, resolution):
        out = points.new_zeros((points.shape[0], features.shape[0], 2))
        embedding_forward_cuda(points, out, features, corner, size, resolution)
        ctx.save_for_backward(points, features, corner, size, resolution)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        points, features, corner, size, resolution = ctx.saved_tensors
        g_pts = torch.zeros_like(points)
        g_feat = torch.zeros_like(features)
        embedding_backward_cuda(points, grad_out.contiguous(), g_pts, g_feat, features, corner, size, resolution)
        return g_pts, g_feat, None, None, None


def HashEmbedding(points, features, corner, size, resolution):
    return _EncodeBox.apply(points, features, corner, size, resolution)


class PyHashGrid(nn.Module):
    def __init__(self, device, bbox_corner, bbox_size, n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                 base_resolution=16, finest_resolution=512, init_mode="xavier"):
        super().__init__()
        self.device = device
        self.bbox_corner = torch.as_tensor(bbox_corner, dtype=torch.float32, device=device).reshape(3).contiguous()
        size = torch.as_tensor(bbox_size, dtype=torch.float32, device=device)
        self.bbox_size = (size.repeat(3) if size.numel() == 1 else size.reshape(3)).contiguous()
        self.n_levels, self.out_dim = n_levels, n_levels * n_features_per_level
        base = torch.as_tensor(base_resolution).float().expand(3) if torch.as_tensor(base_resolution).numel() == 1 \
            else torch.as_tensor(base_resolution)
        fin = torch.as_tensor(finest_resolution).float().expand(3) if torch.as_tensor(finest_resolution).numel() == 1 \
            else torch.as_tensor(finest_resolution)
        self.resolution = level_resolutions(base, fin, n_levels).to(device).contiguous()
        table = torch.zeros(n_levels, 2 ** log2_hashmap_size, n_features_per_level, dtype=torch.float32, device=device)
        if init_mode == "xavier":
            nn.init.xavier_normal_(table)
        elif init_mode == "kaiming":
            nn.init.kaiming_normal_(table)
        elif init_mode == "uniform":
            nn.init.uniform_(table, -1e-4, 1e-4)
        self.features = nn.Parameter(table)

    def forward(self, x):
        lead = x.shape[:-1]
        f = HashEmbedding(x.reshape(-1, 3).contiguous(), self.features, self.bbox_corner, self.bbox_size, self.resolution)
        return f.reshape(*lead, self.out_dim)
