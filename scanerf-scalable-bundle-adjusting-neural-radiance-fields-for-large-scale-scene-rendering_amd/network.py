"""Decoder parameters for the fused kernels.

The decoder is the reference's network.ShallowMLP (network.py:151-190).  Its parameters
travel as the reference's render-time blob (rendering.py:101-112, decoder.h:48-67):
for each layer in state-dict order, [bias(out), W^T flattened (in-major)] -- 13 994 floats.
"""
import math

import torch

PARAMSIZE = 13994
# (state-dict prefix, out, in) in state-dict order (network.py:155-163)
LAYERS = (("Spatial_MLP.mlp.0", 64, 32), ("Spatial_MLP.mlp.2", 64, 64), ("sigma_layer.mlp.0", 1, 32),
          ("diffuse_layer.mlp.0", 3, 32), ("tint_layer.mlp.0", 3, 32), ("Directional_MLP.mlp.0", 64, 48),
          ("Directional_MLP.mlp.2", 64, 64), ("Directional_MLP.mlp.4", 3, 64))


def layers(in_channel=32):
    """LAYERS with `in_channel` encoder features (2 per level): 32 in the reference; BASELINE configs[0] uses 8 levels."""
    return tuple((n, o, in_channel if n == "Spatial_MLP.mlp.0" else i) for n, o, i in LAYERS)


def blob_from_state_dict(sd, in_channel=32):
    parts = []
    for name, o, i in layers(in_channel):
        w, b = sd[name + ".weight"], sd[name + ".bias"]
        assert tuple(w.shape) == (o, i), (name, tuple(w.shape))
        parts += [b.reshape(-1), w.t().reshape(-1)]
    blob = torch.cat(parts).float().contiguous()
    assert blob.numel() == PARAMSIZE - 64 * (32 - in_channel)
    return blob


def state_dict_from_blob(blob):
    sd, k = {}, 0
    for name, o, i in LAYERS:
        sd[name + ".bias"] = blob[k:k + o]
        k += o
        sd[name + ".weight"] = blob[k:k + i * o].reshape(i, o).t()
        k += i * o
    return sd


def xavier_blob(seed=0, device="cpu", bias_scale=0.0, in_channel=32):
    """Random decoder: Xavier-normal weights, zero (or small) biases (network.py:202-205)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, o, i in layers(in_channel):
        sd[name + ".weight"] = torch.randn(o, i, generator=g) * math.sqrt(2.0 / (i + o))
        sd[name + ".bias"] = torch.randn(o, generator=g) * bias_scale
    return blob_from_state_dict(sd, in_channel).to(device)


def skip_levels(global_step):
    """Bit l set: weight_feature(global_step) is exactly zero on level l (levels >= alpha of the coarse-to-fine schedule)."""
    w = weight_feature(global_step)[::2]
    return int(sum(1 << l for l in range(16) if float(w[l]) == 0.0))


def weight_feature(global_step, device="cpu"):
    """Coarse-to-fine level mask of hashgrid/__init__.py:228-235, repeated per feature -> [32]."""
    alpha = max(min(global_step / 10000 * 8 + 8, 16), 0)
    k = torch.arange(16, dtype=torch.float32)
    w = (1 - torch.cos((alpha - k).clamp(0, 1) * math.pi)) / 2
    return w.repeat_interleave(2).to(device)
