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


# ---------------------------------------------------------------------------------------------------------------------
# Drop-in for the reference's decoder module (network.py:127-190): same class names, same module tree, hence the same
# state_dict keys (Spatial_MLP.mlp.{0,2}, sigma_layer.mlp.0, diffuse_layer.mlp.0, tint_layer.mlp.0, Directional_MLP.mlp.{0,2,4})
# -- a checkpoint of the reference loads with load_state_dict, optimiser parameter groups built by name keep working -- and
# the same forward contract (x [..., 32 + 3], weight_feature broadcastable to [..., 32] -> dict of sigma / diffuse / specular /
# tint).  The arithmetic runs in ONE HIP launch each way (csrc/decoder.hip: scanerf_decoder_forward / _backward, split-f16
# matrix cores, f32-equivalent) instead of torch's ~40 kernels over [N,64] intermediates.
import torch.nn as nn  # noqa: E402


class Gaussian_Act(nn.Module):
    """exp(-x^2 / (2 sigma^2)) (network.py:79-84); kept as a module so that the Sequential indices match the reference's."""

    def __init__(self, sigma=0.1):
        super().__init__()
        self.item = 1.0 / (-2 * (sigma ** 2))

    def forward(self, x):
        return torch.exp((x ** 2) * self.item)


class GeneralMLP(nn.Module):
    """network.py:127-148: Linear / activation stack under `.mlp` (Sequential)."""

    def __init__(self, num_in, num_out, activation, hiden_depth=4, hiden_width=64, output_act=False):
        super().__init__()
        assert hiden_depth >= 1
        if hiden_depth == 1:
            mods = [nn.Linear(num_in, num_out)]
        else:
            mods = [nn.Linear(num_in, hiden_width), activation]
            for _ in range(hiden_depth - 2):
                mods += [nn.Linear(hiden_width, hiden_width), activation]
            mods.append(nn.Linear(hiden_width, num_out))
        if output_act:
            mods.append(activation)
        self.mlp = nn.Sequential(*mods)

    def forward(self, x):
        return self.mlp(x)


class ShallowMLP(nn.Module):
    """network.py:151-190.  `forward` = the HIP decoder op for CUDA inputs with 32 feature channels (use_hip, default);
    `forward_torch` = the same computation through the module tree in torch (the reference's own statement of it; what the
    tests compare the op with, and what runs for other channel counts, e.g. BASELINE configs[0]'s 8 levels)."""

    def __init__(self, in_channel=32):
        super().__init__()
        self.in_channel = in_channel
        self.Spatial_MLP = GeneralMLP(in_channel, 64, Gaussian_Act(0.1), 2, 64)
        self.sigma_layer = GeneralMLP(32, 1, nn.Softplus(), 1, None, True)
        self.diffuse_layer = GeneralMLP(32, 3, nn.Sigmoid(), 1, None, True)
        self.tint_layer = GeneralMLP(32, 3, nn.Sigmoid(), 1, None, True)
        self.Directional_MLP = GeneralMLP(32 + 16, 3, Gaussian_Act(0.1), 3, 64)
        self.color_act = nn.Sigmoid()
        self.use_hip = True

    # -- the render-time blob (rendering.py:101-112) as a differentiable function of the named parameters
    def blob(self):
        sd = dict(self.named_parameters())
        parts = []
        for name, o, i in layers(self.in_channel):
            parts += [sd[name + ".bias"].reshape(-1), sd[name + ".weight"].t().reshape(-1)]
        return torch.cat(parts)

    def load_blob(self, blob):
        with torch.no_grad():
            sd, k = dict(self.named_parameters()), 0
            for name, o, i in layers(self.in_channel):
                sd[name + ".bias"].copy_(blob[k:k + o])
                k += o
                sd[name + ".weight"].copy_(blob[k:k + i * o].reshape(i, o).t())
                k += i * o
        return self

    def inference_sigma(self, x):
        H = self.Spatial_MLP(x)
        return self.sigma_layer(H[..., :32])

    def forward_torch(self, x, **kwargs):
        from .tile_model import sh3
        features, viewdirs = x[..., :-3], x[..., -3:]
        viewdirs = viewdirs / (viewdirs.norm(2, dim=-1, keepdim=True) + 1e-8)
        H = self.Spatial_MLP(features * kwargs["weight_feature"])
        sigma = self.sigma_layer(H[..., :32])
        tint = self.tint_layer(H[..., :32])
        c_d = self.diffuse_layer(H[..., :32])
        c_s = self.color_act(self.Directional_MLP(torch.cat([H[..., 32:], sh3(viewdirs)], -1)))
        return {"diffuse": c_d, "specular": c_s, "sigma": sigma, "tint": tint}

    def forward(self, x, **kwargs):
        wf = kwargs["weight_feature"]
        if not (self.use_hip and x.is_cuda and self.in_channel == 32 and x.shape[-1] == 35 and wf.numel() == 32):
            return self.forward_torch(x, **kwargs)   # (per-sample masks, other channel counts: the torch graph)
        from . import decoder_op
        lead = x.shape[:-1]
        sigma, dif, spec, tint = decoder_op.decoder_apply(x.reshape(-1, 35), self.blob(), wf.reshape(-1))
        return {"diffuse": dif.reshape(*lead, 3), "specular": spec.reshape(*lead, 3), "sigma": sigma.reshape(*lead, 1),
                "tint": tint.reshape(*lead, 3)}


def _shallow_forward_parts(self, features, viewdirs, **kwargs):
    """forward() on the two halves of the reference's concatenated input (hashgrid/__init__.py:547: cat([features, rays_d]))
    as separate tensors -- features [..., 32], viewdirs [..., 3] (broadcastable views are fine) -- so that a caller which has
    them apart never builds the [..., 35] tensor.  Same dictionary."""
    wf = kwargs["weight_feature"]
    if not (self.use_hip and features.is_cuda and self.in_channel == 32 and features.shape[-1] == 32 and wf.numel() == 32):
        return self.forward_torch(torch.cat([features, viewdirs.expand(*features.shape[:-1], 3)], -1), **kwargs)
    from . import decoder_op
    lead = features.shape[:-1]
    sigma, dif, spec, tint = decoder_op.decoder_apply_parts(features.reshape(-1, 32), viewdirs.expand(*lead, 3).reshape(-1, 3),
                                                            self.blob(), wf.reshape(-1))
    return {"diffuse": dif.reshape(*lead, 3), "specular": spec.reshape(*lead, 3), "sigma": sigma.reshape(*lead, 1),
            "tint": tint.reshape(*lead, 3)}


ShallowMLP.forward_parts = _shallow_forward_parts


def init_model(model, mode="default"):
    """network.py:196-227 (the modes the reference's callers use)."""
    def xavier_init(layer):
        if isinstance(layer, nn.Linear):
            nn.init.xavier_normal_(layer.weight)
            layer.bias.data.fill_(0.0)

    def kaiming_init(layer):
        if isinstance(layer, nn.Linear):
            nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
            layer.bias.data.fill_(0.0)

    def zeros_init(layer):
        if isinstance(layer, nn.Linear):
            nn.init.zeros_(layer.weight)
            layer.bias.data.fill_(0.0)
    assert mode in ("xavier", "kaiming", "zeros", "default")
    if mode != "default":
        model.apply({"xavier": xavier_init, "kaiming": kaiming_init, "zeros": zeros_init}[mode])
    return model
