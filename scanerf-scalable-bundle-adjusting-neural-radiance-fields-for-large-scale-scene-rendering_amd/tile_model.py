"""Per-tile model and training step: the build's counterpart of the slice of tile.py /
hashgrid/__init__.py that turns the hot-path kernels into a training iteration
(tile.py:639-692 render_rays, :880-1015 train_one_step; hashgrid/__init__.py:413-596).

Two execution paths over the same parameters:

  * "ops":   the reference's own structure -- HIP sampler + HIP hash encoder behind the
             binding-surface names, decoder and compositing in torch (autograd), dense torch
             gradient of the table, fused sparse Adam kernel on the table;
  * "fused": one HIP launch for render forward and one for backward (render.py).

Scope is the foreground branch on synthetic rays; data loading, warp/mono losses, pose
refinement and pruning schedules belong to the trainer.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import network, render
from .cuda import adam_step_cuda, sample_points_grid
from .hashgrid import HashEmbeddingBG, level_resolutions

_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435)


def sh3(v):
    """Real spherical harmonics up to degree 3 of unit vectors (16 values, network.py:38-77 ordering)."""
    x, y, z = v.unbind(-1)
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    return torch.stack([torch.full_like(x, 0.28209479177387814), _C1 * y, _C1 * z, _C1 * x, _C2[0] * xy, _C2[1] * yz,
                        _C2[2] * (2.0 * zz - xx - yy), _C2[3] * xz, _C2[4] * (xx - yy), _C3[0] * y * (3 * xx - yy),
                        _C3[1] * xy * z, _C3[2] * y * (4 * zz - xx - yy), _C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
                        _C3[4] * x * (4 * zz - xx - yy), _C3[5] * z * (xx - yy), _C3[6] * x * (xx - 3 * yy)], -1)


def composite_weights(sigma, dists, rays_d, infinity):
    """Alpha compositing weights of hashgrid/__init__.py:344-360 in torch: delta = dists * |d| (last sample 1e10 for a background
    ray), alpha = 1 - exp(-sigma delta), T_i = prod_{j<i} (1 - alpha_j + 1e-6), w = alpha T -> (w [B,S], T_left [B] = the
    transmittance BEFORE the last sample, the reference's quirk)."""
    delta = dists * rays_d.norm(dim=-1, keepdim=True)
    if infinity:
        delta = torch.cat([delta[:, :-1], torch.full_like(delta[:, :1], 1e10)], 1)
    alpha = 1.0 - torch.exp(-sigma * delta)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-6], 1), 1)[:, :-1]
    return alpha * T, T[:, -1]


@torch.no_grad()
def inverse_z_samples(rays_o, rays_d, box_center, box_half, num_sample, invalid_underground, floor_y=None):
    """Background samples in inverse depth beyond the HashGrid's 2x box (hashgrid/__init__.py:287-337): far = the ray's exit from
    the box (HIP ray_aabb_intersection; 0.1 for rays that miss it), z = 1 / ((1 - t) / (far + 1e-6) + t / 1e6) for t = linspace(0, 1, S),
    dists = differences with 1e-6 last; valid = not leaving through the floor y = floor_y (within 1e-4) when invalid_underground."""
    from .cuda import ray_aabb_intersection
    B, dev = rays_o.shape[0], rays_o.device
    bounds = torch.full((B, 2), -1.0, device=dev)
    ray_aabb_intersection(rays_o.contiguous(), rays_d.contiguous(), box_center.contiguous(), box_half.contiguous(), bounds)
    if invalid_underground:
        exit_y = rays_o[:, 1] + bounds[:, 1] * rays_d[:, 1]
        valid = ~(torch.abs(exit_y - floor_y) < 0.0001)
    else:
        valid = torch.ones(B, dtype=torch.bool, device=dev)
    far = torch.where(torch.any(bounds == -1, dim=-1, keepdim=True), torch.full_like(bounds[:, 1:], 0.1), bounds[:, 1:])
    t = torch.linspace(0.0, 1.0, steps=num_sample, device=dev)[None, :]
    z = (1.0 / (1.0 / (far + 1e-6) * (1.0 - t) + 1.0 / 1e6 * t)).contiguous()
    d = torch.cat([z[:, 1:] - z[:, :-1], torch.full((B, 1), 1e-6, device=dev)], -1).contiguous()
    return z, d, valid


class Decoder(nn.Module):
    """sigma / diffuse / tint / SH-conditioned specular decoder of the reference's ShallowMLP (network.py:151-190).
    The parameters ARE the render-time blob (rendering.py:101-112: per layer [bias, W^T]), one flat tensor: the fused
    kernels read it as it is, their gradient is its .grad, and the optimiser steps one tensor instead of sixteen
    (Adam is element-wise, so the update is the same).  Per-layer views carry the reference's parameter names."""

    def __init__(self, seed=0, in_channel=32):
        super().__init__()
        self.in_channel = in_channel  # 2 features per level: 32 in the reference, 16 for BASELINE configs[0] (ops path only)
        self.layers = network.layers(in_channel)
        self.params = nn.Parameter(network.xavier_blob(seed, in_channel=in_channel))
        self._views = {}

    def _view(self, name, kind):
        """weight [out,in] / bias [out] of a layer as a (differentiable) view of the blob"""
        k = 0
        for n, o, i in self.layers:
            if n == name:
                return self.params[k:k + o] if kind == "bias" else self.params[k + o:k + o + i * o].reshape(i, o).t()
            k += o + i * o
        raise KeyError(name)

    def __getattr__(self, attr):  # e.g. sigma_layer_mlp_0_bias -> view (the names tile.py's optimiser groups use)
        for n, _, _ in network.LAYERS:
            for kind in ("weight", "bias"):
                if attr == n.replace(".", "_") + "_" + kind:
                    return self._view(n, kind)
        return super().__getattr__(attr)

    def ref_state_dict(self):
        return {f"{n}.{k}": self._view(n, k) for n, _, _ in network.LAYERS for k in ("weight", "bias")}

    def load_ref_state_dict(self, sd):
        with torch.no_grad():
            self.params.copy_(network.blob_from_state_dict(sd, self.in_channel).to(self.params.device))

    def blob(self):
        return self.params

    def _lin(self, name, x):
        return F.linear(x, self._view(name, "weight"), self._view(name, "bias"))

    def forward(self, feats, dirs, weight_feature):
        act = lambda u: torch.exp(u * u * -50.0)
        v = dirs / (dirs.norm(2, dim=-1, keepdim=True) + 1e-8)
        H = self._lin("Spatial_MLP.mlp.2", act(self._lin("Spatial_MLP.mlp.0", feats * weight_feature)))
        sigma = F.softplus(self._lin("sigma_layer.mlp.0", H[..., :32]))
        tint = torch.sigmoid(self._lin("tint_layer.mlp.0", H[..., :32]))
        dif = torch.sigmoid(self._lin("diffuse_layer.mlp.0", H[..., :32]))
        h = act(self._lin("Directional_MLP.mlp.0", torch.cat([H[..., 32:], sh3(v)], -1)))
        h = act(self._lin("Directional_MLP.mlp.2", h))
        spec = torch.sigmoid(self._lin("Directional_MLP.mlp.4", h))
        return sigma, dif, spec, tint


class TileModel(nn.Module):
    """Geometry + parameters of one tile (hashgrid/__init__.py:33-92 with model_path == ""):
    `corner`/`size` describe the tile; the hash grid covers the 2x box around it."""

    def __init__(self, corner, size, device, log2_T=19, grid_resolution=(32, 2048), sampler_log2dim=4, seed=0,
                 table_dtype=torch.float32, n_levels=16, fp16_moments=False):
        """fp16_moments (opt-in; tables of >= 2^22 entries behind the t16s backward only): the table's Adam moments are kept in
        half precision and updated as adam_step_cuda_fp16 does (cuda/adam_kernel.cu:98-144) -- NOT what the reference's live code
        runs (torch.optim.Adam with fp32 state, tile.py:301), hence never the default."""
        super().__init__()
        self.fp16_moments = bool(fp16_moments)
        self.n_levels = n_levels  # the reference hard-codes 16; other counts run on the "ops" path only (configs[0]: 8)
        corner = torch.as_tensor(corner, dtype=torch.float32)
        size = torch.as_tensor(size, dtype=torch.float32)
        self.device = device
        self.bbox_center = corner + size / 2.0
        self.bbox_size = size * 2
        self.min_bbox = self.bbox_center - self.bbox_size / 2.0
        fin = (self.bbox_size / self.bbox_size.min() * grid_resolution[1]).int()
        base = (self.bbox_size / self.bbox_size.min() * grid_resolution[0]).int()
        self.resolution = level_resolutions(base, fin, n_levels).to(device).contiguous()
        g = torch.Generator().manual_seed(seed)
        T = 2 ** log2_T
        std = math.sqrt(2.0 / (T * 2 + n_levels * 2))  # xavier_normal_ on [L,T,2] (PyHashGridBG.py:72-73)
        self.features = nn.Parameter((torch.randn(n_levels, T, 2, generator=g) * std).to(device))
        self.table_dtype = table_dtype
        self.decoder = Decoder(seed, 2 * n_levels).to(device)
        self.log2dim = (sampler_log2dim - torch.log2(self.bbox_size.max() / self.bbox_size).int()).int().to(device)
        self.occupied_grid = torch.ones(tuple(int(2 ** k) for k in self.log2dim), dtype=torch.bool, device=device)
        self._occ_full = True  # set_occupancy() keeps it in step with the grid
        self.occ_corner = (self.min_bbox + self.bbox_size / 4.0).to(device).contiguous()
        self.occ_size = (self.bbox_size / 2.0).to(device).contiguous()
        self._min_dev = self.min_bbox.to(device)
        self._size_dev = self.bbox_size.to(device)
        self._center_dev = self.bbox_center.to(device).contiguous()
        self._half_dev = (self.bbox_size / 2.0).to(device).contiguous()
        self.packed = render.PackedDecoder(device)
        self._side_stream = torch.cuda.Stream(device=device) if str(device).startswith("cuda") else None
        # fused sparse Adam state for the table (cuda/adam_kernel.cu semantics)
        self.exp_avg = torch.zeros_like(self.features, dtype=torch.float16 if self.fp16_moments else torch.float32)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.adam_step = 0
        self._half_table = None       # f16 / bf16 gather copy of the table (configs[2]); kept in step by the Adam epilogue
        self._overflow_grad = None    # zero table for the fused scatter's workspace-overflow path (never filled per step)

    def gather_table(self):
        """The table the fused kernels gather from: the fp32 master, or its resident half-precision copy (half the gather
        bytes, fp32 accumulate).  The copy is converted ONCE; after that the accumulate's Adam epilogue rewrites exactly the
        entries it moves (train_step_fused), so there is no per-step full-table conversion."""
        if self.table_dtype == torch.float32:
            return self.features
        if self._half_table is None or self._half_table.dtype != self.table_dtype:
            self._half_table = self.features.detach().to(self.table_dtype).contiguous()
        return self._half_table

    def invalidate_gather_table(self):
        """Call after writing the fp32 master any other way than through train_step_fused (loading a checkpoint, ...)."""
        self._half_table = None

    def overflow_grad(self):
        if self._overflow_grad is None:
            self._overflow_grad = torch.zeros_like(self.features)
        return self._overflow_grad

    def set_occupancy(self, grid):
        """Replace the sampler's occupancy grid (pruning: hashgrid/__init__.py:138-225)."""
        if tuple(grid.shape) != tuple(self.occupied_grid.shape):
            raise ValueError(f"occupancy grid must be {tuple(self.occupied_grid.shape)}, got {tuple(grid.shape)}")
        self.occupied_grid = grid.to(self.device, torch.bool).contiguous()
        self._occ_full = bool(self.occupied_grid.all())

    def weight_feature(self, global_step):
        """Coarse-to-fine mask on the device, cached: it is constant once global_step >= 10 000."""
        key = min(int(global_step), 10000)
        if getattr(self, "_wf_key", None) != key:
            self._wf_key, self._wf = key, network.weight_feature(global_step, self.device)
        return self._wf

    # ---- sampling (no grad: hashgrid/__init__.py:278-285) -------------------------------
    @torch.no_grad()
    def sample(self, rays_o, rays_d, S):
        z = torch.full((rays_o.shape[0], S), -1.0, device=self.device)
        d = torch.full((rays_o.shape[0], S), -1.0, device=self.device)
        sample_points_grid(rays_o, rays_d, z, d, self.occ_corner, self.occ_size, self.occupied_grid, self.log2dim)
        return z, d

    # ---- "ops" path: binding-surface kernels + torch decoder / compositing --------------
    def render_fore_ops(self, rays_o, rays_d, S, global_step, train=True):
        z, dist = self.sample(rays_o, rays_d, S)
        valid = torch.all(z != -1, dim=-1)
        o, d, z, dist = rays_o[valid], rays_d[valid], z[valid], dist[valid]
        B = o.shape[0]
        pts = o[:, None, :] + z[..., None] * d[:, None, :]
        x = (pts.reshape(-1, 3) - self._min_dev) / self._size_dev * 4.0 - 2.0
        feats = HashEmbeddingBG(x.contiguous(), self.features, self.resolution).reshape(B, S, 2 * self.n_levels)
        wf = network.weight_feature(global_step, self.device)[:2 * self.n_levels]
        if self.n_levels == 16 and getattr(self, "hip_decoder", True):
            # the decoder as ONE HIP op each way (csrc/decoder.hip) on the encoder's rows and the per-sample directions -- the two
            # halves of the reference's concatenated input (hashgrid/__init__.py:547), never concatenated (round 6) -- and the
            # compositing as one op each way (csrc/composite.hip) instead of cal_integrate_weight + accumulate x 4 in torch
            from . import decoder_op
            sigma, dif, spec, tint = decoder_op.decoder_apply_parts(feats.reshape(-1, 32), d[:, None, :].expand(B, S, 3).reshape(-1, 3),
                                                                    self.decoder.blob(), wf)
            out_ray, w2 = render.composite_rays(sigma, dif, spec, tint, z, dist, d, False)
            out = {"valid": valid, "depth": out_ray[:, render.DEPTH], "diffuse": out_ray[:, render.DIFFUSE],
                   "specular": out_ray[:, render.SPECULAR], "T_left": out_ray[:, render.T_LEFT], "weights": w2, "rgb": out_ray[:, render.RGB]}
            if train:
                out["l2_reg_specular"] = out_ray[:, render.W_SPEC2].sum() / (3.0 * B)   # = (w.detach() * spec ** 2).sum(1).mean() of the torch form below
            return out
        # other level counts (BASELINE configs[0]: 8), or hip_decoder = False: the torch graph
        sigma, dif, spec, tint = self.decoder(feats, d[:, None, :].expand(B, S, 3), wf)
        w2, T_left = composite_weights(sigma[..., 0], dist, d, False)
        w = w2[..., None]
        out = {"valid": valid, "depth": (w[..., 0] * z).sum(1), "diffuse": (w * dif).sum(1),
               "specular": (w * tint * spec).sum(1), "T_left": T_left, "weights": w[..., 0]}
        out["rgb"] = torch.clamp(out["diffuse"] + out["specular"], 0, 1)
        if train:
            out["l2_reg_specular"] = (w.detach() * spec ** 2).sum(1).mean()
        return out

    # ---- "fused" path -----------------------------------------------------------------------
    @torch.no_grad()
    def render_fore_fused(self, rays_o, rays_d, S, global_step):
        z, dist = self.sample(rays_o, rays_d, S)
        valid = torch.all(z != -1, dim=-1)
        self.packed.pack(self.decoder.blob(), network.weight_feature(global_step, self.device), (network.skip_levels(global_step) if LEVEL_SKIP else 0))
        table = self.gather_table()
        out, w = render.render_forward(rays_o, rays_d, z, dist, table, self.resolution, self.packed,
                                       self.min_bbox.tolist(), self.bbox_size.tolist(), render.FORE, False,
                                       ray_valid=valid)
        return out, w, valid

    # ---- background branch: inverse-depth sampling beyond the 2x box (hashgrid/__init__.py:306-337) ----
    @torch.no_grad()
    def inverse_z_sampling(self, rays_o, rays_d, S, invalid_underground=False):
        """hashgrid/__init__.py:306-337 (see inverse_z_samples)."""
        return inverse_z_samples(rays_o, rays_d, self._center_dev, self._half_dev, S, invalid_underground,
                                 floor_y=(self._center_dev - self._size_dev / 4.0)[1])

    @torch.no_grad()
    def render_rays_fused(self, rays_o, rays_d, S_fg, S_bg, global_step, invalid_underground=False, occlusion_mask=None):
        """tile.py:639-692 on the fused kernels: foreground (occupancy-sampled, contract_fore) and
        background (inverse-z, contract_bg, infinity) renders, merged with the foreground's T_left.
        occlusion_mask [B,1] bool (tile.py:655,661): both branches' valid sets are ANDed with it
        (hashgrid/__init__.py:420-421,479-480); a masked ray renders as zeros with T_left = 1, as every invalid ray."""
        self.packed.pack(self.decoder.blob(), network.weight_feature(global_step, self.device), (network.skip_levels(global_step) if LEVEL_SKIP else 0))
        table = self.gather_table()
        box = (self.min_bbox.tolist(), self.bbox_size.tolist())
        z, dist = self.sample(rays_o, rays_d, S_fg)
        vf = torch.all(z != -1, dim=-1)
        if occlusion_mask is not None:
            vf = vf & occlusion_mask[..., 0]
        fg, wfg = render.render_forward(rays_o, rays_d, z, dist, table, self.resolution, self.packed, *box, render.FORE,
                                        False, ray_valid=vf)
        zb, db, vb = self.inverse_z_sampling(rays_o, rays_d, S_bg, invalid_underground)
        if occlusion_mask is not None:
            vb = vb & occlusion_mask[..., 0]
        bg, wbg = render.render_forward(rays_o, rays_d, zb, db, table, self.resolution, self.packed, *box, render.BG,
                                        True, ray_valid=vb)
        T = fg[:, render.T_LEFT, None]
        return {"fore_valid": vf, "bg_valid": vb, "T_left": T,
                "pred_color": fg[:, render.RGB] + T * bg[:, render.RGB],
                "pred_depth": fg[:, render.DEPTH, None] + T * bg[:, render.DEPTH, None],
                "pred_specular": fg[:, render.SPECULAR] + T * bg[:, render.SPECULAR],
                "pred_diffuse": fg[:, render.DIFFUSE] + T * bg[:, render.DIFFUSE],
                "fg": fg, "bg": bg, "fg_weights": wfg, "bg_weights": wbg}

    # ---- optimiser on the table: fused sparse Adam (only touched entries move) --------------
    @torch.no_grad()
    def table_adam(self, lr, betas=(0.9, 0.99), eps=1e-15):
        g = self.features.grad
        K = self.features.numel() // 8
        adam_step_cuda(self.features.data.view(K, 8), g.view(K, 8), self.exp_avg.view(K, 8),
                       self.exp_avg_sq.view(K, 8), lr, betas[0], betas[1], eps, self.adam_step)
        self.adam_step += 1
        self._half_table = None  # (the fused path's epilogue keeps it in step instead: train_step_fused)


def sphere_shell_occupancy(model, radius, thickness):
    """Synthetic sampler occupancy: cells of the tile's sampling grid whose centre lies within
    thickness/2 of a sphere of `radius` around the tile centre -> bool grid shaped like model.occupied_grid."""
    dims = [int(2 ** k) for k in model.log2dim.tolist()]
    corner, size = model.occ_corner.cpu(), model.occ_size.cpu()
    axes = [corner[a] + (torch.arange(dims[a]) + 0.5) * (size[a] / dims[a]) - model.bbox_center[a] for a in range(3)]
    r = torch.sqrt(axes[0][:, None, None] ** 2 + axes[1][None, :, None] ** 2 + axes[2][None, None, :] ** 2)
    return (torch.abs(r - radius) <= thickness / 2.0).to(model.device).contiguous()


class KernelTimer:
    """HIP-event timing of named sections on torch's current stream (the stream every scanerf
    kernel is launched on).  bench.py uses it for the live per-kernel durations behind
    `roofline`; alg_bytes = algorithmic bytes of ONE launch of that section."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.events = {}
        self.bytes = {}
        self.flops = {}
        self.count = 0

    class _Section:
        def __init__(self, timer, name, alg_bytes, alg_flops=0):
            self.t, self.name, self.alg, self.flp = timer, name, alg_bytes, alg_flops

        def __enter__(self):
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

        def __exit__(self, *exc):
            self.e1.record()
            self.t.events.setdefault(self.name, []).append((self.e0, self.e1))
            self.t.bytes[self.name] = self.alg
            self.t.flops[self.name] = self.flp
            self.t.count += 1

    def section(self, name, alg_bytes=0, alg_flops=0):
        return KernelTimer._Section(self, name, alg_bytes, alg_flops)

    def summary(self):
        torch.cuda.synchronize()
        return {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in self.events.items()}

    def dominant(self, *_):
        """(name, avg ms, algorithmic bytes, algorithmic flops) of the section with the largest time."""
        s = self.summary()
        cand = {k: v for k, v in s.items() if self.bytes.get(k, 0) > 0} or s   # (sections without a byte count: time only)
        if not cand:
            return None, 0.0, 0, 0
        name = max(cand, key=cand.get)
        return name, cand[name], self.bytes[name], self.flops.get(name, 0)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _sec(timer, name, alg_bytes=0, alg_flops=0):
    return timer.section(name, alg_bytes, alg_flops) if timer is not None else _Null()


MLP_FLOPS_PER_SAMPLE = 2 * 13728  # 27 456 FLOP per sample forward


def train_step_ops(model, dec_opt, rays_o, rays_d, target, S, global_step, table_lr=1e-2, timer=None):
    """One iteration of tile.py:880-1015 on the foreground branch: MSE + 0.01*l2_reg_specular
    (criterions.py:142-144, tile.py:999), decoder by torch Adam, table by the fused sparse Adam."""
    import sys
    _enc = sys.modules[__package__ + ".hashgrid.PyHashGridBG"]  # the module (the package re-exports the class)
    _enc.TIMER = timer
    model.features.grad = None
    dec_opt.zero_grad(set_to_none=True)
    with _sec(timer, "forward_total"):
        out = model.render_fore_ops(rays_o, rays_d, S, global_step, train=True)
        loss = F.mse_loss(out["rgb"], target[out["valid"]]) + 0.01 * out["l2_reg_specular"]
    with _sec(timer, "backward_total"):
        loss.backward()
    with _sec(timer, "sparse_adam", model.features.numel() * 32):
        model.table_adam(table_lr)
    dec_opt.step()
    _enc.TIMER = None
    return loss.detach()


# Route switches of the fused steps (module attributes; tests and A/B timings flip them, nothing reads the environment):
LARGE_T_ROUTE = "dfeat"     # tables above 2^21 entries: "dfeat" = stand-alone scatter from the backward's dfeat (default), "fused" = the
                            # backward's own records + the split pass
FORWARD_PLAN = True         # the forward launch counts the backward's record ranges (no separate plan launch)
LEVEL_SKIP = True           # levels whose coarse-to-fine weight is exactly zero are left out of gathers and records (same results)
JSTASH = True               # pose gradients: the forward stashes the encoder's position Jacobians for the backward


def train_step_fused(model, dec_opt, rays_o, rays_d, target, S, global_step, table_lr=1e-2, timer=None,
                     pose_grads=False, fused_scatter=None, compact_rays=None, overlap_plan=False, dec_step=True,
                     fused_adam=True):
    """The same iteration as train_step_ops on the fused kernels: one launch for the render forward,
    one for its adjoint, the atomic-free binned scatter for the table gradient, fused sparse Adam.
    fused_adam (default): the sparse Adam on the table runs in the accumulate's epilogue (no gradient table, model.features.grad
    is NOT set); False keeps accumulate -> model.features.grad -> adam_step_cuda (the binding-surface op).
    pose_grads=True also returns dL/d(rays_o), dL/d(rays_d) (feed them to the pose graph:
    torch.autograd.backward([rays_o, rays_d], [g_o, g_d]) -- camera_utils.py:65-84 in the reference)."""
    B = rays_o.shape[0]
    dev = model.device
    with torch.no_grad():
        with _sec(timer, "sample_points_grid", B * (24 + 2 * 4 * S)):
            z, dist = model.sample(rays_o, rays_d, S)
        valid = render.ray_valid(z)  # all(z != -1) per ray (hashgrid/__init__.py:419)
        if compact_rays is None:  # a fully occupied sampler grid cannot produce invalid rays from inside the tile
            compact_rays = not getattr(model, "_occ_full", False) and not pose_grads
        if compact_rays:
            # valid-ray compaction (hashgrid/__init__.py:419-421: the reference renders rays_o[valid] only) in one HIP launch
            # (csrc/compact.hip: wave ballot + popcount prefix sums): the fused backward runs its waves in lock step, so an
            # invalid ray costs as much as a valid one there
            with _sec(timer, "compact_rays", B * (36 + 8 * S)):
                n, co, cd, ct, cz, cdist = render.compact_rays(valid, rays_o, rays_d, target, z, dist)
            if n < B:
                rays_o, rays_d, target, z, dist = co, cd, ct, cz, cdist
                B = n
                valid = None
            if B == 0:
                return torch.zeros((), device=dev)
        wf = model.weight_feature(global_step)
        blob = model.decoder.blob()
        model.packed.pack(blob, wf, (network.skip_levels(global_step) if LEVEL_SKIP else 0))
        ntile = (S + 31) // 32
        tile_T = torch.empty((B, render.tile_T_columns(S)), device=dev)
        bwd_arith = render.backward_arith(True, pose_grads)
        xstash = torch.empty((B * S, 32), device=dev)  # encoder outputs: 1 GB at 65 536 x 128, saves the re-gather
        box = (model.min_bbox.tolist(), model.bbox_size.tolist(), render.FORE, False)
        T = model.features.shape[1]
        if fused_scatter is None:
            # records emitted by the backward kernel need the level's cursors in ITS LDS (256 buckets per level): above 2^21
            # entries its buckets (T / 256) outgrow the accumulate's LDS image.  Round 4 put a split pass in front of the
            # accumulate (csrc/scatter.hip k_bin_split: T = 2^24, 16 384 rays: 9.6 -> 6.1 ms per step on the fused route), but
            # the stand-alone scatter from dfeat, which emits straight into 2^13-entry buckets, is still ahead there (5.6 ms):
            # it stays the default above 2^21; LARGE_T_ROUTE = "fused" selects the fused route
            fused_scatter = T <= (1 << 21) or LARGE_T_ROUTE == "fused"
        fused = fused_scatter and render.scatter_supported(B, S, T)
        ws = plan_done = None
        side = model._side_stream if overlap_plan else None  # (the plan then has no timer section of its own)
        if fused and side is not None:
            # the record plan depends on the sample positions only: count + scan on a side stream, under the forward.
            # Measured: no gain (the forward slows from 3.14 to 3.6 ms, exactly the 0.25 ms of the plan plus contention),
            # so it is off by default
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ws = render.scatter_plan(rays_o, rays_d, z, model.resolution, T, *box, ray_valid=valid, arith=bwd_arith, skip_levels=model.packed.skip_levels)
                plan_done = torch.cuda.Event()
                plan_done.record(side)
        # gather table: the fp32 master itself, or its resident bf16/f16 copy (configs[2]: half the gather bytes, fp32 accumulate)
        table = model.gather_table()
        with _sec(timer, "render_forward", B * (24 + 20 + S * 16 * 8 * 2 * table.element_size()),
                  B * S * MLP_FLOPS_PER_SAMPLE):
            # t16 backward on the same grid as the forward: the forward kernel counts the scatter records itself (its hash
            # indices are the plan's) -- no separate plan launch (0.25 ms at configs[1])
            jstash = None
            if (fused and ws is None and bwd_arith in render._capi.T16_FAMILY and render.forward_plan_supported(B, S, T)
                    and FORWARD_PLAN):
                # pose refinement: the forward also stashes the encoder's position Jacobians (it has the corner values in
                # registers), so that the backward can chain the feature gradients to the rays without a second pass over the table
                if pose_grads and table.dtype == torch.float32 and JSTASH:
                    jstash = torch.empty(render.jstash_shape(B, S), dtype=render.JSTASH_DTYPE, device=dev)
                out, _, ws = render.render_forward(rays_o, rays_d, z, dist, table, model.resolution, model.packed, *box,
                                                   ray_valid=valid, want_weights=False, tile_T=tile_T, xstash=xstash, plan=True,
                                                   jstash=jstash)
            else:
                out, _ = render.render_forward(rays_o, rays_d, z, dist, table, model.resolution, model.packed, *box,
                                               ray_valid=valid, want_weights=False, tile_T=tile_T, xstash=xstash)
    # loss and dL/d(out_ray) in two launches (the torch graph for it was ~60 tiny kernels with host-bound gaps)
    loss, grad_out = render.photometric_loss_grad(out, target, valid, 0.01)
    with torch.no_grad():
        # the scatter ends in the Adam epilogue: no gradient table (only the overflow table, never filled per step)
        adam_epilogue = fused_adam
        gtab = model.overflow_grad() if adam_epilogue else torch.zeros_like(model.features)
        gblob = torch.zeros(network.PARAMSIZE, device=dev)
        ray_bufs = (torch.zeros(B, ntile, device=dev), torch.zeros(B, 2, 64, device=dev)) if pose_grads else None
        g_o = g_d = None
        if fused and ws is None:
            # count + scan of the scatter records (depends on the sample positions only)
            with _sec(timer, "scatter_plan", B * S * 4):
                ws = render.scatter_plan(rays_o, rays_d, z, model.resolution, T, *box, ray_valid=valid, arith=bwd_arith, skip_levels=model.packed.skip_levels)
        elif plan_done is not None:
            torch.cuda.current_stream().wait_event(plan_done)
        # forward recompute + activation gradients + weight gradients = 3x the forward MLP FLOPs
        with _sec(timer, "render_backward", B * (24 + 20 + S * 16 * 8 * 2 * 4 + S * 16 * 8), 3 * B * S * MLP_FLOPS_PER_SAMPLE):
            ray_pos = torch.zeros(B, 6, device=dev) if jstash is not None else None
            dfeat, _ = render.render_backward(rays_o, rays_d, z, dist, table, model.resolution, model.packed, wf,
                                              *box, out, tile_T, grad_out, ray_valid=valid, grad_blob=gblob, xstash=xstash,
                                              ray_grad_buffers=ray_bufs, scatter=(ws, gtab) if fused else None,
                                              want_dfeat=(pose_grads and jstash is None) or not fused, arith=bwd_arith,
                                              jstash=jstash, ray_pos_grad=ray_pos)
        if pose_grads and jstash is not None:
            g_o, g_d = render.ray_gradients_fused(rays_o, rays_d, blob, ray_pos, ray_bufs[0], ray_bufs[1], ray_valid=valid)
        elif pose_grads:
            g_o, g_d = render.ray_gradients(rays_o, rays_d, z, model.features, model.resolution, blob, box[0], box[1],
                                            box[2], dfeat, ray_bufs[0], ray_bufs[1], ray_valid=valid)
        if adam_epilogue and fused:
            with _sec(timer, "table_grad_accumulate_adam", B * S * 16 * 64 + model.features.numel() * 28):
                render.scatter_accumulate_adam(ws, model.features.data, model.exp_avg, model.exp_avg_sq, table_lr, 0.9, 0.99, 1e-15,
                                               model.adam_step, B, S, half_table=model._half_table, overflow_grad=gtab)
            model.adam_step += 1
        elif adam_epilogue and render.scatter_rays_supported(T, bwd_arith) and model._half_table is None:
            # large tables: the stand-alone scatter places the samples itself (rays + depths), same epilogue
            with _sec(timer, "table_grad_scatter_adam", B * S * 16 * (8 + 16 * 8)):
                render.scatter_table_grad_adam_rays(rays_o, rays_d, [(z, dfeat, valid, render.FORE)], box[0], box[1], model.resolution,
                                                    model.features.data, model.exp_avg, model.exp_avg_sq, table_lr, 0.9, 0.99, 1e-15,
                                                    model.adam_step, overflow_grad=gtab, fp16_moments=model.fp16_moments)
            model.adam_step += 1
        elif adam_epilogue:   # (the other record formats: contracted points from torch, stand-alone binned scatter from dfeat)
            pts = ((rays_o[:, None, :] + z[:, :, None] * rays_d[:, None, :]).reshape(-1, 3) - model._min_dev) \
                / model._size_dev * 4.0 - 2.0
            with _sec(timer, "table_grad_scatter_adam", B * S * 16 * (8 + 16 * 8)):
                render.scatter_table_grad_adam(pts.contiguous(), dfeat, model.resolution, model.features.data, model.exp_avg,
                                               model.exp_avg_sq, table_lr, 0.9, 0.99, 1e-15, model.adam_step,
                                               half_table=model._half_table, overflow_grad=gtab,
                                               compact_records=render.compact_record_format(bwd_arith))
            model.adam_step += 1
        elif fused:
            with _sec(timer, "table_grad_accumulate", B * S * 16 * 64):
                render.scatter_accumulate(ws, gtab, B, S)
        else:
            pts = ((rays_o[:, None, :] + z[:, :, None] * rays_d[:, None, :]).reshape(-1, 3) - model._min_dev) \
                / model._size_dev * 4.0 - 2.0
            with _sec(timer, "table_grad_scatter", B * S * 16 * (8 + 16 * 8)):
                render.scatter_table_grad(pts.contiguous(), dfeat, gtab, model.resolution,
                                          compact_records=render.compact_record_format(bwd_arith) if T > (1 << 21) else -1)
        if not adam_epilogue:
            model.features.grad = gtab
            with _sec(timer, "sparse_adam", model.features.numel() * 28):
                model.table_adam(table_lr)
        model.decoder.params.grad = gblob
        if dec_step:  # False: the caller steps the optimiser itself (it holds more parameter groups: camera poses)
            dec_opt.step()
    return (loss[0], g_o, g_d) if pose_grads else loss[0]


def fgbg_gradients(model, rays_o, rays_d, target, S_fg, S_bg, global_step, invalid_underground=False, timer=None,
                   pose_grads=False, collect=None, collect_rays=False):
    """Loss and parameter gradients of the complete per-tile render of tile.py:639-692 / :880-1015: foreground
    (occupancy-sampled, contract_fore) + T_left * background (inverse-z, contract_bg, infinity), MSE on the merged colour
    over all rays + 0.01 * (l2_reg_specular of both branches) -- two fused forward/backward pairs over the same table and
    decoder.  Returns (loss, grad_table [16,T,2], grad_blob [13994]) (+ dL/d(rays_o), dL/d(rays_d) with pose_grads: t16
    backward on the fp32 table, any table size).  collect: a list that receives (contracted points [N,3], dfeat [16,N,2]) of
    each branch INSTEAD of their scatter into the gradient table (the caller scatters them together: large tables); with
    collect_rays it receives (z [B,S], dfeat, ray_valid, contract mode) -- the input of render.scatter_table_grad_adam_rays."""
    B = rays_o.shape[0]
    dev = model.device
    T = model.features.shape[1]
    if pose_grads and render.backward_arith(True, True) not in render._capi.T16_FAMILY:
        raise RuntimeError("scanerf: fgbg pose gradients need the t16 backward (render.set_arith)")
    g_o = g_d = None
    with torch.no_grad():
        wf = model.weight_feature(global_step)
        model.packed.pack(model.decoder.blob(), wf, (network.skip_levels(global_step) if LEVEL_SKIP else 0))
        box = (model.min_bbox.tolist(), model.bbox_size.tolist())
        branches = []
        z, dist = model.sample(rays_o, rays_d, S_fg)
        branches.append((z, dist, torch.all(z != -1, dim=-1), render.FORE, False))
        zb, db, vb = model.inverse_z_sampling(rays_o, rays_d, S_bg, invalid_underground)
        branches.append((zb, db, vb, render.BG, True))
        outs, state = [], []
        for z_, d_, v_, mode, inf in branches:
            S = z_.shape[1]
            tile_T = torch.empty((B, render.tile_T_columns(S)), device=dev)
            xs = torch.empty((B * S, 32), device=dev)
            js = torch.empty(render.jstash_shape(B, S), dtype=render.JSTASH_DTYPE, device=dev) if pose_grads else None
            out, _ = render.render_forward(rays_o, rays_d, z_, d_, model.features, model.resolution, model.packed, *box, mode, inf,
                                           ray_valid=v_, want_weights=False, tile_T=tile_T, xstash=xs, jstash=js)
            outs.append(out)
            state.append((tile_T, xs, js))
    # merge and loss on the per-ray outputs (tile.py:666-690; criterions.py:142-144; tile.py:999), two HIP launches
    vf, vbg = branches[0][2], branches[1][2]
    loss, gfg, gbg = render.photometric_loss_grad_fgbg(outs[0], outs[1], target, vf, vbg, 0.01)

    class _Leaf:  # (what the loop below reads from the former autograd leaves)
        def __init__(self, g):
            self.grad = g
    fg, bg = _Leaf(gfg), _Leaf(gbg)
    with torch.no_grad():
        gtab = torch.zeros_like(model.features) if collect is None else None
        gblob = torch.zeros(network.PARAMSIZE, device=dev)
        for (z_, d_, v_, mode, inf), out, leaf, (tile_T, xs, js) in zip(branches, outs, (fg, bg), state):
            S = z_.shape[1]
            fused = T <= (1 << 21) and render.scatter_supported(B, S, T) and collect is None  # (see train_step_fused)
            ws = render.scatter_plan(rays_o, rays_d, z_, model.resolution, T, *box, mode, inf, ray_valid=v_, skip_levels=model.packed.skip_levels) if fused else None
            bufs = (torch.zeros(B, (S + 31) // 32, device=dev), torch.zeros(B, 2, 64, device=dev)) if pose_grads else None
            rp = torch.zeros(B, 6, device=dev) if pose_grads else None
            with _sec(timer, "render_backward"):
                dfeat, _ = render.render_backward(rays_o, rays_d, z_, d_, model.features, model.resolution, model.packed, wf, *box,
                                                  mode, inf, out, tile_T, leaf.grad.contiguous(), ray_valid=v_, grad_blob=gblob,
                                                  xstash=xs, scatter=(ws, gtab) if fused else None, want_dfeat=not fused,
                                                  ray_grad_buffers=bufs, jstash=js, ray_pos_grad=rp)
            if pose_grads:
                go_b, gd_b = render.ray_gradients_fused(rays_o, rays_d, model.decoder.blob().detach(), rp, bufs[0], bufs[1], ray_valid=v_)
                g_o, g_d = (go_b, gd_b) if g_o is None else (g_o + go_b, g_d + gd_b)
            if fused:
                render.scatter_accumulate(ws, gtab, B, S)
            elif collect is not None and collect_rays:
                collect.append((z_, dfeat, v_, mode))   # (scatter_table_grad_adam_rays places the samples itself)
            else:
                pts = (rays_o[:, None, :] + z_[:, :, None] * rays_d[:, None, :]).reshape(-1, 3)
                pts = (pts - model._min_dev) / model._size_dev * 4.0 - 2.0
                if mode == render.BG:
                    linf = pts.abs().amax(-1, keepdim=True)
                    pts = pts * ((2.0 - 1.0 / linf) / linf)
                if collect is not None:
                    collect.append((pts.contiguous(), dfeat))
                else:
                    render.scatter_table_grad(pts.contiguous(), dfeat, gtab, model.resolution)
    return (loss[0].detach(), gtab, gblob, g_o, g_d) if pose_grads else (loss[0].detach(), gtab, gblob)


def train_step_fgbg(model, dec_opt, rays_o, rays_d, target, S_fg, S_bg, global_step, table_lr=1e-2,
                    invalid_underground=False, timer=None, pose_grads=False, dec_step=True):
    """One complete training iteration of a tile (tile.py:880-1015: foreground + T_left * background, tile.py:639-692) on the
    fused kernels: both branches' forward, ONE loss launch pair for the merged prediction, both branches' backward emitting
    their scatter records, and ONE accumulate + sparse Adam over both record sets (the two gradients meet in one Adam step).
    Falls back to gradient tables + adam_step_cuda where the fused scatter does not apply (tables above 2^21 entries).
    pose_grads=True (fp32 tables, t16 backward): also returns dL/d(rays_o), dL/d(rays_d) of the merged prediction -- the sum of
    the two branches' ray gradients, each formed inside its backward launch from the forward's position Jacobians
    (-> (loss, g_o, g_d)); dec_step=False: the caller steps the decoder's optimiser (it holds the camera parameters too)."""
    B = rays_o.shape[0]
    dev = model.device
    T = model.features.shape[1]
    fused = (T <= (1 << 21) or LARGE_T_ROUTE == "fused") and render.scatter_supported(B, S_fg, T) \
        and render.scatter_supported(B, S_bg, T) and render.backward_arith() != render._capi.ARITH_F32
    if pose_grads and model.gather_table().dtype != torch.float32:
        raise RuntimeError("scanerf: train_step_fgbg(pose_grads=True) gathers from the fp32 table")
    if not fused:
        # tables above 2^21 entries (the reference's default is 2^24): the two branches' feature gradients go through ONE
        # stand-alone binned scatter that ends in the sparse Adam (no gradient table, no zero-fill, no dense optimiser scan)
        binned = T > (1 << 21) and render.backward_arith() != render._capi.ARITH_F32 and model._half_table is None
        parts = [] if binned else None
        by_rays = binned and render.scatter_rays_supported(T, render.backward_arith())
        r = fgbg_gradients(model, rays_o, rays_d, target, S_fg, S_bg, global_step, invalid_underground, timer, pose_grads=pose_grads,
                           collect=parts, collect_rays=by_rays)
        loss, gtab, gblob = r[:3]
        with torch.no_grad():
            if by_rays:
                with _sec(timer, "table_grad_scatter_adam", B * (S_fg + S_bg) * 16 * (8 + 16 * 8)):
                    render.scatter_table_grad_adam_rays(rays_o, rays_d, parts, model.min_bbox.tolist(), model.bbox_size.tolist(),
                                                        model.resolution, model.features.data, model.exp_avg, model.exp_avg_sq, table_lr,
                                                        0.9, 0.99, 1e-15, model.adam_step, overflow_grad=model.overflow_grad(),
                                                        fp16_moments=model.fp16_moments)
                model.adam_step += 1
            elif binned:
                pts = torch.cat([p_ for p_, _ in parts], 0)
                dfe = torch.cat([f_ for _, f_ in parts], 1).contiguous()
                with _sec(timer, "table_grad_scatter_adam", pts.shape[0] * 16 * (8 + 16 * 8)):
                    render.scatter_table_grad_adam(pts, dfe, model.resolution, model.features.data, model.exp_avg, model.exp_avg_sq,
                                                   table_lr, 0.9, 0.99, 1e-15, model.adam_step, overflow_grad=model.overflow_grad(),
                                                   compact_records=render.compact_record_format(render.backward_arith()))
                model.adam_step += 1
            else:
                model.features.grad = gtab
                model.table_adam(table_lr)
            model.decoder.params.grad = gblob
            if dec_step:
                dec_opt.step()
        return (loss, r[3], r[4]) if pose_grads else loss
    with torch.no_grad():
        wf = model.weight_feature(global_step)
        model.packed.pack(model.decoder.blob(), wf, (network.skip_levels(global_step) if LEVEL_SKIP else 0))
        box = (model.min_bbox.tolist(), model.bbox_size.tolist())
        table = model.gather_table()
        with _sec(timer, "sample_points_grid", B * (24 + 2 * 4 * S_fg)):
            z, dist = model.sample(rays_o, rays_d, S_fg)
        vf = render.ray_valid(z)
        zb, db, vb = model.inverse_z_sampling(rays_o, rays_d, S_bg, invalid_underground)
        branches = ((z, dist, vf, render.FORE, False, S_fg), (zb, db, vb, render.BG, True, S_bg))
        need_bg = render.lib().scanerf_render_scatter_workspace_bytes(B, S_bg, T)
        if getattr(model, "_ws_bg", None) is None or model._ws_bg.numel() < need_bg:
            model._ws_bg = torch.empty(need_bg, dtype=torch.uint8, device=dev)  # the background branch's own record workspace
        outs, state = [], []
        for (z_, d_, v_, mode, inf, S), wsbuf in zip(branches, (None, model._ws_bg)):
            tile_T = torch.empty((B, render.tile_T_columns(S)), device=dev)
            xs = torch.empty((B * S, 32), device=dev)
            # (the forward launch reserves the backward's record ranges as well where the two kernels share a grid)
            in_fwd = (render.backward_arith(True, False) in render._capi.T16_FAMILY and render.forward_plan_supported(B, S, T)
                      and FORWARD_PLAN)
            js = torch.empty(render.jstash_shape(B, S), dtype=render.JSTASH_DTYPE, device=dev) if pose_grads else None
            with _sec(timer, "render_forward", B * (24 + 20 + S * 16 * 8 * 2 * table.element_size()), B * S * MLP_FLOPS_PER_SAMPLE):
                r = render.render_forward(rays_o, rays_d, z_, d_, table, model.resolution, model.packed, *box, mode, inf,
                                          ray_valid=v_, want_weights=False, tile_T=tile_T, xstash=xs, plan=in_fwd, plan_workspace=wsbuf,
                                          jstash=js)
            outs.append(r[0])
            state.append((tile_T, xs, r[2] if in_fwd else None, js))
        loss, gfg, gbg = render.photometric_loss_grad_fgbg(outs[0], outs[1], target, vf, vb, 0.01)
        gblob = torch.zeros(network.PARAMSIZE, device=dev)
        overflow = model.overflow_grad()
        wss = []
        g_o = g_d = None
        for (z_, d_, v_, mode, inf, S), out, g, (tile_T, xs, ws, js), wsbuf in zip(branches, outs, (gfg, gbg), state, (None, model._ws_bg)):
            if ws is None:
                with _sec(timer, "scatter_plan", B * S * 4):
                    ws = render.scatter_plan(rays_o, rays_d, z_, model.resolution, T, *box, mode, inf, ray_valid=v_, workspace=wsbuf, skip_levels=model.packed.skip_levels)
            bufs = (torch.zeros(B, (S + 31) // 32, device=dev), torch.zeros(B, 2, 64, device=dev)) if pose_grads else None
            rp = torch.zeros(B, 6, device=dev) if pose_grads else None
            with _sec(timer, "render_backward", B * (24 + 20 + S * 16 * 8 * 2 * 4 + S * 16 * 8), 3 * B * S * MLP_FLOPS_PER_SAMPLE):
                render.render_backward(rays_o, rays_d, z_, d_, table, model.resolution, model.packed, wf, *box, mode, inf, out,
                                       tile_T, g, ray_valid=v_, grad_blob=gblob, xstash=xs, scatter=(ws, overflow), want_dfeat=False,
                                       ray_grad_buffers=bufs, jstash=js, ray_pos_grad=rp)
            if pose_grads:
                go_b, gd_b = render.ray_gradients_fused(rays_o, rays_d, model.decoder.blob().detach(), rp, bufs[0], bufs[1], ray_valid=v_)
                g_o, g_d = (go_b, gd_b) if g_o is None else (g_o + go_b, g_d + gd_b)
            wss.append(ws)
        with _sec(timer, "table_grad_accumulate_adam", B * (S_fg + S_bg) * 16 * 64 + model.features.numel() * 28):
            render.scatter_accumulate_adam2(wss[0], S_fg, wss[1], S_bg, model.features.data, model.exp_avg, model.exp_avg_sq,
                                            table_lr, 0.9, 0.99, 1e-15, model.adam_step, B, half_table=model._half_table,
                                            overflow_grad=overflow)
        model.adam_step += 1
        model.decoder.params.grad = gblob
        if dec_step:
            dec_opt.step()
    return (loss[0], g_o, g_d) if pose_grads else loss[0]
