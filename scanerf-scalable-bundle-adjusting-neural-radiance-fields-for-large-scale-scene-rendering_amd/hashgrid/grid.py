"""`HashGrid`: the reference's scene-representation module (hashgrid/__init__.py:32-596) with its constructor arguments,
attributes and method names, on this package's HIP ops -- what tile.py constructs (tile.py:114-125) and calls
(tile.py:639-692: render_fore_rays / render_bg_rays; :866-877: pruning_grid; :510-531: export).

render_batch_rays has two routes with the same outputs (the reference's dictionary):
  * fused (default where it applies: 16 levels, contract_fore / contract_bg, a decoder that exposes its blob -- this package's
    network.ShallowMLP or tile_model.Decoder): ONE differentiable op, render.FusedRenderRays (one HIP launch forward, one
    backward); every loss a caller builds on rgb / depth / diffuse / specular / tint / T_left / l2_reg_specular reaches the
    table, the decoder and the rays through loss.backward();
  * op by op (any decoder module, `fused = False`): the reference's own sequence -- HIP encoder op through the autograd
    wrapper, decoder(inputs, weight_feature=...), torch compositing (hashgrid/__init__.py:545-594).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from .. import render
from ..cuda import sample_points_grid, voxelize_mesh
from .lib.HASHGRID import Sampler
from .PyHashGridBG import PyHashGridBG

TRAIN, INFERENCE = 0, 1   # cfg.py:1-2


class HashGrid(nn.Module):
    def __init__(self, device, bbox_corner, bbox_size, log2_hashmap_size=24, grid_resolution=(32, 2048), sampler_log2dim=4,
                 init_outside=False, model_path="", near=None, far=None):
        super().__init__()
        self.device = device
        bbox_corner = torch.as_tensor(bbox_corner, dtype=torch.float32)
        bbox_size = torch.as_tensor(bbox_size, dtype=torch.float32)
        self.bbox_center = bbox_corner + bbox_size / 2.0
        self.bbox_size = bbox_size * 2                     # 2 times for the background (:50)
        self.min_bbox = self.bbox_center - self.bbox_size / 2.0
        self.max_bbox = self.bbox_center + self.bbox_size / 2.0
        self.log2_hashmap_size = log2_hashmap_size
        self.finest_resolution = (self.bbox_size / self.bbox_size.min() * grid_resolution[1]).int().cpu()
        self.base_resolution = (self.bbox_size / self.bbox_size.min() * grid_resolution[0]).int().cpu()
        self.HE = PyHashGridBG(self.device, self.min_bbox, self.bbox_size, n_levels=16, n_features_per_level=2,
                               log2_hashmap_size=log2_hashmap_size, base_resolution=self.base_resolution,
                               finest_resolution=self.finest_resolution, init_mode="xavier").to(device)
        self.sampler = Sampler()
        self.last_sampler_log2dim = sampler_log2dim
        self.sampler_log2dim = sampler_log2dim - torch.log2(self.bbox_size.max() / self.bbox_size).int()
        self.occupied_grid = torch.zeros(tuple(int(2 ** k) for k in self.sampler_log2dim), dtype=torch.bool)
        self.outside = torch.zeros_like(self.occupied_grid)
        voxelize_mesh(self.sampler_log2dim.cpu().int(), (self.min_bbox + self.bbox_size / 4.0).cpu(), (self.bbox_size / 2.0).cpu(),
                      model_path, self.occupied_grid, init_outside, self.outside)
        if near is not None and far is not None:
            self.occupied_grid[:, -int(near / far * self.occupied_grid.shape[1]):, :] = False
        self.occupied_grid = self.occupied_grid.to(device)
        self.outside = self.outside.to(device)
        self.grid_resolution = torch.tensor([2 ** int(k) for k in self.sampler_log2dim], dtype=torch.int32, device=device)
        for name in ("bbox_center", "bbox_size", "min_bbox", "max_bbox"):   # (the reference receives device tensors from tile.py)
            setattr(self, name, getattr(self, name).to(device))
        self.fused = True
        self.last_render_route = None

    # ---- checkpoints / deployment files (:94-105, :248-266) ----------------------------------------------------------------
    _CKPT = (("occupied_grid", "occupied_grid"), ("sampler_log2dim", "sampler_log2dim"), ("grid_resolution", "grid_resolution"))

    def export_check_point(self):
        ck = {key: getattr(self, attr).detach().cpu().numpy() for key, attr in self._CKPT}
        ck["features"] = self.HE.features.detach().cpu().numpy()
        return ck

    def load_check_point(self, ckp):
        for key, attr in self._CKPT:
            setattr(self, attr, torch.from_numpy(ckp[key]).to(self.device))
        self.HE.features = nn.Parameter(torch.from_numpy(ckp["features"]).to(self.device))

    def export(self, path):
        from ..renderer import write_feature_npz
        write_feature_npz(path, self.HE.features, self.occupied_grid, self.min_bbox, self.bbox_size, self.sampler_log2dim, self.HE.resolution)

    def load(self, path):
        f = np.load(os.path.join(path, "feature.npz"))
        on = lambda k: torch.from_numpy(f[k]).to(self.device)
        self.HE.features = nn.Parameter(on("features").float())
        self.occupied_grid, self.sampler_log2dim, self.HE.resolution = on("occupied_grid"), on("grid_log2dim"), on("resolution")
        self.block_corner, self.block_size = on("block_corner"), on("block_size")

    def toCPU(self):
        self.HE.resolution = self.HE.resolution.cpu()

    def toGPU(self):
        self.HE.resolution = self.HE.resolution.to(self.device)

    # ---- coarse-to-fine pruning (:138-225) -----------------------------------------------------------------------------------
    @torch.no_grad()
    def pruning_tile_grid(self, global_step, decoder, sub_split=False, pruning_th=0.4, batch_size=92 ** 3):
        from .. import trainer
        wf = self.weight_feature(global_step).repeat_interleave(2, dim=-1)
        self.occupied_grid, self.sampler_log2dim = trainer.prune_occupancy(
            self.occupied_grid, self.sampler_log2dim, self.finest_resolution, self.HE, decoder.inference_sigma, wf, global_step, sub_split,
            pruning_th, batch_size, self.device)
        self.grid_resolution = torch.tensor([2 ** int(k) for k in self.sampler_log2dim], dtype=torch.int32, device=self.device)

    @torch.no_grad()
    def pruning_grid(self, global_step, decoder, log2dim, pruning_th):
        assert log2dim >= self.last_sampler_log2dim, f"log2dim {log2dim} last_sampler_log2dim {self.last_sampler_log2dim}"
        sub_split = log2dim != self.last_sampler_log2dim
        if sub_split:
            self.last_sampler_log2dim = self.last_sampler_log2dim + 1
        self.pruning_tile_grid(global_step, decoder, sub_split=sub_split, pruning_th=pruning_th)

    def weight_feature(self, global_step):
        alpha = max(min(global_step / 10000 * 8 + 8, 16), 0)
        k = torch.arange(16, dtype=torch.float32, device=self.device)
        return (1 - (alpha - k).clamp_(min=0, max=1).mul_(math.pi).cos_()) / 2

    def weight_bg_feature(self, ratio):
        alpha = torch.clamp(ratio * 8 + 8, 0, 16)
        k = torch.arange(16, dtype=torch.float32, device=self.device)
        weight = (1 - (alpha - k[None, ...]).clamp_(min=0, max=1).mul_(math.pi).cos_()) / 2
        return weight.repeat_interleave(2, dim=-1)

    # Small constant tensors the C ABI wants on the other side (host floats of the box; the sampler's log2dim on the device): converted
    # once per VALUE, not per call -- `.tolist()` of a device tensor and `.to(device)` of a pageable host tensor each stall the host
    # until the device has drained (three of the four synchronisations of a step, tools/sync_probe.py).  Keyed by tensor object and
    # version counter: a reassigned or in-place modified attribute (checkpoint load, pruning) is converted again.
    def _converted(self, name, how):
        t = getattr(self, name)
        hit = self.__dict__.setdefault("_conv_cache", {}).get(name + how)
        if hit is None or hit[0] is not t or hit[1] != t._version:   # (the entry holds the tensor: its address cannot be reused meanwhile)
            hit = (t, t._version, t.tolist() if how == "list" else t.to(self.device).int().contiguous())
            self._conv_cache[name + how] = hit
        return hit[2]

    # ---- sampling (:278-337; no gradients) -------------------------------------------------------------------------------------
    def samplePoints(self, rays_o, rays_d, num_sample):
        z_vals = torch.full((rays_o.shape[0], num_sample), -1, dtype=torch.float32, device=self.device)
        dists = torch.full((rays_o.shape[0], num_sample), -1, dtype=torch.float32, device=self.device)
        sample_points_grid(rays_o.detach().contiguous(), rays_d.detach().contiguous(), z_vals, dists,
                           (self.min_bbox + self.bbox_size / 4.0).contiguous(), (self.bbox_size / 2.0).contiguous(),
                           self.occupied_grid, self._converted("sampler_log2dim", "dev_int"))
        return z_vals, dists

    def invalid_sampling_underground(self, rays_o, rays_d, bound):
        exit_y = rays_o[:, 1] + bound[:, 1] * rays_d[:, 1]
        return ~(torch.abs(exit_y - (self.bbox_center - self.bbox_size / 4.0)[1]) < 0.0001)

    @torch.no_grad()
    def inverse_z_sampling(self, rays_o, rays_d, num_sample, invalid_underground=True, perturb=False):
        from ..tile_model import inverse_z_samples
        return inverse_z_samples(rays_o, rays_d, self.bbox_center, self.bbox_size / 2.0, num_sample, invalid_underground,
                                 floor_y=(self.bbox_center - self.bbox_size / 4.0)[1])

    # ---- compositing in torch (:344-366; used by the op-by-op route) ---------------------------------------------------------
    def cal_integrate_weight(self, sigma, z_vals, dists, rays_d, infinity=True):
        from ..tile_model import composite_weights
        w, T_left = composite_weights(sigma[..., 0], dists, rays_d, infinity)
        return w[..., None], T_left

    def accumulate(self, weights, attr):
        return torch.sum(weights * attr, dim=1)

    def contract_fore(self, x):
        return (x - self.min_bbox) / self.bbox_size * 4 - 2, None

    def contract_bg(self, x):
        x = (x - self.min_bbox) / self.bbox_size * 4 - 2
        linf = torch.max(torch.abs(x), dim=-1, keepdim=True)[0]
        return x * ((2 - 1.0 / linf) / linf), None

    # ---- rendering (:413-596) ----------------------------------------------------------------------------------------------------
    def _masked(self, valid, out, B, like_o, like_d):
        if valid is None:   # every ray rendered: the scatter into zero / one filled tensors is the identity
            return out["rgb"], out["depth"], out["T_left"][:, None], out["specular"], out["diffuse"]
        rgb, depth = torch.zeros_like(like_o), torch.zeros_like(like_d[..., :1])
        transparency = torch.ones_like(like_d[..., :1])
        specular, diffuse = torch.zeros_like(like_o), torch.zeros_like(like_o)
        rgb = rgb.index_put((valid,), out["rgb"])
        depth = depth.index_put((valid,), out["depth"])
        transparency = transparency.index_put((valid,), out["T_left"][:, None])
        specular = specular.index_put((valid,), out["specular"])
        diffuse = diffuse.index_put((valid,), out["diffuse"])
        return rgb, depth, transparency, specular, diffuse

    def render_fore_rays(self, rays_o, rays_d, num_sample, decoder, mode, occlusion_mask=None, infinity=False, **kwargs):
        z_vals, dists = self.samplePoints(rays_o, rays_d, num_sample)
        valid = torch.all(z_vals != -1, dim=-1)
        if occlusion_mask is not None:
            valid = valid & occlusion_mask[..., 0]
        # (all rays valid -- a fully occupied grid, rays from inside the tile: the boolean-mask gathers and the scatter back are
        # identities; skipping them saves four [B,S]-sized copies per branch and changes no value)
        sel = None if bool(valid.all()) else valid
        pick = (lambda t: t) if sel is None else (lambda t: t[sel])
        out, ret = self.render_batch_rays(pick(rays_o), pick(rays_d), pick(z_vals), pick(dists), decoder, mode, self.contract_fore,
                                          out_normal=False, infinity=infinity, global_step=kwargs["global_step"])
        if ret is False:
            return None, False
        rgb, depth, transparency, specular, diffuse = self._masked(sel, out, rays_o.shape[0], rays_o, rays_d)
        out_dict = dict(out)
        out_dict.update({"fore_valid": valid, "pred_color": rgb, "pred_depth": depth, "specular": specular, "diffuse": diffuse,
                         "T_left": transparency})
        return out_dict, True

    def render_bg_rays(self, rays_o, rays_d, num_sample, decoder, mode, occlusion_mask=None, infinity=True, **kwargs):
        if kwargs["bg_mode"] == "IZ":
            z_vals, dists, valid = self.inverse_z_sampling(rays_o.detach(), rays_d.detach(), num_sample, kwargs["invalid_underground"])
        else:   # "BS" needs the mesh tracer (fastMesh: out of scope)
            return None, False
        if occlusion_mask is not None:
            valid = valid & occlusion_mask[..., 0]
        sel = None if bool(valid.all()) else valid
        pick = (lambda t: t) if sel is None else (lambda t: t[sel])
        out, ret = self.render_batch_rays(pick(rays_o), pick(rays_d), pick(z_vals), pick(dists), decoder, mode, self.contract_bg,
                                          out_normal=False, infinity=infinity, global_step=kwargs["global_step"])
        if ret is False:
            return None, ret
        rgb, depth, transparency, specular, diffuse = self._masked(sel, out, rays_o.shape[0], rays_o, rays_d)
        out_dict = dict(out)
        out_dict.update({"valid": valid, "rgb": rgb, "depth": depth, "specular": specular, "diffuse": diffuse, "T_left": transparency})
        return out_dict, ret

    def _fused_route(self, decoder, contract_func, out_normal, rays_o):
        if not (self.fused and rays_o.is_cuda and not out_normal and hasattr(decoder, "blob") and self.HE.n_levels == 16):
            return None
        if getattr(decoder, "in_channel", 32) != 32:
            return None
        if contract_func == self.contract_fore:
            return render.FORE
        if contract_func == self.contract_bg:
            return render.BG
        return None

    def render_batch_rays(self, rays_o, rays_d, z_vals, dists, decoder, mode, contract_func, out_normal=False, infinity=False, **kwargs):
        if z_vals.shape[0] == 0:
            return None, False
        global_step = kwargs["global_step"]
        cmode = self._fused_route(decoder, contract_func, out_normal, rays_o)
        if cmode is not None:
            from .. import network
            self.last_render_route = "fused"
            wf = self.weight_feature(global_step).repeat_interleave(2, dim=-1)
            out_ray, weights = render.fused_render_rays(
                rays_o, rays_d, z_vals, dists, self.HE.features, decoder.blob(), self.HE.resolution.to(self.device).int().contiguous(),
                wf, self._converted("min_bbox", "list"), self._converted("bbox_size", "list"), cmode, infinity, None, network.skip_levels(global_step), True)
            return render.render_batch_rays_dict(out_ray, weights, mode is TRAIN or mode == TRAIN), True
        # ---- op by op: encoder op -> decoder module -> compositing op, each a library call behind its own autograd node (the
        # route for a caller that holds the pieces apart: `self.HE`, `decoder`, the per-sample outputs)
        self.last_render_route = "ops"
        B, S = z_vals.shape
        world = (rays_o[:, None, :] + z_vals[..., None] * rays_d[:, None, :]).reshape(-1, 3)
        if out_normal and not world.requires_grad:
            world.requires_grad_(True)
        pts, per_sample_mask = world, None
        if contract_func is not None:
            pts, per_sample_mask = contract_func(world)
        feats = self.HE(pts).reshape(B, S, 32)
        wf = self.weight_feature(global_step).repeat_interleave(2, dim=-1)[None, None, :]
        if per_sample_mask is not None:
            wf = wf * per_sample_mask.reshape(B, S, 32)
        dirs = rays_d[:, None, :].expand(B, S, 3)
        if hasattr(decoder, "forward_parts"):
            dec = decoder.forward_parts(feats, dirs, weight_feature=wf)
        else:   # (any module with the reference's call signature)
            dec = decoder(torch.cat([feats, dirs], -1), weight_feature=wf)
        train = mode is TRAIN or mode == TRAIN
        if rays_o.is_cuda and S <= 512:
            out_ray, weights = render.composite_rays(dec["sigma"], dec["diffuse"], dec["specular"], dec["tint"], z_vals, dists, rays_d, infinity)
            out = render.render_batch_rays_dict(out_ray, weights, train)
        else:   # (CPU tensors never reach here on the product path: the encoder op raises first; kept for S > 512)
            weights, T_left = self.cal_integrate_weight(dec["sigma"], z_vals, dists, rays_d, infinity=infinity)
            acc = lambda v: self.accumulate(weights, v)
            out = {"diffuse": acc(dec["diffuse"]), "tint": acc(dec["tint"]), "specular": acc(dec["tint"] * dec["specular"]),
                   "depth": acc(z_vals[..., None]), "T_left": T_left, "weights": weights}
            out["rgb"] = torch.clamp(out["diffuse"] + out["specular"], 0, 1)
            if train:
                out["l2_reg_specular"] = self.accumulate(weights.detach(), dec["specular"] ** 2).mean()
        if out_normal:   # surface normals: -d(sigma)/d(position), normalised and detached, composited with the weights
            g = torch.autograd.grad(dec["sigma"].sum(), world, retain_graph=True)[0]
            n = -g / (g.norm(2, dim=-1, keepdim=True) + 1e-8)
            out["normal"] = self.accumulate(out["weights"], n.reshape(B, S, 3).detach())   # (weights NOT detached: :586)
        return out, True
