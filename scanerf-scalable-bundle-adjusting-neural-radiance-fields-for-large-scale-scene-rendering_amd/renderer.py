"""Multi-tile novel-view renderer: the build's counterpart of RenderingHashGrid's render path
(rendering.py:93-174 parse_blocks, :286-544 render_rays_base) on the HIP render-time ops, plus the
tile export / load formats it consumes (hashgrid/__init__.py:248-257 feature.npz, tile.py:516-529).

Per view: compute_ray_forward -> ray_block_intersection -> argsort(near) -> per tracing step
{sample_points -> prepare_points -> pts_inference -> accumulate_color} -> update_outgoing_bidx ->
per blended background {inverse_z_sampling -> bg_pts_inference_v2 -> accumulate_color} -> merge.
"""
import os

import numpy as np
import torch

from . import network
from .cuda import compute_ray_forward
from .hashgrid.lib import HASHGRID as _HG
from .hashgrid import (accumulate_color, bg_pts_inference_v2, inverse_z_sampling, prepare_points, process_occupied_grid,
                       SKIP_UNSAMPLED, pts_inference, pts_inference_tracing, ray_block_intersection, sample_points, sort_tracing_blocks,
                       tracing_fusable, update_outgoing_bidx)


def write_feature_npz(path, features, occupied_grid, min_bbox, bbox_size, log2dim, resolution):
    """`feature.npz` as HashGrid.export writes it (hashgrid/__init__.py:248-257): fp16 table, occupancy, the 2x box, the sampler's
    log2dim, the level resolutions -- in that key order (golden G12 compares with a file written by the reference's own writer)."""
    cpu = lambda t: t.detach().cpu().numpy()
    np.savez(os.path.join(path, "feature.npz"), features=cpu(features).astype(np.float16), occupied_grid=cpu(occupied_grid),
             block_corner=cpu(min_bbox), block_size=cpu(bbox_size), grid_log2dim=cpu(log2dim), resolution=cpu(resolution))


def export_tile(path, model):
    """feature.npz (fp16 table, occupancy, 2x-box corner/size, log2dim, resolution) + decoder.pth
    (state dict with the reference's ShallowMLP key names)."""
    os.makedirs(path, exist_ok=True)
    write_feature_npz(path, model.features, model.occupied_grid, model.min_bbox, model.bbox_size, model.log2dim, model.resolution)
    torch.save({k: v.detach().cpu() for k, v in model.decoder.ref_state_dict().items()}, os.path.join(path, "decoder.pth"))


def load_tile(path):
    f = np.load(os.path.join(path, "feature.npz"))
    sd = torch.load(os.path.join(path, "decoder.pth"), map_location="cpu")
    return {"features": f["features"], "occupied_grid": f["occupied_grid"], "block_corner": f["block_corner"],
            "block_size": f["block_size"], "grid_log2dim": f["grid_log2dim"], "resolution": f["resolution"],
            "blob": network.blob_from_state_dict(sd).numpy()}


def render_box(block_corner, block_size):
    """feature.npz stores the 2x HashGrid box (hashgrid/__init__.py:50,254-255); the renderer traces the inner tile box:
    corner + size / 4, size / 2 (rendering.py:164-165)."""
    return block_corner + block_size / 4.0, block_size / 2.0


class TileSetRenderer:
    def __init__(self, device, tiles):
        """tiles: dicts as returned by load_tile (block_corner / block_size describe the 2x HashGrid box;
        the renderer works on the inner tile box: rendering.py:164-165)."""
        self.device = device
        # route switches (A/B timing, tests): slot lists derived inside the inference kernel (pts_inference_tracing) instead of
        # prepare_points + pts_inference; a tracing pass runs on the running rays alone when fewer than this many tenths run
        self.fuse_slots = True
        self.compact_below_tenths = 9
        self.skip_zero_transmittance_background = True   # exact: see render_rays
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(device).contiguous()
        self.feature_tables = t(np.stack([x["features"] for x in tiles]), torch.float16)
        self.params = t(np.stack([x["blob"] for x in tiles]), torch.float32)
        # the decoders' MFMA images are packed ONCE, here, and owned by this renderer (not looked up per call in a cache
        # keyed on an address the allocator may hand to the next renderer's blobs)
        from .hashgrid.lib.HASHGRID import PackedDecoders
        self.packed = PackedDecoders(self.params)
        self.resolution = t(np.stack([x["resolution"] for x in tiles]), torch.int32)
        self.grid_log2dim = t(np.stack([x["grid_log2dim"] for x in tiles]), torch.int32)
        grids = [np.asarray(x["occupied_grid"]).reshape(-1) for x in tiles]
        self.grid_starts = t(np.cumsum([0] + [g.size for g in grids[:-1]]), torch.int64)
        self.occupied_grid = t(np.concatenate(grids), torch.bool)
        corner = t(np.stack([x["block_corner"] for x in tiles]), torch.float32)
        size = t(np.stack([x["block_size"] for x in tiles]), torch.float32)
        self.block_corner, self.block_size = (x.contiguous() for x in render_box(corner, size))
        # sampling grid: each tile's occupancy dilated into the tiles it overlaps (rendering.py:168-173)
        self.fake_occupied_grid = self.occupied_grid.clone()
        for i in range(len(tiles)):
            process_occupied_grid(i, int(torch.prod(2 ** self.grid_log2dim[i]).cpu()), self.block_corner, self.block_size,
                                  self.occupied_grid, self.grid_starts, self.grid_log2dim, self.fake_occupied_grid)

    def compute_rays(self, H, W, K, c2w):
        """Pixel-centre rays of one view (rendering.py:272-284) through the HIP ray kernel."""
        dev = self.device
        j, i = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
        locs = torch.stack([torch.zeros_like(i), i, j], -1).reshape(-1, 3).int().contiguous()
        o = torch.empty(H * W, 3, device=dev)
        d = torch.empty(H * W, 3, device=dev)
        compute_ray_forward(o, d, torch.as_tensor(K, dtype=torch.float32, device=dev).reshape(1, 9).contiguous(),
                            torch.as_tensor(c2w, dtype=torch.float32, device=dev)[:3, :4].reshape(1, 12).contiguous(), locs)
        return o, d

    @torch.no_grad()
    def render_rays(self, rays_o, rays_d, num_sample=128, num_bg_sample=128, sample_range=1e6, layout=2,
                    skip_saturated_background=False):
        """skip_saturated_background: False (default) = the reference's behaviour, every ray gets its blended background
        (rendering.py:534-536); True = rays whose foreground transmittance is <= 1e-5 (the threshold at which rendering.py:356
        stops tracing them) get none: colour changes by <= 1e-5, but the returned DEPTH loses transparency * bg_depth, which can
        reach ~1e-5 * sample_range = 10 units (tests/test_gpu_render_time.py bounds both).
        layout of the per-sample work arrays between the ops (same ops, same arithmetic per sample; scanerf_hip.h
        `sample_major`): 0 = the reference's [B,S]; 2 (default) = [B/32,S,32]: a wave of the inference kernel holds ONE depth
        index of 32 neighbouring rays -- neighbouring pixels share their cells down to the fine levels, so the table gathers of
        a wave fall on a few lines instead of 32 per level -- and walks along those rays; 1 = [S,B] (measured slower: every
        CU on the same depth slab)."""
        dev, nb = self.device, self.block_corner.shape[0]
        lay = int(layout)
        n_rays = rays_o.shape[0]
        # the single-pass comparison kernel (HASHGRID.INFER_ARITH = "f32"; also what > 64 tiles or >= 2^31 samples fall back to)
        # reads the reference's [B,S] arrays only
        single_pass = (_HG.INFER_ARITH == "f32" or nb > 64
                       or (n_rays + 31) * max(num_sample, num_bg_sample) >= 2 ** 31)
        if single_pass:
            lay = 0
        if lay == 2 and n_rays % 32:  # pad with copies of the last ray
            pad = 32 - n_rays % 32
            rays_o = torch.cat([rays_o, rays_o[-1:].expand(pad, 3)]).contiguous()
            rays_d = torch.cat([rays_d, rays_d[-1:].expand(pad, 3)]).contiguous()
        B = rays_o.shape[0]
        sm = lay
        shp = {0: lambda S, *tail: (B, S, *tail), 1: lambda S, *tail: (S, B, *tail), 2: lambda S, *tail: (B // 32, S, 32, *tail)}[lay]
        inter = torch.full((B, nb, 2), 1e7, device=dev)
        ray_block_intersection(rays_o, rays_d, self.block_corner, self.block_size, inter)
        tracing_blocks = sort_tracing_blocks(inter)   # = torch.argsort(inter[..., 0], dim=-1, stable=True)
        max_tracing = int(torch.mean((inter != 1e7).float(), dim=-1).sum(dim=-1).max().cpu())
        transp = torch.ones(B, 1, device=dev)
        dif, spec, depth = torch.zeros(B, 3, device=dev), torch.zeros(B, 3, device=dev), torch.zeros(B, 1, device=dev)
        tracing_idx = torch.zeros(B, 1, dtype=torch.int32, device=dev)
        z_start = torch.zeros(B, 1, device=dev)
        pd = torch.empty(shp(num_sample, 3), device=dev)
        ps = torch.empty(shp(num_sample, 3), device=dev)
        pa = torch.empty(shp(num_sample, 1), device=dev)

        fuse_slots = tracing_fusable(nb) and not single_pass and self.fuse_slots

        def fg_pass(ro, rd, it, tb, tidx, zst, tr, df, sp, dp, running, pd, ps, pa):
            """One tracing pass (rendering.py:356-420) over the rays given: per-ray state tidx / zst / tr / df / sp / dp updated in place."""
            n = ro.shape[0]
            shp_ = {0: lambda S, *tail: (n, S, *tail), 1: lambda S, *tail: (S, n, *tail), 2: lambda S, *tail: (n // 32, S, 32, *tail)}[lay]
            z = torch.full(shp_(num_sample), -1.0, device=dev)
            dd = torch.full(shp_(num_sample), -1.0, device=dev)
            sample_points(ro, rd, self.block_corner, self.block_size, self.fake_occupied_grid, self.grid_starts,
                          self.grid_log2dim, tb, it, tidx, zst, z, dd, sample_major=sm)
            if fuse_slots:   # the slot lists never exist: the inference kernel derives them (pts_inference_tracing); rays that
                # got no sample in this pass (first depth -1) are neither written by it nor read by the accumulation
                pts_inference_tracing(ro, rd, z, dd, running, it, self.feature_tables, self.packed, self.resolution, self.occupied_grid,
                                      self.grid_starts, self.grid_log2dim, self.block_corner, self.block_size, pd, ps, pa,
                                      sample_major=sm | SKIP_UNSAMPLED)
                accumulate_color(pd, ps, pa, tr, z, df, sp, dp, sample_major=sm | SKIP_UNSAMPLED)
                return
            else:
                bi = torch.full(shp_(num_sample, 4), -1, dtype=torch.int16, device=dev)
                prepare_points(z, running, it, bi, sample_major=sm)
                pts_inference(ro, rd, z, dd, bi, self.feature_tables, self.packed, self.resolution, self.occupied_grid,
                              self.grid_starts, self.grid_log2dim, self.block_corner, self.block_size, pd, ps, pa, sample_major=sm)
            accumulate_color(pd, ps, pa, tr, z, df, sp, dp, sample_major=sm)

        # tiles each ray's path meets (the sorted list ends at the first miss): a ray whose tracing index has reached its own count has
        # nothing left to sample -- sample_points leaves it at once (`bound.x == kInf`), so dropping it from `running` here (the
        # reference's mask compares with the view's maximum, rendering.py:356) changes no value and lets the later passes run on
        # the few rays that do cross a second tile
        n_hit = (inter[..., 0] != 1e7).sum(dim=-1, keepdim=True).to(torch.int32).clamp_(max=max_tracing)
        for _ in range(max_tracing):
            running = ((tracing_idx < n_hit) & (transp > 1e-5))[:, 0].contiguous()
            n_run = int(running.sum())
            if n_run == 0:
                break
            if n_run * 10 > B * self.compact_below_tenths:   # (compaction costs ~0.1 ms of gathers)
                fg_pass(rays_o, rays_d, inter, tracing_blocks, tracing_idx, z_start, transp, dif, spec, depth, running, pd, ps, pa)
                continue
            # Not every ray runs (rays that miss every tile; later passes: most have left the tiles or are saturated): the pass runs
            # on the running rays alone -- every op
            # is per ray, so each running ray gets the values the full-size pass gives it; the others' colours are untouched either
            # way, and their tracing state (which the full-size pass advances) is never read again: a ray that stopped stays stopped
            # (the full-size pass costs its fills, slot lists and accumulation for every ray: ~4 ms at 1920x1080 for nothing)
            idx = running.nonzero()[:, 0]
            if lay == 2 and n_run % 32:
                idx = torch.cat([idx, idx[-1:].expand(32 - n_run % 32)])
            n = idx.shape[0]
            sub = [t[idx].contiguous() for t in (rays_o, rays_d, inter, tracing_blocks, tracing_idx, z_start, transp, dif, spec, depth)]
            shp_n = {0: lambda S, *tail: (n, S, *tail), 1: lambda S, *tail: (S, n, *tail), 2: lambda S, *tail: (n // 32, S, 32, *tail)}[lay]
            fg_pass(*sub, torch.ones(n, dtype=torch.bool, device=dev), torch.empty(shp_n(num_sample, 3), device=dev),
                    torch.empty(shp_n(num_sample, 3), device=dev), torch.empty(shp_n(num_sample, 1), device=dev))
            for full, part in zip((tracing_idx, z_start, transp, dif, spec, depth), sub[4:]):
                full[idx] = part   # (the padding repeats the last ray: equal values land on it twice)
        # blended backgrounds of the exit tile(s)
        bg_b = torch.full((B, 4), -1, dtype=torch.int16, device=dev)
        bg_w = torch.zeros(B, 4, device=dev)
        update_outgoing_bidx(rays_o, rays_d, self.block_corner, self.block_size, tracing_blocks, inter, bg_b, bg_w, 0.12, False)
        bg_w = bg_w / torch.sum(bg_w, dim=-1, keepdim=True)
        # rays whose foreground transmittance is EXACTLY zero (an opacity rounded to 1.0 on the way: hard surfaces) get no background:
        # it would be multiplied by that zero -- `dif + 0 * bgd` is `dif` bit for bit, colour and depth alike -- so nothing changes
        # but the decoder work of their 128 background samples (exact, unlike skip_saturated_background below)
        if self.skip_zero_transmittance_background:
            bg_b[(transp == 0)[:, 0]] = -1
        if skip_saturated_background:
            # rays the foreground has saturated get no background: it would enter the pixel with weight <= 1e-5 -- 400x below
            # one 8-bit step -- and on an opaque scene it is most of the frame's decoder work
            bg_b[(transp <= 1e-5)[:, 0]] = -1
        n_blend = int((bg_w > 0).sum(dim=-1).max().cpu())
        bgd, bgs, bgz = torch.zeros(B, 3, device=dev), torch.zeros(B, 3, device=dev), torch.zeros(B, 1, device=dev)
        if num_bg_sample != num_sample:
            pd = torch.empty(shp(num_bg_sample, 3), device=dev)
            ps = torch.empty(shp(num_bg_sample, 3), device=dev)
            pa = torch.empty(shp(num_bg_sample, 1), device=dev)
        for i in range(n_blend):
            zb = torch.full(shp(num_bg_sample), -1.0, device=dev)
            inverse_z_sampling(inter, bg_b[:, i].contiguous(), zb, sample_range, sample_major=sm)
            # (bg_pts_inference_v2 writes every sample: zeros where the ray has no background tile at this blend step)
            bg_pts_inference_v2(rays_o, rays_d, zb, bg_b, i, self.block_corner, self.block_size, self.resolution,
                                self.feature_tables, self.packed, pd, ps, pa, sample_major=sm)
            t1 = torch.ones(B, 1, device=dev)
            td, ts, tz = torch.zeros(B, 3, device=dev), torch.zeros(B, 3, device=dev), torch.zeros(B, 1, device=dev)
            accumulate_color(pd, ps, pa, t1, zb, td, ts, tz, sample_major=sm)
            w = torch.nan_to_num(bg_w[:, i:i + 1])  # rays with no exit tile: 0/0 in the reference's normalisation
            bgd += td * w
            bgs += ts * w
            bgz += tz * w
        out = (dif + transp * bgd, spec + transp * bgs, depth + transp * bgz, transp)
        return tuple(t[:n_rays] for t in out) if B != n_rays else out

    def render(self, H, W, K, c2w, **kw):
        o, d = self.compute_rays(H, W, K, c2w)
        dif, spec, depth, transp = self.render_rays(o, d, **kw)
        return dif.reshape(H, W, 3), spec.reshape(H, W, 3), depth.reshape(H, W, 1), transp.reshape(H, W, 1)
