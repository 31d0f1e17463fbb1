"""Training harness around the fused kernels: what TILE.create_optimizer / train / train_one_step (tile.py:296-332,
:818-877, :880-1015) and HashGrid.pruning_grid (hashgrid/__init__.py:138-225) do around the hot path -- learning-rate
schedules, 2x2 patch ray selection, the coarse-to-fine occupancy pruning schedule, checkpoints -- for one tile on one GPU.
Data loading, warp / monocular losses and image logging stay outside (out of scope).
"""
import math

import torch
import torch.nn.functional as F

from . import formats, network
from .hashgrid import HashEmbeddingBG
from .tile_model import train_step_fgbg, train_step_fused


# ---- scheduler.py:9-71 ---------------------------------------------------------------------------------------------
def decay_func1(step, decay_step, decay_rate):
    return decay_rate ** ((step / decay_step) ** 0.1)


def decay_func2(step, decay_step, decay_rate):
    return decay_rate ** (step / decay_step)


class Scheduler:
    """eta(step) = start_eta * decay_rate^(step / decay_steps) inside [start_itr, end_itr), 0 outside; decay_steps
    defaults to iterations / log_{decay_rate}(end_eta / start_eta), so eta(iterations) = end_eta (scheduler.py:15-52)."""

    def __init__(self, name, start_eta, end_eta, iterations, groups=(), decay_rate=0.1, decay_steps=None, start_itr=0,
                 end_itr=100000000, decay_func=2):
        self.decay_steps = iterations / math.log(end_eta / start_eta, decay_rate) if decay_steps is None else decay_steps
        self.decay_rate, self.start_eta, self.eta = decay_rate, start_eta, start_eta
        self.groups, self.name, self.start_itr, self.end_itr = list(groups), name, start_itr, end_itr
        self.decay_func = {1: decay_func1, 2: decay_func2}[decay_func]

    def value(self, global_step):
        if global_step < self.start_itr or global_step >= self.end_itr:
            return 0
        return self.start_eta * self.decay_func(global_step, self.decay_steps, self.decay_rate)

    def step(self, global_step, optimizer=None):
        self.eta = self.value(global_step)
        if optimizer is not None:
            groups = optimizer.param_groups if not self.groups else [optimizer.param_groups[i] for i in self.groups]
            for g in groups:
                g["lr"] = self.eta
        return self.eta


class SchedulerManager:
    def __init__(self, scheduler_list):
        self.scheduler_list = scheduler_list

    def getEta(self):
        return [s.eta for s in self.scheduler_list], [s.name for s in self.scheduler_list]

    def getInfo(self):
        return "".join("Eta %-10s\t%.8f\n" % (n, e) for e, n in zip(*self.getEta()))

    def step(self, global_step, optimizer=None):
        for s in self.scheduler_list:
            s.step(global_step, optimizer)


# ---- patch rays (tile.py:902-915, tools/utils.py:89-103) ------------------------------------------------------------
def get_ray_idx(idx, patch_size, H, W):
    """Pixel indices of patch_size x patch_size patches whose top-left pixels are idx (row-major in the patch)."""
    ar = torch.arange(patch_size, device=idx.device)
    offset = ar[None, :].repeat(patch_size, 1) + (ar * W)[:, None].repeat(1, patch_size)
    return (idx[:, None, None] + offset[None, ...]).reshape(-1)


def sample_patch_ray_idx(batch_size, num_camera, H, W, device, patch_size=2, generator=None):
    """The reference's per-iteration pixel choice: batch_size // num_camera rays per view as 2x2 patches whose corners
    come from two independent permutations of the columns and rows (the same pixels are used for every view)."""
    num_patch = (batch_size // num_camera) // (patch_size ** 2)
    px = torch.randperm(W - patch_size, device=device, generator=generator)[:num_patch]
    py = torch.randperm(H - patch_size, device=device, generator=generator)[:num_patch]
    return get_ray_idx(py * W + px, patch_size, H, W)


# ---- occupancy pruning (hashgrid/__init__.py:131-225) ---------------------------------------------------------------
def _mesh_grid(max_res, device):
    X, Y, Z = torch.meshgrid(torch.arange(0, int(max_res[0]), device=device), torch.arange(0, int(max_res[1]), device=device),
                             torch.arange(0, int(max_res[2]), device=device), indexing="ij")
    return torch.stack([X, Y, Z], -1).reshape(-1, 3)


@torch.no_grad()
def sigma_of_features(model, feats):
    """decoder.inference_sigma (network.py:168-170): softplus(sigma_layer(Spatial_MLP(x)[..., :32]))."""
    dec = model.decoder
    act = lambda u: torch.exp(u * u * -50.0)
    H = dec._lin("Spatial_MLP.mlp.2", act(dec._lin("Spatial_MLP.mlp.0", feats)))
    return F.softplus(dec._lin("sigma_layer.mlp.0", H[..., :32]))


@torch.no_grad()
def prune_occupancy(occupied_grid, log2dim, finest_resolution, encode, sigma_of, weight_feature32, global_step, sub_split, pruning_th,
                    batch_size, device):
    """The occupancy re-derivation of hashgrid/__init__.py:138-213 on arrays: every occupied cell (optionally split 2x per axis) is
    probed on a regular lattice of sample_resolution^3 points of the tile ([-1,1]^3 of the 2x box, contracted coordinates); it stays
    occupied when the largest alpha = 1 - exp(-sigma) over its lattice exceeds pruning_th.  encode(points [N,3]) -> features [N,32]
    (the HIP encoder binding), sigma_of(features * mask) -> sigma; finest_resolution: per-axis int vector (the table's finest level).
    -> (new grid bool, new log2dim int32)."""
    log2dim = log2dim.to(device) + (1 if sub_split else 0)
    grid_resolution = (2 ** log2dim).to(device)
    fin = torch.as_tensor(finest_resolution).to(device)
    total_res = fin / 4.0 if global_step < 10000 else fin / 2.0
    sample_resolution = ((total_res / 2.0) / grid_resolution).int()
    occ = occupied_grid
    if sub_split:
        occ = occ.repeat_interleave(2, 0).repeat_interleave(2, 1).repeat_interleave(2, 2)
    locs = torch.nonzero(occ).long()
    new_grid = torch.zeros(tuple(int(r) for r in grid_resolution), dtype=torch.bool, device=device)
    if locs.shape[0] and int(torch.prod(sample_resolution)) > 0:
        grid_corner = locs / grid_resolution
        grid_point = _mesh_grid(sample_resolution, device) / (sample_resolution * grid_resolution)
        run = max(int(batch_size / int(torch.prod(sample_resolution))), 1)
        wf = weight_feature32.reshape(1, 32)
        alpha_res = torch.zeros(locs.shape[0], device=device)
        for i in range(0, locs.shape[0], run):
            pts = (grid_corner[i:i + run, None, :] + grid_point[None, ...]) * 2 - 1
            n = pts.shape[0]
            alpha = 1 - torch.exp(-1.0 * sigma_of(encode(pts.reshape(-1, 3).float().contiguous()).reshape(-1, 32) * wf))
            alpha_res[i:i + n] = alpha.reshape(n, -1).max(dim=-1)[0]
        keep = locs[alpha_res > pruning_th]
        new_grid[keep[:, 0], keep[:, 1], keep[:, 2]] = True
    return new_grid.contiguous(), log2dim.int()


@torch.no_grad()
def pruning_tile_grid(model, global_step, sub_split=False, pruning_th=0.4, batch_size=92 ** 3, finest_resolution=2048):
    """HashGrid.pruning_tile_grid for a TileModel (prune_occupancy on its table through the HIP encoder binding)."""
    dev = model.device
    fin = torch.as_tensor(model.bbox_size / model.bbox_size.min() * finest_resolution).int()
    new_grid, log2dim = prune_occupancy(
        model.occupied_grid, model.log2dim, fin, lambda p: HashEmbeddingBG(p, model.features.detach(), model.resolution),
        lambda f: sigma_of_features(model, f), network.weight_feature(global_step, dev), global_step, sub_split, pruning_th, batch_size, dev)
    model.log2dim = log2dim
    model.occupied_grid = new_grid
    model._occ_full = bool(new_grid.all())
    return int(new_grid.sum())


@torch.no_grad()
def pruning_grid(model, global_step, log2dim, pruning_th, **kw):
    """hashgrid/__init__.py:215-225: split at most one level per call until the sampler reaches `log2dim` (the value for
    the tile's LONGEST axis, as cfg.TRAINING.GRID_LOG2DIM gives it)."""
    last = int(model.log2dim.max())
    assert log2dim >= last, f"log2dim {log2dim} last_sampler_log2dim {last}"
    return pruning_tile_grid(model, global_step, sub_split=log2dim > last, pruning_th=pruning_th, **kw)


# ---- the per-tile loop ------------------------------------------------------------------------------------------------
class TileTrainer:
    """One tile's optimisation state and iteration (tile.py:296-332, :866-877, :880-1015) on the fused kernels.

    get_batch(step) -> (rays_o [B,3], rays_d [B,3], target [B,3]) supplies the rays (the reference draws 2x2 patches
    per view: sample_patch_ray_idx).  The table is stepped by the fused sparse Adam kernel with the scheduled
    learning rate; the decoder by torch Adam with weight_decay 1e-6 (tile.py:308)."""

    def __init__(self, model, get_batch, total_step=40000, eta_hash=1e-2, eta_decoder=1e-3, grid_log2dim=(4, 5, 6, 7, 8, 9),
                 pruning_th=(0.1, 0.2, 0.3, 0.4), adjust_step=2000, dynamic_start=None, dynamic_end=None, dynamic_step=None,
                 num_sample=128, num_bg_sample=0, finest_resolution=2048, consensus=None, cameras=None, eta_cam=1e-3,
                 cam_start_step=0, admm=False):
        """cameras (cameras.CameraSet): pose refinement on -- get_batch(step) then returns (locs [B,3] int32 (view, px, py),
        target [B,3]) and the rays are generated from the current poses; se3_refine is the optimiser's second parameter
        group with its own schedule (tile.py:316-323).  admm: add the consensus penalty (consensus.py:70-76) to the
        pose gradient."""
        self.model, self.get_batch = model, get_batch
        self.cameras, self.admm = cameras, admm
        groups = [{"params": model.decoder.parameters(), "lr": eta_decoder, "weight_decay": 1e-6}]
        sches = [Scheduler("decoder", eta_decoder, 0.1 * eta_decoder, total_step, groups=[0])]
        if cameras is not None:
            groups.append({"params": [cameras.se3_refine], "lr": eta_cam})
            sches.append(Scheduler("cam", eta_cam, 0.1 * eta_cam, total_step, groups=[1], start_itr=cam_start_step, end_itr=total_step))
        self.dec_opt = torch.optim.Adam(groups)
        self.table_sche = SchedulerManager([Scheduler("featureGrid", eta_hash, 0.1 * eta_hash, total_step)])
        self.sche = SchedulerManager(sches)
        self.table_lr = eta_hash
        self.total_step, self.global_step = total_step, 0
        self.grid_log2dim, self.pruning_th, self.adjust_step = list(grid_log2dim), list(pruning_th), adjust_step
        self.dynamic_start = adjust_step if dynamic_start is None else dynamic_start
        self.dynamic_end = total_step if dynamic_end is None else dynamic_end
        self.dynamic_step = adjust_step if dynamic_step is None else dynamic_step
        self.num_sample, self.num_bg_sample, self.finest_resolution = num_sample, num_bg_sample, finest_resolution
        self.consensus = consensus

    def maybe_prune(self):
        """tile.py:866-877: every dynamic_step iterations inside [dynamic_start, dynamic_end]."""
        s = self.global_step
        if not (self.dynamic_start <= s <= self.dynamic_end and s % self.dynamic_step == 0):
            return None
        log2dim = self.grid_log2dim[min(s // self.adjust_step, len(self.grid_log2dim) - 1)]
        th = self.pruning_th[min(s // self.adjust_step, len(self.pruning_th) - 1)]
        log2dim = max(log2dim, int(self.model.log2dim.max()))
        return pruning_grid(self.model, s, log2dim, th, finest_resolution=self.finest_resolution)

    def _train_one_step_poses(self):
        """Iteration with pose refinement: rays from the cameras (HIP ray kernel, differentiable), fused render + adjoint with
        the ray gradients, which the ray kernel's backward reduces to dL/dC2W and torch carries to se3_refine."""
        locs, target = self.get_batch(self.global_step)
        cams = self.cameras
        cams.se3_refine.grad = None
        rays_o, rays_d = cams.get_rays(locs)
        if self.num_bg_sample > 0:   # the complete iteration (foreground + T_left * background), tile.py:639-692
            loss, g_o, g_d = train_step_fgbg(self.model, self.dec_opt, rays_o.detach(), rays_d.detach(), target, self.num_sample,
                                             self.num_bg_sample, self.global_step, table_lr=self.table_lr, pose_grads=True,
                                             dec_step=False)
        else:
            loss, g_o, g_d = train_step_fused(self.model, self.dec_opt, rays_o.detach(), rays_d.detach(), target, self.num_sample,
                                              self.global_step, table_lr=self.table_lr, pose_grads=True, dec_step=False)
        torch.autograd.backward([rays_o, rays_d], [g_o, g_d])
        if self.admm and self.consensus is not None and bool(self.consensus.overlap_flags.any()):
            self.consensus.camera_loss(cams.se3_refine).backward()
        self.dec_opt.step()
        return loss

    def train_one_step(self):
        if self.cameras is not None:
            loss = self._train_one_step_poses()
            self.table_sche.step(self.global_step)
            self.table_lr = self.table_sche.scheduler_list[0].eta
            self.sche.step(self.global_step, self.dec_opt)
            self.global_step += 1
            return loss
        rays_o, rays_d, target = self.get_batch(self.global_step)
        if self.num_bg_sample > 0:
            loss = train_step_fgbg(self.model, self.dec_opt, rays_o, rays_d, target, self.num_sample, self.num_bg_sample,
                                   self.global_step, table_lr=self.table_lr)
        else:
            loss = train_step_fused(self.model, self.dec_opt, rays_o, rays_d, target, self.num_sample, self.global_step,
                                    table_lr=self.table_lr)
        # the reference steps its schedulers AFTER the optimisers: the rate computed at step s is used at step s+1
        self.table_sche.step(self.global_step)
        self.table_lr = self.table_sche.scheduler_list[0].eta
        self.sche.step(self.global_step, self.dec_opt)
        self.global_step += 1
        return loss

    def train(self, steps, on_step=None):
        for _ in range(steps):
            self.maybe_prune()
            loss = self.train_one_step()
            if on_step is not None:
                on_step(self.global_step, loss)
        return loss

    def export_check_point(self, path):
        self.model.table_lr = self.table_lr
        return formats.export_check_point(path, self.model, self.consensus, self.dec_opt, self.global_step)

    def load_check_point(self, path):
        self.global_step = formats.load_check_point(path, self.model, self.consensus, self.dec_opt)
        self.table_sche.step(max(self.global_step - 1, 0))
        self.table_lr = self.table_sche.scheduler_list[0].eta
        return self.global_step
