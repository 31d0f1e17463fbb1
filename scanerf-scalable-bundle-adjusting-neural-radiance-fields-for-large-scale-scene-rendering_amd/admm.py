"""Multi-tile ADMM driver: the counterpart of ADMMTrainer (admm_trainer.py:61-337) as a collective program.

Reference: one multiprocessing.Process per GPU plus a master process; every SYN_ITERS iterations each tile commits its
poses to shared dictionaries (tile.py:477-508), the master averages them (admm_trainer.py:124-179) and the tiles
synchronise; tiles that share a GPU are swapped through host memory between their turns (tile.py:574-636).

Here: one process per GPU (torchrun), tile t on rank t mod world (admm_trainer.py:74-83), all of a rank's tiles resident in
HBM (288 GB: no swapping), and the two exchanges are collectives every rank takes part in with the same schedule:
consensus.exchange (all-reduce SUM of a [N_cam,8] buffer) and occlusion.exchange_shared_depth (all-reduce MIN).  The
"workers" are anything with the TileTrainer interface (train_one_step(), .cameras.se3_refine, .consensus), so the schedule is
testable on CPU with stand-ins (tests/test_admm_driver_cpu.py).
"""
import os

import torch
import torch.distributed as dist

from . import cameras as cam_mod
from . import consensus as cons
from . import formats


def tiles_of_rank(num_tiles, rank, world):
    """admm_trainer.py:74-83: tile t runs on GPU t mod nGPU."""
    return [t for t in range(num_tiles) if t % world == rank]


def syn_schedule(total_step, syn_start, syn_iters):
    """Lengths of the training stretches between exchanges (admm_trainer.py:233-262): an optional first stretch of
    SYN_START iterations, then SYN_ITERS each, until TOTAL_STEP iterations are spent (the last stretch is not shortened)."""
    steps = [s for s in (syn_start, syn_iters) if s > 0]
    out, left, i = [], total_step, 0
    while left > 0 and steps:
        out.append(steps[i])
        left -= steps[i]
        i = min(i + 1, len(steps) - 1)
    return out


class AdmmDriver:
    """Runs this rank's tiles through the ADMM schedule.

    trainers: this rank's TileTrainer objects (each with .cameras = CameraSet over its visible views and .consensus =
    ConsensusState over the same views' global camera ids).  Every rank must construct the driver with the same
    total_step / syn_start / syn_iters: the exchanges are collectives."""

    def __init__(self, trainers, total_step, syn_iters=100, syn_start=0, confidence=None, log_dir=None, group=None,
                 depth_hooks=None):
        self.trainers = list(trainers)
        self.stretches = syn_schedule(total_step, syn_start, syn_iters)
        self.confidence = confidence
        self.log_dir, self.group = log_dir, group
        # optional shared-depth exchange: (publish(trainer) -> camera ids it wrote, consume(trainer), shared_depth buffer).
        # The collective itself is issued HERE, exactly once per stretch on every rank whatever the number of tiles a rank
        # owns (inside a per-trainer hook, ranks with different tile counts would issue different numbers of collectives
        # and hang)
        self.depth_hooks = depth_hooks
        self.history = []

    def _rank(self):
        return dist.get_rank(self.group) if dist.is_available() and dist.is_initialized() else 0

    def synchronize(self):
        """commit -> reduce -> synchronize of the reference in one collective round; returns (dual, primal) residuals."""
        states = [t.consensus for t in self.trainers]
        se3s = [t.cameras.se3_refine.detach() for t in self.trainers]
        confs = None if self.confidence is None else [self.confidence(t) for t in self.trainers]
        if not states:  # a rank without tiles still has to take part in the collective
            raise RuntimeError("AdmmDriver: every rank needs at least one tile (tile t -> rank t mod world)")
        dual, primal = cons.exchange(states, se3s, confs, group=self.group)
        self.history.append((float(dual), float(primal)))
        if self.log_dir is not None and self._rank() == 0:
            with open(os.path.join(self.log_dir, "admm_error.txt"), "a") as f:  # admm_trainer.py:169-170
                f.write(f"primal_residual: {float(primal):.8f}\tdual_residual: {float(dual):.8f}\n")
        return dual, primal

    def run(self, on_stretch=None):
        self.synchronize()  # the reference exchanges once before the first iteration (admm_trainer.py:222-231)
        from . import occlusion
        for n, iters in enumerate(self.stretches):
            published = []
            for t in self.trainers:
                if self.depth_hooks is not None and n > 0:
                    self.depth_hooks[1](t)  # update_occlusion_mask with what the last exchange delivered
                t.admm = True
                for _ in range(iters):
                    t.maybe_prune()
                    t.train_one_step()
                if self.depth_hooks is not None:
                    published += list(self.depth_hooks[0](t) or [])  # render_shared_depth into the rank's buffer
            if self.depth_hooks is not None:
                occlusion.exchange_shared_depth(self.depth_hooks[2], published, group=self.group)
            res = self.synchronize()
            if on_stretch is not None:
                on_stretch(n, res)
        return self.history

    # ---- results --------------------------------------------------------------------------------------------------------
    def shared_poses(self, num_camera_global):
        """[N_cam,6] consensus poses as every rank holds them after the last exchange."""
        st = self.trainers[0].consensus
        return st.prev_shared[:num_camera_global]

    def write_refined_cameras(self, path, ks, ori_c2ws, H, W, num_camera_global=None):
        """admm_trainer.py:181-184: refined_rts = se3_to_SE3(shared) o ori_rts -> refined_camera.log (rank 0 writes)."""
        ori_c2ws = torch.as_tensor(ori_c2ws, dtype=torch.float32)[..., :3, :4]
        n = ori_c2ws.shape[0] if num_camera_global is None else num_camera_global
        shared = self.shared_poses(n).detach().cpu()
        rts = cam_mod.pose_compose([cam_mod.se3_to_SE3(shared), cam_mod.pose_invert(ori_c2ws)])
        c2ws = cam_mod.pose_invert(rts)
        if self._rank() == 0:
            formats.write_campara(path, torch.as_tensor(ks).reshape(n, 3, 3).numpy(), c2ws.numpy(), H, W)
        return c2ws
