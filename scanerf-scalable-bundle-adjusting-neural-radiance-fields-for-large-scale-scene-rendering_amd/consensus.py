"""ADMM camera consensus as a collective (RCCL all-reduce over xGMI on MI355X).

The reference funnels every tile's poses through a master process and Python
multiprocessing proxies (admm_trainer.py:124-179, tile.py:477-508).  Here every rank
scatters its tiles' {conf*se3 (6), conf (1), 1 (1)} rows into a dense zero [N_cam, 8] fp32
buffer at the global camera ids and one in-place all-reduce(SUM) replaces the master:
afterwards each rank computes the shared poses, the overlap set and the residuals locally
(identical on all ranks up to fp32 summation order).  Whenever a process group exists the
reduce IS the collective, also at world size 1 (so that a one-GPU box exercises RCCL:
tests/test_gpu_consensus.py, bench.py's `consensus_ms`); only without a process group does the
same code run without it.

Math (file:line in the reference):
  shared   = sum(conf*pose) / sum(conf), 0 -> 1 guard          admm_trainer.py:147-155
  overlap  = count >= 2                                         admm_trainer.py:153
  dual     = mean |shared_prev - shared|                        admm_trainer.py:157
  primal   = mean over tiles of mean |pose_t - shared[idx_t]|   admm_trainer.py:161-168
  delta   += 1.5 * (se3_refine - shared[idx])   (over-relaxed)  consensus.py:40-45
  penalty  = mean(rho * (se3 - shared + delta)^2 [overlap])     consensus.py:70-76
"""
import torch
import torch.distributed as dist


class ConsensusState:
    """ADMM state of ONE tile (consensus.py:18-21) plus the rank-local exchange buffer."""

    def __init__(self, num_camera_global, cam_idx, device, rho=0.0):
        self.n_cam = int(num_camera_global)
        self.cam_idx = cam_idx.to(device=device, dtype=torch.long)
        M = self.cam_idx.numel()
        self.device = device
        self.shared_se3 = torch.zeros(M, 6, device=device)
        self.delta_se3 = torch.zeros(M, 6, device=device)
        self.overlap_flags = torch.zeros(M, dtype=torch.bool, device=device)
        self.rho = torch.ones(6, device=device) * rho
        self.prev_shared = torch.zeros(self.n_cam, 6, device=device)
        self.buf = torch.zeros(self.n_cam, 8, device=device)
        self.dual_residual = None
        self.primal_residual = None

    def exchange(self, se3_refine, confidence=None):
        return exchange([self], [se3_refine], [confidence])

    def camera_loss(self, se3_refine):
        c = (se3_refine - self.shared_se3 + self.delta_se3) ** 2
        return torch.mean(self.rho[None, :] * c[self.overlap_flags])


def collective_active(group=None):
    """A process group exists AND this rank is a member of `group` (None = the default group): the exchanges go through its
    all-reduce (RCCL for CUDA tensors), whatever the world size.  A rank outside `group` takes the local path."""
    return dist.is_available() and dist.is_initialized() and dist.get_rank(group) >= 0


@torch.no_grad()
def exchange(states, se3_list, conf_list=None, group=None):
    """One consensus round for the tiles this rank owns (>=1; admm_trainer.py:74-83 maps tile t to
    rank t mod nGPU).  Returns (dual_residual, primal_residual) as 0-dim tensors."""
    s0 = states[0]
    buf = s0.buf
    buf.zero_()
    conf_list = conf_list or [None] * len(states)
    for st, se3, conf in zip(states, se3_list, conf_list):
        conf = torch.ones(st.cam_idx.numel(), device=st.device) if conf is None else conf
        rows = torch.cat([conf[:, None] * se3.detach(), conf[:, None], torch.ones_like(conf)[:, None]], 1)
        buf.index_add_(0, st.cam_idx, rows)
    ntiles = torch.tensor([float(len(states))], device=s0.device)
    if collective_active(group):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)  # RCCL over xGMI, in place, compute stream
    w = buf[:, 6:7]
    shared = buf[:, :6] / torch.where(w == 0, torch.ones_like(w), w)
    overlap = buf[:, 7] >= 2
    dual = torch.mean(torch.abs(s0.prev_shared - shared))
    primal = torch.zeros(1, device=s0.device)
    for st, se3 in zip(states, se3_list):
        primal += torch.mean(torch.abs(se3.detach() - shared[st.cam_idx]))
    if collective_active(group):
        pr = torch.cat([primal, ntiles])
        dist.all_reduce(pr, op=dist.ReduceOp.SUM, group=group)
        primal, ntiles = pr[:1], pr[1:]
    primal = (primal / ntiles)[0]
    for st, se3 in zip(states, se3_list):
        st.shared_se3 = shared[st.cam_idx].clone()
        st.delta_se3 = st.delta_se3 + 1.5 * (se3.detach() - st.shared_se3)
        st.overlap_flags |= overlap[st.cam_idx]
        st.prev_shared = shared
        st.dual_residual, st.primal_residual = dual, primal
    return dual, primal
