"""SURVEY.md 8 row a17 on the GPU: the ADMM camera consensus (admm_trainer.py:137-179, consensus.py:40-50,70-76) on CUDA
tensors against the oracle's restatement of the master loop and golden G8 -- VALUES, not flags -- first without a process
group, then through a world-size-1 `nccl` (= RCCL) group initialised in this process, so that both collectives of the
multi-GPU path (consensus all-reduce(SUM), shared-depth all-reduce(MIN)) run through RCCL on the one-GPU box."""
import numpy as np
import pytest
import torch
import torch.distributed as dist

from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N_CAM = 14


def _tiles():
    """3 tiles on one rank, unequal confidences; camera 5 is seen by all three, cameras 2 and 9 by two, camera 13 by none,
    camera 0 only by tile 0 with confidence 0 (the 0 -> 1 guard of admm_trainer.py:154)."""
    g = torch.Generator().manual_seed(3)
    idx = [[0, 1, 2, 5, 7], [2, 3, 5, 9, 10, 11], [4, 5, 6, 8, 9, 12]]
    tiles = []
    for t, ix in enumerate(idx):
        conf = torch.rand(len(ix), generator=g) * 3 + 0.1
        if t == 0:
            conf[0] = 0.0
        tiles.append({"idx": ix, "pose": torch.randn(len(ix), 6, generator=g) * 0.02, "confidence": conf})
    return tiles


def _check_round(C, tiles, states, prev_shared, deltas, rho):
    """One exchange on the GPU vs the oracle; returns (shared, new deltas) of the oracle for the next round."""
    shared, overlap, dual, primal = O.consensus_reduce(tiles, N_CAM, prev_shared)
    d, p = C.exchange(states, [t["pose"].to(DEV) for t in tiles], [t["confidence"].to(DEV) for t in tiles])
    np.testing.assert_allclose(float(d), float(dual), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(float(p), float(primal), rtol=1e-5, atol=1e-9)
    new_deltas = []
    for st, t, dl in zip(states, tiles, deltas):
        idx = torch.tensor(t["idx"])
        np.testing.assert_allclose(st.shared_se3.cpu().numpy(), shared[idx].numpy(), rtol=1e-5, atol=1e-8)
        want_delta = O.consensus_update(t["pose"], shared[idx], dl)
        np.testing.assert_allclose(st.delta_se3.cpu().numpy(), want_delta.numpy(), rtol=1e-5, atol=1e-8)
        new_deltas.append(want_delta)
        assert st.shared_se3.is_cuda and st.delta_se3.is_cuda
    return shared, overlap, new_deltas


def _run_two_rounds(C):
    tiles = _tiles()
    rho = 0.05
    states = [C.ConsensusState(N_CAM, torch.tensor(t["idx"]), DEV, rho=rho) for t in tiles]
    deltas = [torch.zeros(len(t["idx"]), 6) for t in tiles]
    shared, overlap, deltas = _check_round(C, tiles, states, None, deltas, rho)
    flags = [overlap[torch.tensor(t["idx"])].clone() for t in tiles]
    for st, f, t in zip(states, flags, tiles):
        assert torch.equal(st.overlap_flags.cpu(), f)
    assert bool(overlap[5]) and bool(overlap[2]) and bool(overlap[9]) and not bool(overlap[13]) and not bool(overlap[0])
    np.testing.assert_array_equal(shared[13].numpy(), np.zeros(6, np.float32))   # seen by nobody: 0 / 1
    np.testing.assert_array_equal(shared[0].numpy(), np.zeros(6, np.float32))    # confidence 0: 0 / 1 (admm_trainer.py:154)
    # the penalty every tile adds to its loss (consensus.py:70-76), value and gradient
    for st, t, dl, f in zip(states, tiles, deltas, flags):
        se3 = t["pose"].clone().requires_grad_(True)
        want = O.camera_loss(se3, shared[torch.tensor(t["idx"])], dl, f, torch.ones(6) * rho)
        want.backward()
        got_in = t["pose"].detach().clone().to(DEV).requires_grad_(True)
        got = st.camera_loss(got_in)
        got.backward()
        np.testing.assert_allclose(float(got.detach()), float(want.detach()), rtol=1e-5)
        np.testing.assert_allclose(got_in.grad.cpu().numpy(), se3.grad.numpy(), rtol=1e-5, atol=1e-10)
    # second round: the poses moved, the dual residual is taken against the first round's shared poses, the duals accumulate
    g = torch.Generator().manual_seed(4)
    for t in tiles:
        t["pose"] = t["pose"] + torch.randn(t["pose"].shape, generator=g) * 0.005
    _check_round(C, tiles, states, shared, deltas, rho)


def test_consensus_exchange_values_on_cuda_vs_oracle():
    import scanerf_amd  # noqa
    from scanerf_amd import consensus as C
    assert not dist.is_initialized()
    _run_two_rounds(C)


def test_consensus_state_on_cuda_reproduces_reference_golden_g8(golden):
    """ConsensusManager.update / camera_loss of the reference itself (golden G8) on CUDA tensors."""
    import scanerf_amd  # noqa
    from scanerf_amd import consensus as C
    g = golden("g8_consensus")
    M = g["se3_refine"].shape[0]
    t = lambda a: torch.from_numpy(a).to(DEV)
    st = C.ConsensusState(M, torch.arange(M), DEV, rho=0.05)
    st.delta_se3 = t(g["delta0"]).clone()
    st.shared_se3 = t(g["shared"])
    st.delta_se3 = st.delta_se3 + 1.5 * (t(g["se3_refine"]) - st.shared_se3)
    st.overlap_flags[t(g["overlap_idxs"])] = True
    np.testing.assert_allclose(st.delta_se3.cpu().numpy(), g["delta1"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(float(st.camera_loss(t(g["se3_refine"]))), float(g["loss"]), rtol=1e-5)


def _rccl_mapped():
    return any("librccl" in ln for ln in open("/proc/self/maps"))


def test_both_collectives_through_rccl_world1():
    """A real `nccl` process group of one rank, initialised in-process from a HashStore (no child process, no exec): the
    consensus all-reduce(SUM) and the shared-depth all-reduce(MIN) then run through RCCL, and give the values above."""
    import scanerf_amd  # noqa
    from scanerf_amd import consensus as C, occlusion as OC
    assert not dist.is_initialized()
    dist.init_process_group("nccl", store=dist.HashStore(), rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        assert C.collective_active() and dist.get_backend() == "nccl"
        _run_two_rounds(C)          # every exchange() above now calls dist.all_reduce on CUDA buffers
        torch.cuda.synchronize()
        assert _rccl_mapped(), "librccl is not mapped: the all-reduce did not go through RCCL"
        # shared-depth exchange (tile.py:436-475 -> occlusion.exchange_shared_depth): maps published this round arrive, others stay
        g = torch.Generator().manual_seed(0)
        shared_depth = torch.full((6, 12, 16), OC.NO_DEPTH, device=DEV)
        old = torch.rand(12, 16, generator=g).to(DEV) + 1
        shared_depth[4] = old                      # delivered by an earlier round, not re-published
        new1, new3 = torch.rand(12, 16, generator=g).to(DEV) + 2, torch.rand(12, 16, generator=g).to(DEV) + 3
        shared_depth[1], shared_depth[3] = new1, new3
        out = OC.exchange_shared_depth(shared_depth, published=[1, 3])
        torch.cuda.synchronize()
        assert torch.equal(out[1], new1) and torch.equal(out[3], new3) and torch.equal(out[4], old)
        assert torch.isinf(out[0]).all() and torch.isinf(out[2]).all() and torch.isinf(out[5]).all()
        # timing of one consensus exchange through RCCL (what bench.py reports as consensus_ms)
        st = C.ConsensusState(800, torch.arange(120), DEV)
        se3 = torch.randn(120, 6, device=DEV) * 1e-3
        for _ in range(3):
            st.exchange(se3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            st.exchange(se3)
        e1.record()
        torch.cuda.synchronize()
        print(f"consensus exchange through RCCL (world 1, 800 cameras): {e0.elapsed_time(e1) / 10:.3f} ms")
    finally:
        dist.destroy_process_group()
