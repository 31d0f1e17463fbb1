"""ADMM consensus: single-process math vs the oracle (and golden G8), and the all-reduce
formulation under a world_size-2 gloo group on CPU (the N>1 path of bench.py)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _tiles(seed=0, n_cam=30, n_tiles=4, M=12):
    g = torch.Generator().manual_seed(seed)
    tiles = []
    for t in range(n_tiles):
        idx = torch.sort(torch.randperm(n_cam, generator=g)[:M])[0]
        tiles.append({"idx": idx.tolist(), "pose": torch.randn(M, 6, generator=g) * 0.01,
                      "confidence": torch.rand(M, generator=g) + 0.5})
    return tiles


def test_exchange_matches_oracle_single_process():
    import scanerf_amd  # noqa
    from oracle import oracle as O
    from scanerf_amd import consensus as C
    tiles = _tiles()
    shared, overlap, dual, primal = O.consensus_reduce(tiles, 30)
    states = [C.ConsensusState(30, torch.tensor(t["idx"]), "cpu", rho=0.05) for t in tiles]
    d, p = C.exchange(states, [t["pose"] for t in tiles], [t["confidence"] for t in tiles])
    np.testing.assert_allclose(float(d), float(dual), rtol=1e-6)
    np.testing.assert_allclose(float(p), float(primal), rtol=1e-6)
    for st, t in zip(states, tiles):
        idx = torch.tensor(t["idx"])
        np.testing.assert_allclose(st.shared_se3.numpy(), shared[idx].numpy(), rtol=1e-6, atol=1e-9)
        assert torch.equal(st.overlap_flags, overlap[idx])
        np.testing.assert_allclose(st.delta_se3.numpy(), O.consensus_update(t["pose"], shared[idx], torch.zeros(12, 6)).numpy(),
                                   rtol=1e-6, atol=1e-9)


def test_update_and_penalty_match_reference_golden(golden):
    import scanerf_amd  # noqa
    from scanerf_amd import consensus as C
    g = golden("g8_consensus")
    M = g["se3_refine"].shape[0]
    st = C.ConsensusState(M, torch.arange(M), "cpu", rho=0.05)
    st.delta_se3 = torch.from_numpy(g["delta0"]).clone()
    # drive the state exactly as ConsensusManager.update does (consensus.py:40-50)
    st.shared_se3 = torch.from_numpy(g["shared"])
    st.delta_se3 = st.delta_se3 + 1.5 * (torch.from_numpy(g["se3_refine"]) - st.shared_se3)
    st.overlap_flags[torch.from_numpy(g["overlap_idxs"])] = True
    np.testing.assert_allclose(st.delta_se3.numpy(), g["delta1"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(float(st.camera_loss(torch.from_numpy(g["se3_refine"]))), float(g["loss"]), rtol=1e-6)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import scanerf_amd  # noqa
    from scanerf_amd import consensus as C
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tiles = _tiles()
    mine = [t for i, t in enumerate(tiles) if i % world == rank]  # tile t -> rank t mod nGPU
    states = [C.ConsensusState(30, torch.tensor(t["idx"]), "cpu") for t in mine]
    d, p = C.exchange(states, [t["pose"] for t in mine], [t["confidence"] for t in mine])
    q.put((rank, float(d), float(p), [s.shared_se3.numpy() for s in states], [s.overlap_flags.numpy() for s in states]))
    dist.destroy_process_group()


def test_allreduce_formulation_world2_gloo():
    from oracle import oracle as O
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=120) for _ in ps])
    [p.join(60) for p in ps]
    tiles = _tiles()
    shared, overlap, dual, primal = O.consensus_reduce(tiles, 30)
    for rank, d, p, sh, ov in res:
        np.testing.assert_allclose(d, float(dual), rtol=1e-5)
        np.testing.assert_allclose(p, float(primal), rtol=1e-5)
        mine = [t for i, t in enumerate(tiles) if i % 2 == rank]
        for t, a, b in zip(mine, sh, ov):
            idx = torch.tensor(t["idx"])
            np.testing.assert_allclose(a, shared[idx].numpy(), rtol=1e-5, atol=1e-8)
            assert np.array_equal(b, overlap[idx].numpy())
