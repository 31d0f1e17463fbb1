"""The multi-tile ADMM schedule (admm.py, counterpart of admm_trainer.py:61-337) on CPU with stand-in tile trainers: the
exchange schedule, the tile -> rank map, consensus pulling overlapping cameras together, identical results with one process
and with a world_size-2 gloo group, and the refined camera log."""
import os
import socket
import sys
import types

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_CAM, N_TILES, M = 14, 4, 6


def _tile_views(t):  # consecutive tiles share 2 of their 6 cameras
    return [(4 * t + k) % N_CAM for k in range(M)]


class _ToyTrainer:
    """Stands in for TileTrainer: pulls its cameras' se3_refine towards a tile-specific target by gradient descent, plus the
    ADMM penalty once the driver switched it on."""

    def __init__(self, t):
        import scanerf_amd  # noqa
        from scanerf_amd import consensus as C
        g = torch.Generator().manual_seed(100 + t)
        self.cameras = types.SimpleNamespace(se3_refine=torch.nn.Parameter(torch.zeros(M, 6)))
        self.consensus = C.ConsensusState(N_CAM, torch.tensor(_tile_views(t)), "cpu", rho=5.0)
        self.target = torch.randn(M, 6, generator=g) * 0.01
        self.admm = False
        self.steps = 0
        self.prunes = 0

    def maybe_prune(self):
        self.prunes += 1

    def train_one_step(self):
        p = self.cameras.se3_refine
        loss = ((p - self.target) ** 2).mean()
        if self.admm and bool(self.consensus.overlap_flags.any()):
            loss = loss + self.consensus.camera_loss(p)
        g, = torch.autograd.grad(loss, p)
        with torch.no_grad():
            p -= 2.0 * g
        self.steps += 1


def _run(rank, world):
    import scanerf_amd  # noqa
    from scanerf_amd import admm
    mine = admm.tiles_of_rank(N_TILES, rank, world)
    trainers = [_ToyTrainer(t) for t in mine]
    drv = admm.AdmmDriver(trainers, total_step=50, syn_iters=10, syn_start=5)
    hist = drv.run()
    return mine, trainers, drv, hist


def test_schedule_and_tile_map():
    import scanerf_amd  # noqa
    from scanerf_amd import admm
    assert admm.tiles_of_rank(32, 3, 8) == [3, 11, 19, 27]
    assert admm.syn_schedule(40000, 0, 100) == [100] * 400
    assert admm.syn_schedule(50, 5, 10) == [5, 10, 10, 10, 10, 10]  # the last stretch is not shortened (admm_trainer.py:233-262)
    assert admm.syn_schedule(30, 0, 0) == []


def test_single_process_consensus_converges(tmp_path):
    import scanerf_amd  # noqa
    from scanerf_amd import cameras as CM
    from scanerf_amd import formats
    mine, trainers, drv, hist = _run(0, 1)
    assert mine == [0, 1, 2, 3] and len(hist) == 1 + 6 and all(t.steps == 55 and t.prunes == 55 for t in trainers)
    # cameras seen by two tiles end up closer to each other than their private targets are
    before, after = [], []
    for a in range(N_TILES):
        b = (a + 1) % N_TILES
        va, vb = _tile_views(a), _tile_views(b)
        for cam in set(va) & set(vb):
            ia, ib = va.index(cam), vb.index(cam)
            before.append(float((trainers[a].target[ia] - trainers[b].target[ib]).abs().mean()))
            after.append(float((trainers[a].cameras.se3_refine[ia] - trainers[b].cameras.se3_refine[ib]).detach().abs().mean()))
    assert len(before) >= 8 and np.mean(after) < 0.5 * np.mean(before), (np.mean(before), np.mean(after))
    assert hist[-1][1] < hist[1][1]  # primal residual shrinks
    # refined camera log: shared poses applied on top of the original cameras (admm_trainer.py:181-184)
    ori = torch.cat([torch.eye(3).repeat(N_CAM, 1, 1), torch.arange(N_CAM * 3.0).reshape(N_CAM, 3, 1)], -1)
    ks = torch.tensor([[100.0, 0, 32, 0, 100, 24, 0, 0, 1]]).repeat(N_CAM, 1)
    c2ws = drv.write_refined_cameras(tmp_path / "refined_camera.log", ks, ori, 48, 64)
    Ks, C2Ws, H, W = formats.read_campara(tmp_path / "refined_camera.log", return_shape=True)
    assert (H, W) == (48, 64) and C2Ws.shape == (N_CAM, 3, 4)
    np.testing.assert_allclose(C2Ws, c2ws.numpy(), atol=1e-7)
    want = CM.pose_invert(CM.pose_compose([CM.se3_to_SE3(drv.shared_poses(N_CAM)), CM.pose_invert(ori)]))
    np.testing.assert_allclose(c2ws.numpy(), want.detach().numpy(), atol=1e-7)
    assert float((c2ws - ori).abs().max()) > 1e-4  # the consensus moved them


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine, trainers, drv, hist = _run(rank, world)
    q.put((rank, mine, hist, [t.cameras.se3_refine.detach().numpy() for t in trainers], drv.shared_poses(N_CAM).numpy()))
    dist.destroy_process_group()


def test_world2_gloo_matches_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=180) for _ in ps], key=lambda t: t[0])
    [p.join(60) for p in ps]
    _, trainers, drv, hist = _run(0, 1)
    for rank, mine, h, se3s, shared in res:
        assert mine == [t for t in range(N_TILES) if t % 2 == rank]
        np.testing.assert_allclose(np.array(h), np.array(hist), rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(shared, drv.shared_poses(N_CAM).numpy(), rtol=1e-4, atol=1e-8)
        for t, got in zip(mine, se3s):
            np.testing.assert_allclose(got, trainers[t].cameras.se3_refine.detach().numpy(), rtol=1e-4, atol=1e-8)


def _depth_driver_worker(rank, world, port, q):
    """3 tiles on 2 ranks (rank 0 owns tiles 0 and 2, rank 1 owns tile 1): with the shared-depth collective inside a
    per-trainer hook the ranks would issue 2 vs 1 collectives per stretch and hang; the driver issues exactly one."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import scanerf_amd  # noqa
    from scanerf_amd import admm
    from scanerf_amd import occlusion as OC
    mine = admm.tiles_of_rank(3, rank, world)
    trainers = [_ToyTrainer(t) for t in mine]
    for t, tr in zip(mine, trainers):
        tr.tile = t
    shared = torch.full((N_CAM, 2, 3), OC.NO_DEPTH)
    rounds = {"n": 0}
    seen = []

    def publish(tr):  # tile t publishes camera t's map; its depth grows from one stretch to the next
        shared[tr.tile] = 10.0 * tr.tile + tr.steps
        return [tr.tile]

    def consume(tr):
        seen.append((tr.tile, shared[:3, 0, 0].tolist()))

    drv = admm.AdmmDriver(trainers, total_step=20, syn_iters=10, depth_hooks=(publish, consume, shared))
    drv.run()
    q.put((rank, mine, seen, shared[:3, 0, 0].tolist()))
    dist.destroy_process_group()


def test_world2_shared_depth_exchange_once_per_stretch_with_unequal_tile_counts():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_depth_driver_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=180) for _ in ps], key=lambda t: t[0])
    [p.join(60) for p in ps]
    assert [r[1] for r in res] == [[0, 2], [1]]
    # after stretch 2 every rank holds every tile's LATEST map (steps = 20): replaced, not min'ed with the first round's
    for _, _, seen, final in res:
        assert final == [20.0, 30.0, 40.0]
        # what the tiles consumed at the start of stretch 2 = the first round's maps (steps = 10), identical on both ranks
        # (a rank's later tiles also see what its earlier tiles have just re-published locally)
        assert seen[0][1] == [10.0, 20.0, 30.0]
        assert all(vals[1:] == [20.0, 30.0] for _, vals in seen)
