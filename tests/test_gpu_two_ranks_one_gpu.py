"""SURVEY.md 8(e) with what one GPU allows: the multi-tile ADMM driver as TWO RANKS (two processes, torch.distributed) with REAL tile
trainers on the fused kernels -- tile t on rank t mod 2 (admm_trainer.py:74-83), consensus all-reduce(SUM) and shared-depth
all-reduce(MIN) entered by both ranks once per stretch -- against the same two tiles driven by ONE process.  Both ranks share the
box's single MI355X, so the process group is gloo on CUDA tensors (RCCL refuses two ranks on one device; RCCL itself is exercised at
world size 1 in tests/test_gpu_consensus.py).  What this pins beyond the CPU gloo tests: rank-dependent tile ownership with real GPU
state, identical residual histories and consensus poses on both ranks, the same trajectory as the single-process run.

The two rank processes are children of the test itself (started and joined inside it; nothing happens at collection); they
rendezvous through a FileStore in the test's temporary directory."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"
H, W, S_, N_CAM = 24, 32, 32, 5
VIEWS = [[0, 1, 2], [2, 3, 4]]   # camera 2 is seen by both tiles
TOTAL, SYN = 8, 4


def _build_trainer(t):
    """Tile t with everything derived from seeds of its own (the same objects whichever process builds them)."""
    import scanerf_amd  # noqa
    from scanerf_amd import cameras as CM, consensus as C, trainer
    from scanerf_amd.tile_model import TileModel
    g = torch.Generator().manual_seed(100 + t)
    eye = torch.eye(3)
    all_c2w = torch.stack([torch.cat([eye, torch.tensor([[x0], [0.0], [-3.0]])], -1) for x0 in (-2.0, 0.0, 3.5, 6.0, 9.0)])
    all_ks = torch.tensor([[40.0, 0, W / 2, 0, 40.0, H / 2, 0, 0, 1]]).repeat(N_CAM, 1).reshape(N_CAM, 3, 3)
    vs = VIEWS[t]
    m = TileModel([-4.0 + 8.0 * t, -4, -4], [8, 8, 8], DEV, log2_T=13, seed=t)
    with torch.no_grad():
        m.features.mul_(100.0)
    cams = CM.CameraSet(all_ks[vs], all_c2w[vs], DEV, noise=torch.randn(3, 6, generator=g) * 0.01)
    locs = CM.pixel_locs(3, torch.arange(H * W), W, DEV)
    tgt = torch.rand(locs.shape[0], 3, generator=g).to(DEV)
    tr = trainer.TileTrainer(m, lambda s, locs=locs, tgt=tgt: (locs, tgt), total_step=20, num_sample=S_, adjust_step=1000, cameras=cams,
                             eta_cam=1e-3, consensus=C.ConsensusState(N_CAM, torch.tensor(vs), DEV, rho=1.0))
    tr.views = vs
    return tr


def _drive(trainers, group=None):
    from scanerf_amd import admm, cameras as CM, occlusion as OC
    shared_depth = torch.full((N_CAM, H // 2, W // 2), OC.NO_DEPTH, device=DEV)

    def rays_of(tr, v):
        ro, rd = tr.cameras.get_rays(CM.pixel_locs(3, torch.arange(H * W), W, DEV)[v * H * W:(v + 1) * H * W])
        return ro.detach(), rd.detach()

    def publish(tr):
        return OC.render_shared_depth(tr.model, lambda v: rays_of(tr, v), H, W, tr.views, torch.nonzero(tr.consensus.overlap_flags)[:, 0],
                                      shared_depth, S_fg=S_, S_bg=16, global_step=tr.global_step)

    def consume(tr):
        OC.update_occlusion_mask(tr.model, lambda v: rays_of(tr, v), H, W, tr.views, shared_depth, kernel_size=5)

    drv = admm.AdmmDriver(trainers, total_step=TOTAL, syn_iters=SYN, depth_hooks=(publish, consume, shared_depth), group=group)
    hist = drv.run()
    torch.cuda.synchronize()
    return {"hist": [list(map(float, h)) for h in hist], "shared": drv.shared_poses(N_CAM).cpu().tolist(),
            "se3": [tr.cameras.se3_refine.detach().cpu().tolist() for tr in trainers],
            "flags": [tr.consensus.overlap_flags.cpu().tolist() for tr in trainers],
            "depth_finite": torch.isfinite(shared_depth).reshape(N_CAM, -1).all(1).cpu().tolist(), "steps": [tr.global_step for tr in trainers]}


def _worker(rank, world, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", store=dist.FileStore(os.path.join(outdir, "rendezvous"), world), rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from scanerf_amd import admm
    mine = admm.tiles_of_rank(len(VIEWS), rank, world)
    res = _drive([_build_trainer(t) for t in mine])
    res["tiles"] = mine
    json.dump(res, open(os.path.join(outdir, f"rank{rank}.json"), "w"))
    dist.destroy_process_group()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--worker":
    _worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    sys.exit(0)


@pytest.mark.gpu
def test_two_ranks_with_real_tile_trainers_match_each_other_and_the_single_process_run():
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU device node")
    out = tempfile.mkdtemp(prefix="scanerf_two_ranks_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), "2", out], env=env,
                              stdout=open(os.path.join(out, f"rank{r}.log"), "w"), stderr=subprocess.STDOUT) for r in range(2)]
    try:
        for r, p in enumerate(procs):
            try:
                rc = p.wait(timeout=300)
            except subprocess.TimeoutExpired:
                raise AssertionError(f"rank {r} did not finish: " + open(os.path.join(out, f"rank{r}.log")).read()[-2000:])
            assert rc == 0, open(os.path.join(out, f"rank{r}.log")).read()[-3000:]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    ranks = [json.load(open(os.path.join(out, f"rank{r}.json"))) for r in range(2)]
    assert ranks[0]["tiles"] == [0] and ranks[1]["tiles"] == [1]
    # both ranks hold the same residual history and the same consensus poses (every rank derives them from the reduced buffer)
    assert ranks[0]["hist"] == ranks[1]["hist"] and len(ranks[0]["hist"]) == 3
    np.testing.assert_array_equal(np.array(ranks[0]["shared"]), np.array(ranks[1]["shared"]))
    assert ranks[0]["depth_finite"] == ranks[1]["depth_finite"] and ranks[0]["depth_finite"][2] and not ranks[0]["depth_finite"][0]
    for r, t in ((0, 0), (1, 1)):
        assert ranks[r]["flags"][0] == [v == 2 for v in VIEWS[t]] and ranks[r]["steps"] == [TOTAL]
    # ... and the trajectory of ONE process driving both tiles.  Since round 6 the per-camera sums of the ray adjoint are taken in a
    # fixed order (csrc/rays.hip), so every kernel of an iteration is bit-reproducible and the two runs execute the same
    # arithmetic on the same values: the trajectories are EQUAL, not close
    one = _drive([_build_trainer(0), _build_trainer(1)])
    np.testing.assert_array_equal(np.array(ranks[0]["hist"]), np.array(one["hist"]))
    np.testing.assert_array_equal(np.array(ranks[0]["shared"]), np.array(one["shared"]))
    for t in range(2):
        np.testing.assert_array_equal(np.array(ranks[t]["se3"][0]), np.array(one["se3"][t]))
    assert np.abs(np.array(one["shared"])).max() > 0 and np.isfinite(np.array(ranks[0]["hist"])).all()
