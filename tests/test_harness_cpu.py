"""CPU-side checks of the harness around the hot path: learning-rate scheduler against the
reference's golden G9, patch ray indices, on-disk formats (round trips and literal layouts the reference's parsers
expect), the voxelize oracle's known answers, and the shared-depth exchange under a world_size-2 gloo group."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_scheduler_matches_reference_golden_g9(golden):
    import scanerf_amd  # noqa
    from oracle import oracle as O
    from scanerf_amd.trainer import Scheduler, SchedulerManager
    g = golden("g9_scheduler")
    sch = Scheduler("grid", float(g["start"]), float(g["end"]), int(g["iters"]))
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.0)
    for s, want in zip(g["steps"], g["eta"]):
        sch.step(int(s), opt)
        assert abs(opt.param_groups[0]["lr"] - want) <= 1e-12 * abs(want) + 1e-18
        assert abs(O.scheduler_eta(int(s), float(g["start"]), float(g["end"]), int(g["iters"])) - want) <= 1e-12 * abs(want)
    # window and group selection (scheduler.py:38-52)
    opt2 = torch.optim.SGD([{"params": [torch.nn.Parameter(torch.zeros(1))]}, {"params": [torch.nn.Parameter(torch.zeros(1))]}], lr=7.0)
    cam = Scheduler("cam", 1e-3, 1e-4, 1000, groups=[1], start_itr=100, end_itr=1000)
    mgr = SchedulerManager([cam])
    mgr.step(50, opt2)
    assert opt2.param_groups[1]["lr"] == 0 and opt2.param_groups[0]["lr"] == 7.0
    mgr.step(500, opt2)
    assert abs(opt2.param_groups[1]["lr"] - 1e-3 * 0.1 ** 0.5) < 1e-12
    mgr.step(1000, opt2)
    assert opt2.param_groups[1]["lr"] == 0
    assert "cam" in mgr.getInfo()


def test_patch_ray_indices():
    import scanerf_amd  # noqa
    from scanerf_amd.trainer import get_ray_idx, sample_patch_ray_idx
    idx = get_ray_idx(torch.tensor([0, 13]), 2, 6, 10)
    assert idx.tolist() == [0, 1, 10, 11, 13, 14, 23, 24]  # tools/utils.py:89-103: row-major inside each patch
    g = torch.Generator().manual_seed(0)
    r = sample_patch_ray_idx(2048, 8, 120, 160, "cpu", generator=g)
    assert r.shape[0] == (2048 // 8 // 4) * 4 and int(r.max()) < 120 * 160 and int(r.min()) >= 0
    px = r.reshape(-1, 4)
    assert torch.equal(px[:, 1] - px[:, 0], torch.ones_like(px[:, 0])) and torch.equal(px[:, 2] - px[:, 0], torch.full_like(px[:, 0], 160))


def test_camera_log_roundtrip_and_layout(tmp_path):
    import scanerf_amd  # noqa
    from scanerf_amd import formats as F
    rng = np.random.default_rng(0)
    ks = np.tile(np.array([[1234.5678, 0, 640.25], [0, 1233.1, 360.5], [0, 0, 1]], np.float32), (3, 1, 1))
    c2ws = rng.normal(size=(3, 3, 4)).astype(np.float32)
    p = tmp_path / "refined_camera.log"
    F.write_campara(p, ks, c2ws, 720, 1280)
    lines = open(p).read().split("\n")
    assert lines[0] == "0" and lines[7] == "1" and lines[2] == "1280 720 0 1000" and lines[6] == "0 0 0 1"
    assert lines[1] == f"{ks[0, 0, 0]:.2f} {ks[0, 1, 1]:.2f} {ks[0, 0, 2]} {ks[0, 1, 2]}"
    Ks, C2Ws, H, W = F.read_campara(p, return_shape=True)
    assert (H, W) == (720, 1280) and Ks.shape == (3, 3, 3) and C2Ws.shape == (3, 3, 4)
    np.testing.assert_allclose(C2Ws, c2ws, atol=6e-9 + 1e-8)
    np.testing.assert_allclose(Ks[:, 0, 0], np.round(ks[:, 0, 0].astype(np.float64), 2), atol=1e-3)
    assert Ks[0, 0, 2] == np.float32(640.25) and Ks[0, 1, 2] == np.float32(360.5)


def test_tile_tables_roundtrip(tmp_path):
    import scanerf_amd  # noqa
    from scanerf_amd import formats as F
    corners = np.array([[-4, -4, -4], [4, -4, -4.5]], np.float64)
    F.write_tile_info(tmp_path / "tile_info.txt", corners, [8, 8, 8], [32, 8192], flags=[0, 1])
    txt = open(tmp_path / "tile_info.txt").read().split("\n")
    assert txt[0] == "# TILEID(1) BBOX_CORNER(3) BBOX_SIZE(3) RESOLUTION(2) FLAG(1)"
    assert txt[2] == "1 4.00 -4.00 -4.50 8.00 8.00 8.00 32 8192 1"
    t = F.read_tile_info(tmp_path / "tile_info.txt", 1)
    assert t["corner"] == [4.0, -4.0, -4.5] and t["size"] == [8.0, 8.0, 8.0] and t["resolution"] == [32, 8192] and t["init_outside"]
    assert len(F.read_tile_info(tmp_path / "tile_info.txt")) == 2
    F.write_training_views(tmp_path / "training_views.txt", [[3, 1, 4], [1, 5, 9, 2, 6]])
    assert open(tmp_path / "training_views.txt").read() == "0\n3 1 4\n1\n1 5 9 2 6\n"
    assert F.read_training_views(tmp_path / "training_views.txt", 1) == [1, 5, 9, 2, 6]
    F.write_cams(tmp_path / "cams.npz", np.zeros((2, 3, 4)), np.ones((2, 3, 3)), [7, 9])
    c2ws, ks, idxs = F.read_cams(tmp_path / "cams.npz")
    assert c2ws.shape == (2, 3, 4) and ks.shape == (2, 3, 3) and idxs.tolist() == [7, 9]


def _cube_mesh(center, half):
    c, h = np.asarray(center, np.float32), np.float32(half)
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32) * h + c
    f = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4],
                  [1, 5, 7], [1, 7, 3]], np.int32)
    return v, f


def test_ply_roundtrip_ascii_and_binary(tmp_path):
    import scanerf_amd  # noqa
    from scanerf_amd import formats as F
    v, f = _cube_mesh([0.5, -1, 2], 1.25)
    for binary in (True, False):
        p = tmp_path / f"m{int(binary)}.ply"
        F.write_ply(p, v, f, binary=binary)
        v2, f2 = F.read_ply(p)
        assert v2.dtype == np.float32 and f2.dtype == np.int32
        np.testing.assert_array_equal(v2, v)
        np.testing.assert_array_equal(f2, f)
    # extra vertex properties and an alternate list name, as meshing tools write them
    with open(tmp_path / "extra.ply", "w") as fh:
        fh.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                 "property uchar red\nelement face 1\nproperty list uchar uint vertex_index\nend_header\n"
                 "0 0 0 255\n1 0 0 255\n0 1 0 255\n3 0 1 2\n")
    v3, f3 = F.read_ply(tmp_path / "extra.ply")
    assert v3.tolist() == [[0, 0, 0], [1, 0, 0], [0, 1, 0]] and f3.tolist() == [[0, 1, 2]]


def test_voxelize_oracle_known_answers():
    """cuda/include/voxelize.h:12-119 by hand: grid 8^3 over [0,8)^3 (cell = 1).  A triangle spanning x,y in [2.2,3.8],
    z = 5.5 has the box [2.2,3.8]^2 x {5.5}; inflated 1.5x about its centre (3,3,5.5): [1.8,4.2]^2 x [5.5,5.5] ->
    cells x,y in 1..4, z = 5.  init_out marks every cell whose centre is outside that union box."""
    from oracle import oracle as O
    v = np.array([[2.2, 2.2, 5.5], [3.8, 2.2, 5.5], [2.2, 3.8, 5.5], [100, 100, 100], [101, 100, 100], [100, 101, 100]], np.float32)
    f = np.array([[0, 1, 2], [3, 4, 5]], np.int32)  # the second face misses the grid entirely
    vis, out = O.voxelize_mesh(v, f, [3, 3, 3], [0, 0, 0], [8, 8, 8], False)
    want = np.zeros((8, 8, 8), bool)
    want[1:5, 1:5, 5] = True
    assert np.array_equal(vis, want) and not out.any()
    vis, out = O.voxelize_mesh(v, f, [3, 3, 3], [0, 0, 0], [8, 8, 8], True)
    cx = np.arange(8) + 0.5
    inside = ((cx > 1.8) & (cx < 4.2))
    exp_out = ~(inside[:, None, None] & inside[None, :, None] & (cx == 5.5)[None, None, :])  # z: only the centre of cell 5
    assert np.array_equal(out, exp_out) and np.array_equal(vis, want | exp_out)
    # an empty face list with init_out: the union box is empty, everything is outside
    vis, out = O.voxelize_mesh(v, np.zeros((0, 3), np.int32), [2, 3, 1], [0, 0, 0], [4, 8, 2], True)
    assert vis.all() and out.all() and vis.shape == (4, 8, 2)


def test_checkpoint_roundtrip_cpu(tmp_path):
    """Checkpoint keys follow tile.py:541-569 / hashgrid/__init__.py:94-107 / consensus.py:25-38; a second model restored
    from the file holds the same table, decoder, occupancy, Adam moments and ADMM state."""
    import scanerf_amd  # noqa
    from scanerf_amd import consensus as C
    from scanerf_amd import formats as F
    from scanerf_amd.tile_model import TileModel
    m = TileModel([-4, -4, -4], [8, 8, 8], "cpu", log2_T=10, seed=3)
    m.exp_avg.normal_()
    m.exp_avg_sq.uniform_()
    m.adam_step = 17
    m.set_occupancy(torch.rand(16, 16, 16) > 0.5)
    cs = C.ConsensusState(20, torch.arange(5), "cpu", rho=0.05)
    cs.delta_se3.normal_()
    opt = torch.optim.Adam([{"params": m.decoder.parameters(), "lr": 1e-3, "weight_decay": 1e-6}])
    m.decoder.params.grad = torch.randn_like(m.decoder.params)
    opt.step()
    p = F.export_check_point(tmp_path / "checkpoint-123-0.pt", m, cs, opt, 123)
    raw = torch.load(p, map_location="cpu", weights_only=False)
    assert set(raw) == {"global_step", "hashgrid", "admm", "decoder", "featureGrid_optimizer", "optimizer"}
    assert set(raw["hashgrid"]) == {"occupied_grid", "sampler_log2dim", "grid_resolution", "features"}
    assert set(raw["admm"]) == {"shared_se3", "delta_se3", "overlap_flags", "rho"}
    assert "Spatial_MLP.mlp.0.weight" in raw["decoder"] and raw["decoder"]["Directional_MLP.mlp.4.weight"].shape == (3, 64)
    m2 = TileModel([-4, -4, -4], [8, 8, 8], "cpu", log2_T=10, seed=9)
    cs2 = C.ConsensusState(20, torch.arange(5), "cpu")
    opt2 = torch.optim.Adam([{"params": m2.decoder.parameters(), "lr": 1e-3, "weight_decay": 1e-6}])
    assert F.load_check_point(p, m2, cs2, opt2) == 123
    assert torch.equal(m2.features, m.features) and torch.equal(m2.decoder.params, m.decoder.params)
    assert torch.equal(m2.occupied_grid, m.occupied_grid) and not m2._occ_full and m2.adam_step == 17
    assert torch.equal(m2.exp_avg, m.exp_avg) and torch.equal(m2.exp_avg_sq, m.exp_avg_sq)
    assert torch.equal(cs2.delta_se3, cs.delta_se3) and torch.equal(cs2.rho, cs.rho)
    assert torch.equal(opt2.state_dict()["state"][0]["exp_avg"], opt.state_dict()["state"][0]["exp_avg"])


def _depth_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import scanerf_amd  # noqa
    from scanerf_amd import occlusion as OC
    dist.init_process_group("gloo", rank=rank, world_size=world)
    buf = torch.full((6, 4, 5), OC.NO_DEPTH)
    # round 1: camera c is published by rank c % 3 if that rank exists; camera 5 by nobody
    mine = [cam for cam in range(5) if cam % 3 == rank]
    for cam in mine:
        buf[cam] = torch.arange(20.0).reshape(4, 5) + 100 * cam
    OC.exchange_shared_depth(buf, mine)
    first = buf.clone()
    # round 2: the publishers re-render and every depth INCREASES (a surface receded), except camera 0 whose publisher
    # (rank 0) publishes nothing this round: its entry must survive.  A MIN over the persistent buffers would keep the old,
    # smaller values.
    again = [cam for cam in mine if cam != 0]
    for cam in again:
        buf[cam] = torch.arange(20.0).reshape(4, 5) + 100 * cam + 7.0
    OC.exchange_shared_depth(buf, again)
    q.put((rank, first.numpy(), buf.numpy()))
    dist.destroy_process_group()


def test_shared_depth_exchange_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_depth_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    [p.join(60) for p in ps]
    base = np.arange(20.0).reshape(4, 5)
    for _, first, second in res:
        for cam in range(6):
            if cam < 5 and cam % 3 < 2:
                np.testing.assert_array_equal(first[cam], base + 100 * cam)
                # round 2 REPLACES what was republished (depths went up by 7) and keeps camera 0, which nobody republished
                np.testing.assert_array_equal(second[cam], base + 100 * cam + (0.0 if cam == 0 else 7.0))
            else:
                assert np.isinf(first[cam]).all() and np.isinf(second[cam]).all()
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][2], res[1][2])


def test_camera_algebra_matches_reference_golden_g7(golden):
    """cameras.py Lie / pose algebra against camera.py (golden G7): se3_to_SE3, invert, compose."""
    import scanerf_amd  # noqa
    from scanerf_amd import cameras as CM
    g = golden("g7_camera")
    SE3 = CM.se3_to_SE3(torch.from_numpy(g["se3"]))
    np.testing.assert_allclose(SE3.numpy(), g["SE3"], rtol=1e-6, atol=1e-7)
    w2c = CM.pose_invert(torch.from_numpy(g["c2w"]))
    np.testing.assert_allclose(w2c.numpy(), g["w2c"], rtol=1e-6, atol=1e-7)
    comp = CM.pose_compose([SE3, w2c])
    np.testing.assert_allclose(comp.numpy(), g["composed"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(CM.pose_invert(comp).numpy(), g["composed_inv"], rtol=1e-6, atol=1e-7)
    # small-angle limit and the pixel layout of a batch
    np.testing.assert_allclose(CM.se3_to_SE3(torch.zeros(2, 6)).numpy(), np.tile(np.eye(3, 4, dtype=np.float32), (2, 1, 1)))
    locs = CM.pixel_locs(3, torch.tensor([0, 9, 17]), 8, "cpu")
    assert locs.dtype == torch.int32 and locs.tolist()[:4] == [[0, 0, 0], [0, 1, 1], [0, 1, 2], [1, 0, 0]]


def test_metrics_match_reference_golden_g11(golden, tmp_path):
    """metrics.ssim / psnr against the reference's tools/ssim.py and tools/utils.py (golden G11); evaluation log layout."""
    import scanerf_amd  # noqa
    from scanerf_amd import metrics as M
    g = golden("g11_ssim")
    a, b, sm = (torch.from_numpy(g[k]) for k in ("a", "b", "smooth"))
    np.testing.assert_allclose(float(M.ssim(a, b)), float(g["ssim_ab"]), rtol=2e-5)
    np.testing.assert_allclose(float(M.ssim(a, a)), 1.0, rtol=1e-6)
    np.testing.assert_allclose(float(M.ssim(a, sm)), float(g["ssim_asmooth"]), rtol=1e-4)
    np.testing.assert_allclose(M.ssim(a, b, size_average=False).numpy(), g["ssim_ab_per_image"], rtol=2e-5)
    np.testing.assert_allclose(M.psnr(a[0].permute(1, 2, 0) * 255.0, b[0].permute(1, 2, 0) * 255.0), float(g["psnr_ab0"]), rtol=1e-6)
    views = [(None, None, (a[i].permute(1, 2, 0) * 255.0)) for i in range(2)]
    order = iter(range(2))
    mp, ms, rows = M.evaluate_views(lambda H, W, K, c2w: b[next(order)].permute(1, 2, 0), views, tmp_path / "eval.txt")
    txt = open(tmp_path / "eval.txt").read().split("\n")
    assert txt[0].startswith("img 0 psnr ") and "\tssim " in txt[0] and txt[2].startswith("mean psnr ")
    np.testing.assert_allclose(rows[0][1], float(g["psnr_ab0"]), rtol=1e-5)
    np.testing.assert_allclose(ms, float(g["ssim_ab"]), rtol=1e-3)


def test_tile_export_read_from_the_reference_writers_files_golden_g12(golden, tmp_path):
    """f3: tests/golden/g12_tile/ was written by the REFERENCE's own code (HashGrid.export, hashgrid/__init__.py:248-257,
    and tile.py:521's torch.save of the decoder state dict; tests/golden/make_golden_formats.py).  renderer.load_tile must
    give what the reference's consumer makes of those files (rendering.py:101-112 blob via tools.utils.extract_MLP_para,
    :164-165 box rule -- both captured in g12_expected.npz), and this repo's writer must produce the same file layout."""
    import scanerf_amd  # noqa
    from scanerf_amd import renderer as R
    exp = golden("g12_expected")
    t = R.load_tile(os.path.join(ROOT, "tests", "golden", "g12_tile"))
    assert t["blob"].dtype == np.float32 and np.array_equal(t["blob"], exp["blob"])          # layer order, [bias, W^T]
    assert t["features"].dtype == np.float16 and t["features"].shape == (16, 64, 2)
    assert np.array_equal(t["features"], exp["features_f32"].astype(np.float16))
    assert t["occupied_grid"].dtype == np.bool_ and np.array_equal(t["occupied_grid"], exp["occupied_grid"])
    assert t["resolution"].dtype == np.int32 and np.array_equal(t["resolution"], exp["resolution"])
    assert t["grid_log2dim"].dtype == np.int32 and np.array_equal(t["grid_log2dim"], exp["grid_log2dim"])
    # the file stores the 2x box; the renderer's box rule gives back the tile itself
    c, z = R.render_box(torch.from_numpy(t["block_corner"]), torch.from_numpy(t["block_size"]))
    assert np.array_equal(c.numpy(), exp["render_block_corner"]) and np.array_equal(z.numpy(), exp["render_block_size"])
    assert np.allclose(c.numpy(), exp["tile_corner"]) and np.allclose(z.numpy(), exp["tile_size"])
    # this repo's writer: same keys, dtypes and shapes as the reference's file, and a model restored from the
    # reference's file exports the same arrays
    from scanerf_amd.tile_model import TileModel
    m = TileModel(exp["tile_corner"].tolist(), exp["tile_size"].tolist(), "cpu", log2_T=6, seed=1, grid_resolution=(4, 64), sampler_log2dim=3)
    assert np.array_equal(m.log2dim.cpu().numpy(), exp["grid_log2dim"])
    with torch.no_grad():
        m.features.copy_(torch.from_numpy(exp["features_f32"]))
        m.decoder.params.copy_(torch.from_numpy(exp["blob"]))
    m.set_occupancy(torch.from_numpy(exp["occupied_grid"]))
    assert np.array_equal(m.resolution.cpu().numpy(), exp["resolution"])
    R.export_tile(str(tmp_path / "mine"), m)
    ref_f, my_f = np.load(os.path.join(ROOT, "tests", "golden", "g12_tile", "feature.npz")), np.load(tmp_path / "mine" / "feature.npz")
    assert sorted(ref_f.files) == sorted(my_f.files)
    for k in ref_f.files:
        assert ref_f[k].dtype == my_f[k].dtype and ref_f[k].shape == my_f[k].shape, k
        assert np.array_equal(ref_f[k], my_f[k]), k
    ref_sd = torch.load(os.path.join(ROOT, "tests", "golden", "g12_tile", "decoder.pth"), map_location="cpu")
    my_sd = torch.load(tmp_path / "mine" / "decoder.pth", map_location="cpu")
    assert list(ref_sd.keys()) == list(my_sd.keys())                                          # the consumer walks the keys IN ORDER
    for k in ref_sd:
        assert ref_sd[k].shape == my_sd[k].shape and torch.equal(ref_sd[k], my_sd[k]), k


def test_hashgrid_constant_conversions_follow_the_value():
    """HashGrid._converted (round 6): the host lists / device copies of the box and the sampler's log2dim are made once per VALUE --
    an in-place change (version counter) or a reassigned attribute (checkpoint load) is converted again; an unchanged one is the
    same object (no `.tolist()` / `.to(device)`: each stalls the host on a GPU)."""
    import scanerf_amd  # noqa: F401
    from scanerf_amd.hashgrid.grid import HashGrid

    class Stub:   # (the method needs `device` and the attribute only)
        device = "cpu"
        _converted = HashGrid._converted
    g = Stub()
    g.min_bbox = torch.tensor([1.0, 2.0, 3.0])
    g.sampler_log2dim = torch.tensor([4, 4, 3])
    a = g._converted("min_bbox", "list")
    assert a == [1.0, 2.0, 3.0] and g._converted("min_bbox", "list") is a
    d = g._converted("sampler_log2dim", "dev_int")
    assert d.dtype == torch.int32 and d.tolist() == [4, 4, 3] and g._converted("sampler_log2dim", "dev_int") is d
    g.min_bbox.add_(1.0)
    assert g._converted("min_bbox", "list") == [2.0, 3.0, 4.0]
    g.min_bbox = torch.tensor([9.0, 9.0, 9.0])
    assert g._converted("min_bbox", "list") == [9.0, 9.0, 9.0]
    g.sampler_log2dim = torch.tensor([5, 5, 5])
    assert g._converted("sampler_log2dim", "dev_int").tolist() == [5, 5, 5]
