"""The product's GPU path against the reference's own outputs captured in round 5 (tests/golden/make_golden_sampling.py):
G14 HashGrid.inverse_z_sampling, G15 render_fore_rays / render_bg_rays (valid masks, fills, values)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


def _sd(g):
    return {k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")}


def test_inverse_z_sampling_reproduces_reference_golden_g14(golden):
    """hashgrid/__init__.py:306-337 on the GPU (HIP ray_aabb_intersection + torch): the reference's z / dists / valid."""
    import scanerf_amd  # noqa
    from scanerf_amd.tile_model import TileModel
    g = golden("g14_inverse_z")
    size = g["bbox_size"] / 2
    corner = g["bbox_center"] - size / 2
    m = TileModel(corner.tolist(), size.tolist(), DEV, log2_T=10)
    for ug in (0, 1):
        z, d, v = m.inverse_z_sampling(T(g["rays_o"]).to(DEV), T(g["rays_d"]).to(DEV), int(g["S"]), invalid_underground=bool(ug))
        np.testing.assert_array_equal(v.cpu().numpy(), g["valid_ug%d" % ug])
        np.testing.assert_allclose(z.cpu().numpy(), g["z_ug%d" % ug], rtol=2e-6)
        np.testing.assert_allclose(d.cpu().numpy(), g["dists_ug%d" % ug], rtol=1e-4, atol=1e-6 * float(g["z_ug0"].max()))


@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_render_rays_fused_reproduces_reference_golden_g15(golden, tag):
    """render_fore_rays / render_bg_rays of the reference (INFERENCE mode) vs TileModel.render_rays_fused: same valid sets
    (sampler sentinel, under-ground rule, occlusion mask), zeros / T_left = 1 on invalid rays, values within 1e-4."""
    import scanerf_amd  # noqa
    from scanerf_amd import network, render
    from scanerf_amd.tile_model import TileModel
    g = golden("g15_render_masks")
    m = TileModel(g["tile_corner"].tolist(), g["tile_size"].tolist(), DEV, log2_T=10, sampler_log2dim=4)
    assert np.array_equal(m.resolution.cpu().numpy(), g["res"])
    m.set_occupancy(T(g["occ"]))
    with torch.no_grad():
        m.features.copy_(T(g["features"]).to(DEV))
    m.decoder.load_ref_state_dict(_sd(g))
    mask = T(g["occlusion_mask"]).to(DEV) if tag == "mask" else None
    S = int(g["S"])
    out = m.render_rays_fused(T(g["rays_o"]).to(DEV), T(g["rays_d"]).to(DEV), S, S, int(g["global_step"]),
                              invalid_underground=True, occlusion_mask=mask)
    vf, vb = out["fore_valid"].bool().cpu().numpy(), out["bg_valid"].bool().cpu().numpy()
    np.testing.assert_array_equal(vf, g[f"fg_{tag}_m1_fore_valid"])
    np.testing.assert_array_equal(vb, g[f"bg_{tag}_m1_valid"])
    assert 0 < vf.sum() < vf.size and 0 < vb.sum() < vb.size
    fg, bg = out["fg"].cpu().numpy(), out["bg"].cpu().numpy()
    tol = dict(rtol=1e-4, atol=2e-6)
    for arr, pre, rgbk, depk in ((fg, "fg", "pred_color", "pred_depth"), (bg, "bg", "rgb", "depth")):
        np.testing.assert_allclose(arr[:, render.RGB], g[f"{pre}_{tag}_m1_{rgbk}"], **tol)
        np.testing.assert_allclose(arr[:, render.DEPTH], g[f"{pre}_{tag}_m1_{depk}"][:, 0], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(arr[:, render.DIFFUSE], g[f"{pre}_{tag}_m1_diffuse"], **tol)
        np.testing.assert_allclose(arr[:, render.SPECULAR], g[f"{pre}_{tag}_m1_specular"], **tol)
        np.testing.assert_allclose(arr[:, render.T_LEFT], g[f"{pre}_{tag}_m1_T_left"][:, 0], **tol)
    # invalid rays: exactly the reference's fills
    assert not fg[~vf][:, [0, 1, 2, 3, 5, 6, 7, 8, 9, 10]].any() and (fg[~vf][:, render.T_LEFT] == 1).all()
    assert not bg[~vb][:, [0, 1, 2, 3, 5, 6, 7, 8, 9, 10]].any() and (bg[~vb][:, render.T_LEFT] == 1).all()
    # merged prediction (tile.py:666-690)
    Tl = g[f"fg_{tag}_m1_T_left"]
    np.testing.assert_allclose(out["pred_color"].cpu().numpy(), g[f"fg_{tag}_m1_pred_color"] + Tl * g[f"bg_{tag}_m1_rgb"], **tol)


@pytest.mark.parametrize("tag,sub", [("same", False), ("split", True)])
def test_pruning_reproduces_reference_golden_g16(golden, tag, sub):
    """Coarse-to-fine occupancy pruning (hashgrid/__init__.py:138-213) as the reference computed it: the reference-shaped
    HashGrid module (HIP encoder op + decoder.inference_sigma) and trainer.pruning_tile_grid on a TileModel.  A cell whose
    largest alpha lies within 1e-4 of the threshold may fall either way (f32 encoder differences): at most 2 such cells."""
    import scanerf_amd  # noqa
    from scanerf_amd import network, trainer
    from scanerf_amd.hashgrid import HashGrid
    from scanerf_amd.tile_model import TileModel
    g = golden("g16_pruning")
    gb, gf = (int(v) for v in g["grid_resolution"])
    step, th = int(g[f"{tag}_step"]), float(g[f"{tag}_th"])
    sd = _sd(g)
    hg = HashGrid(DEV, T(g["tile_corner"]), T(g["tile_size"]), log2_hashmap_size=10, grid_resolution=[gb, gf], sampler_log2dim=3)
    assert np.array_equal(hg.HE.resolution.cpu().numpy(), g["res"])
    hg.occupied_grid = T(g["occ0"]).to(DEV)
    with torch.no_grad():
        hg.HE.features.copy_(T(g["features"]).to(DEV))
    dec = network.ShallowMLP(32)
    dec.load_state_dict(sd)
    hg.pruning_tile_grid(step, dec.to(DEV), sub_split=sub, pruning_th=th, batch_size=4096)
    assert np.array_equal(hg.sampler_log2dim.cpu().numpy(), g[f"{tag}_log2dim"])
    d1 = int((hg.occupied_grid.cpu().numpy() != g[f"{tag}_grid"]).sum())
    m = TileModel(g["tile_corner"].tolist(), g["tile_size"].tolist(), DEV, log2_T=10, grid_resolution=(gb, gf), sampler_log2dim=3)
    m.set_occupancy(T(g["occ0"]))
    with torch.no_grad():
        m.features.copy_(T(g["features"]).to(DEV))
    m.decoder.load_ref_state_dict(sd)
    trainer.pruning_tile_grid(m, step, sub_split=sub, pruning_th=th, batch_size=4096, finest_resolution=gf)
    assert np.array_equal(m.log2dim.cpu().numpy(), g[f"{tag}_log2dim"])
    d2 = int((m.occupied_grid.cpu().numpy() != g[f"{tag}_grid"]).sum())
    print(f"pruning vs the reference ({tag}): {d1} / {d2} of {g[f'{tag}_grid'].size} cells differ (HashGrid module / TileModel)")
    assert d1 <= 2 and d2 <= 2, (d1, d2)


def test_camera_set_rays_and_pose_gradient_reproduce_reference_golden_g17(golden):
    """cameras.CameraSet on the HIP ray kernels (compute_ray_forward; backward = compute_ray_backward, the adjoint) against
    camera_utils.CAM.getRays and the reference's autograd of a loss on the rays w.r.t. se3_refine."""
    import scanerf_amd  # noqa
    from scanerf_amd import cameras as CM
    g = golden("g17_cam_rays")
    cam = CM.CameraSet(T(g["ks"]), T(g["c2ws"]), DEV, noise=T(g["noise"]))
    with torch.no_grad():
        cam.se3_refine.copy_(T(g["se3_refine"]).to(DEV))
    C, n = g["c2ws"].shape[0], g["ray_idx"].shape[0]
    np.testing.assert_allclose(cam.get_poses().detach().cpu().numpy(), g["poses"], rtol=1e-5, atol=1e-6)
    ro, rd = cam.get_rays_idx(int(g["W"]), T(g["ray_idx"]))
    np.testing.assert_allclose(ro.detach().cpu().numpy().reshape(C, n, 3), g["rays_o"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rd.detach().cpu().numpy().reshape(C, n, 3), g["rays_d"], rtol=1e-4, atol=1e-5)
    loss = (ro.reshape(C, n, 3) * T(g["w_o"]).to(DEV)).sum() + (rd.reshape(C, n, 3) * T(g["w_d"]).to(DEV)).sum()
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-4, atol=1e-4)
    loss.backward()
    np.testing.assert_allclose(cam.se3_refine.grad.cpu().numpy(), g["grad_se3_refine"], rtol=2e-4, atol=2e-5)


def test_hashgrid_module_files_round_trip(tmp_path):
    """HashGrid.export / load (feature.npz, hashgrid/__init__.py:248-266; the renderer's loader reads it) and export_check_point /
    load_check_point (tile.py:541-569) on the reference-shaped module."""
    import scanerf_amd  # noqa
    from scanerf_amd import renderer
    from scanerf_amd.hashgrid import HashGrid
    torch.manual_seed(1)
    hg = HashGrid(DEV, torch.tensor([0.0, 0, 0]), torch.tensor([4.0, 2.0, 4.0]), log2_hashmap_size=10, grid_resolution=[16, 256], sampler_log2dim=3)
    hg.occupied_grid = (torch.rand(tuple(hg.occupied_grid.shape)) < 0.5).to(DEV)
    hg.export(str(tmp_path))
    f = np.load(tmp_path / "feature.npz")
    assert list(f.keys()) == ["features", "occupied_grid", "block_corner", "block_size", "grid_log2dim", "resolution"]
    assert f["features"].dtype == np.float16 and np.array_equal(f["resolution"], hg.HE.resolution.cpu().numpy())
    np.testing.assert_array_equal(f["block_corner"], hg.min_bbox.cpu().numpy())
    torch.save(network_state(), tmp_path / "decoder.pth")
    tile = renderer.load_tile(str(tmp_path))       # the render-time loader accepts the pair
    assert tile["features"].shape == (16, 1024, 2) and tile["blob"].shape == (13994,)
    hg2 = HashGrid(DEV, torch.tensor([0.0, 0, 0]), torch.tensor([4.0, 2.0, 4.0]), log2_hashmap_size=10, grid_resolution=[16, 256], sampler_log2dim=3)
    hg2.load(str(tmp_path))
    assert torch.equal(hg2.occupied_grid, hg.occupied_grid) and torch.equal(hg2.HE.features.detach(), hg.HE.features.detach().half().float())
    ck = hg.export_check_point()
    hg2.load_check_point(ck)
    assert torch.equal(hg2.HE.features.detach(), hg.HE.features.detach()) and torch.equal(hg2.sampler_log2dim.cpu(), hg.sampler_log2dim.cpu())


def network_state():
    from scanerf_amd import network
    return {k: v.detach().clone() for k, v in network.init_model(network.ShallowMLP(32), "xavier").state_dict().items()}
