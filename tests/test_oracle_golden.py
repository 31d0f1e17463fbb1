"""The oracle's torch half against golden vectors captured from the reference's Python
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import torch

from oracle import oracle as O

T = torch.from_numpy


def _sd(g):
    return {k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")}


def test_g1_mlp_forward(golden):
    g = golden("g1_mlp")
    out = O.mlp_forward(_sd(g), T(g["x"]), T(g["weight_feature"]))
    for k in ("sigma", "diffuse", "specular", "tint"):
        np.testing.assert_allclose(out[k].numpy(), g[k], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.mlp_sigma(_sd(g), T(g["x"][:, :32])).numpy(), g["sigma_only"], rtol=1e-5, atol=1e-6)


def test_g1_blob_decoder_matches_torch_mlp(golden):
    """a15: decoder.h blob inference == network.py forward on the same weights (weight_feature == 1)."""
    g = golden("g1_mlp")
    sd = _sd(g)
    blob = O.pack_blob(sd)
    assert blob.numel() == O.PARAMSIZE
    sd2 = O.unpack_blob(blob)
    for k in sd:
        assert torch.equal(sd[k], sd2[k])
    x = T(g["x"])
    ref = O.mlp_forward(sd, x, torch.ones(1, 32))
    sigma, diff, spec, tint = O.decoder_inference(blob, x[:, :32], x[:, 32:])
    np.testing.assert_allclose(sigma, ref["sigma"][:, 0].numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(diff, ref["diffuse"].numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(tint, ref["tint"].numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(spec, (ref["tint"] * ref["specular"]).numpy(), rtol=2e-5, atol=1e-6)


def test_g2_sh(golden):
    g = golden("g2_sh")
    np.testing.assert_allclose(O.sh_deg3(T(g["dirs"])).numpy(), g["sh"], rtol=1e-6, atol=1e-7)


def test_g3_composite(golden):
    g = golden("g3_composite")
    for inf in (0, 1):
        w, tl = O.cal_integrate_weight(T(g["sigma"]), T(g["dists"]), T(g["rays_d"]), infinity=bool(inf))
        np.testing.assert_allclose(w.numpy(), g["weights_inf%d" % inf], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(tl.numpy(), g["T_left_inf%d" % inf], rtol=1e-6)
        np.testing.assert_allclose(torch.sum(w * T(g["attr"]), 1).numpy(), g["acc_inf%d" % inf], rtol=1e-6)


def test_g4_contract(golden):
    g = golden("g4_contract")
    p, mn, sz = T(g["pts"]), T(g["min_bbox"]), T(g["bbox_size"])
    np.testing.assert_array_equal(O.contract_fore(p, mn, sz).numpy(), g["fore"])
    np.testing.assert_array_equal(O.contract_bg(p, mn, sz).numpy(), g["bg"])


def test_g5_weight_feature(golden):
    g = golden("g5_weight_feature")
    for s, w in zip(g["steps"], g["w"]):
        np.testing.assert_array_equal(O.weight_feature(int(s)).numpy(), w)


def test_g6_render_batch(golden):
    g = golden("g6_render_batch")
    sd = _sd(golden("g1_mlp"))
    mn, sz = torch.tensor([-8.0, -8.0, -8.0]), torch.tensor([16.0, 16.0, 16.0])
    for tag, fn, inf in (("fg", O.contract_fore, False), ("bg", O.contract_bg, True)):
        for mode in (0, 1):
            out = O.render_batch_rays(T(g["rays_o"]), T(g["rays_d"]), T(g["z_%s" % tag]), T(g["d_%s" % tag]),
                                      T(g["features"]), T(g["res"]), sd, mode,
                                      lambda x: fn(x, mn, sz), int(g["global_step"]), infinity=inf)
            for k in ("rgb", "depth", "T_left", "weights", "diffuse", "specular", "tint"):
                np.testing.assert_allclose(out[k].numpy(), g["%s_m%d_%s" % (tag, mode, k)], rtol=2e-5, atol=1e-7,
                                           err_msg=f"{tag} mode{mode} {k}")
            if mode == O.TRAIN:
                np.testing.assert_allclose(out["l2_reg_specular"].numpy(), g["%s_m0_l2_reg_specular" % tag], rtol=2e-5)


def test_g7_compute_ray_matches_camera_py(golden):
    """a1: compute_ray_forward (cuda_utils.h:143-155) == camera.get_center_and_ray_v2 semantics
    (+0.5 pixel centre, d not normalised) on the same C2W / K."""
    g = golden("g7_camera")
    H, W = int(g["H"]), int(g["W"])
    c2w = g["composed_inv"]  # camera.py works with w2c; the kernel takes c2w
    ncam, nidx = c2w.shape[0], g["ray_idx"].shape[0]
    locs = np.zeros((ncam, nidx, 3), np.int32)
    locs[..., 0] = np.arange(ncam)[:, None]
    locs[..., 1] = (g["ray_idx"] % W)[None, :]
    locs[..., 2] = (g["ray_idx"] // W)[None, :]
    o, d = O.compute_ray_forward(locs.reshape(-1, 3), g["ks"].reshape(ncam, 9), c2w.reshape(ncam, 12))
    np.testing.assert_allclose(o.reshape(ncam, nidx, 3), g["center"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(d.reshape(ncam, nidx, 3), g["ray"], rtol=1e-4, atol=1e-5)


def test_g8_consensus(golden):
    g = golden("g8_consensus")
    d1 = O.consensus_update(T(g["se3_refine"]), T(g["shared"]), T(g["delta0"]))
    np.testing.assert_allclose(d1.numpy(), g["delta1"], rtol=1e-6, atol=1e-9)
    loss = O.camera_loss(T(g["se3_refine"]), T(g["shared"]), d1, T(g["flags"]), T(g["rho"]))
    np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=1e-6)


def test_g10_adam_formula_matches_torch_adam(golden):
    """a14: the fused sparse Adam formula == torch.optim.Adam when every grad is non-zero."""
    g = golden("g10_adam")
    p = g["p0"].copy()
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    O.adam_step(p, g["g0"], m, v, 1e-2, 0.9, 0.99, 1e-15, 0)
    np.testing.assert_allclose(p, g["p1"], rtol=2e-5, atol=1e-7)
    O.adam_step(p, g["g1"], m, v, 1e-2, 0.9, 0.99, 1e-15, 1)
    np.testing.assert_allclose(p, g["p2"], rtol=2e-5, atol=1e-7)


# ---- round 5: G13-G15 (tests/golden/make_golden_sampling.py)
def test_g13_level_resolutions_per_axis(golden):
    """PyHashGridBG.__init__'s resolution rule (hashgrid/PyHashGridBG.py:53-62) on cubic and non-cubic boxes: the oracle's
    restatement AND the product's host function give the reference's own integers."""
    import scanerf_amd  # noqa
    from scanerf_amd.hashgrid import level_resolutions
    g = golden("g13_resolutions")
    for i in range(int(g["n"])):
        bbox_size = T(g[f"tile_size{i}"]) * 2
        gb, gf = (int(v) for v in g[f"grid_resolution{i}"])
        fin = (bbox_size / bbox_size.min() * gf).int()
        base = (bbox_size / bbox_size.min() * gb).int()
        np.testing.assert_array_equal(O.level_resolutions(base, fin, 16).numpy(), g[f"resolution{i}"])
        np.testing.assert_array_equal(level_resolutions(base, fin, 16).numpy(), g[f"resolution{i}"])
        tile = O.Tile([0, 0, 0], g[f"tile_size{i}"].tolist(), grid_resolution=(gb, gf))
        np.testing.assert_array_equal(tile.res.numpy(), g[f"resolution{i}"])


def test_g14_inverse_z_sampling(golden):
    g = golden("g14_inverse_z")
    for ug in (0, 1):
        z, d, v = O.inverse_z_sampling(T(g["rays_o"]), T(g["rays_d"]), T(g["bbox_center"]), T(g["bbox_size"]), int(g["S"]),
                                       invalid_underground=bool(ug))
        np.testing.assert_array_equal(v.numpy(), g["valid_ug%d" % ug])
        np.testing.assert_allclose(z.numpy(), g["z_ug%d" % ug], rtol=1e-6)
        np.testing.assert_allclose(d.numpy(), g["dists_ug%d" % ug], rtol=1e-5, atol=1e-7)


def g15_tile(g):
    tile = O.Tile(g["tile_corner"].tolist(), g["tile_size"].tolist(), log2_T=10, sampler_log2dim=4)
    tile.occ = T(g["occ"])
    assert np.array_equal(tile.res.numpy(), g["res"]) and np.array_equal(tile.log2dim.numpy(), g["log2dim"])
    return tile


def test_g15_fore_and_bg_valid_masks_and_fills(golden):
    """render_fore_rays / render_bg_rays (hashgrid/__init__.py:413-509): valid sets (sampler sentinel, under-ground rule,
    occlusion mask), zero / one fills of the invalid rays, values of the valid ones."""
    g = golden("g15_render_masks")
    tile, sd = g15_tile(g), _sd(g)
    for tag, m in (("nomask", None), ("mask", T(g["occlusion_mask"]))):
        for mode in (0, 1):
            out = O.render_rays(tile, T(g["features"]), sd, T(g["rays_o"]), T(g["rays_d"]), int(g["S"]), int(g["S"]), mode,
                                int(g["global_step"]), invalid_underground=True, occlusion_mask=m)
            np.testing.assert_array_equal(out["fore_valid"].numpy(), g[f"fg_{tag}_m{mode}_fore_valid"])
            np.testing.assert_array_equal(out["bg_valid"].numpy(), g[f"bg_{tag}_m{mode}_valid"])
            for k, gk in (("rgb", "pred_color"), ("depth", "pred_depth"), ("specular", "specular"), ("diffuse", "diffuse"),
                          ("T_left", "T_left")):
                np.testing.assert_allclose(out["fg"][k].numpy(), g[f"fg_{tag}_m{mode}_{gk}"], rtol=2e-5, atol=1e-7, err_msg=f"fg {k}")
            for k in ("rgb", "depth", "specular", "diffuse", "T_left"):
                np.testing.assert_allclose(out["bg"][k].numpy(), g[f"bg_{tag}_m{mode}_{k}"], rtol=2e-5, atol=1e-7, err_msg=f"bg {k}")
            if mode == O.TRAIN:
                np.testing.assert_allclose(out["fg_l2_reg_specular"].numpy(), g[f"fg_{tag}_l2_reg_specular"], rtol=2e-5)
                np.testing.assert_allclose(out["bg_l2_reg_specular"].numpy(), g[f"bg_{tag}_l2_reg_specular"], rtol=2e-5)


def test_g16_pruning_tile_grid(golden):
    """HashGrid.pruning_tile_grid run by the reference (hashgrid/__init__.py:138-213): the oracle's restatement gives the same
    occupancy grids, on the same level and across one 2x split."""
    g = golden("g16_pruning")
    sd = _sd(g)
    bbox_size = T(g["tile_size"]) * 2
    for tag, sub in (("same", False), ("split", True)):
        new, l2d = O.pruning_tile_grid(T(g["occ0"]), torch.tensor([3, 3, 3]), T(g["features"]), T(g["res"]), sd, bbox_size,
                                       int(g[f"{tag}_step"]), sub, float(g[f"{tag}_th"]), finest_resolution=int(g["grid_resolution"][1]),
                                       batch_size=4096)
        np.testing.assert_array_equal(l2d.numpy(), g[f"{tag}_log2dim"])
        diff = int((new.numpy() != g[f"{tag}_grid"]).sum())
        assert diff == 0, (tag, diff)


def _cam_from_g17(g, device="cpu"):
    import scanerf_amd  # noqa
    from scanerf_amd import cameras as CM
    cam = CM.CameraSet(T(g["ks"]), T(g["c2ws"]), device, noise=T(g["noise"]))
    with torch.no_grad():
        cam.se3_refine.copy_(T(g["se3_refine"]).to(device))
    return cam, CM


def test_g17_cam_rays_and_pose_gradient(golden):
    """camera_utils.CAM (the reference's pose-parameter module): poses from the product's Lie / pose algebra (pure torch), rays from
    the oracle's compute_ray_forward, and dL/d(se3_refine) = the oracle's compute_ray_backward (the mathematically correct adjoint,
    ref_bug = 0) chained through torch autograd of the pose algebra -- against the reference's own autograd."""
    g = golden("g17_cam_rays")
    cam, CM = _cam_from_g17(g)
    poses = cam.get_poses()
    np.testing.assert_allclose(poses.detach().numpy(), g["poses"], rtol=1e-5, atol=1e-6)
    C, n, W = g["c2ws"].shape[0], g["ray_idx"].shape[0], int(g["W"])
    locs = CM.pixel_locs(C, T(g["ray_idx"]), W, "cpu").numpy()
    o, d = O.compute_ray_forward(locs, g["ks"].reshape(C, 9), poses.detach().numpy().reshape(C, 12))
    np.testing.assert_allclose(o.reshape(C, n, 3), g["rays_o"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(d.reshape(C, n, 3), g["rays_d"], rtol=1e-4, atol=1e-5)
    gC = O.compute_ray_backward(g["w_o"].reshape(-1, 3), g["w_d"].reshape(-1, 3), g["ks"].reshape(C, 9), locs, C, ref_bug=False)
    poses.backward(T(np.asarray(gC, np.float32)).reshape(C, 3, 4))
    np.testing.assert_allclose(cam.se3_refine.grad.numpy(), g["grad_se3_refine"], rtol=2e-4, atol=2e-5)


def test_g19_composite_values_and_gradients(golden):
    """G19 (round 6): the reference's compositing sequence under its own autograd -- the oracle's cal_integrate_weight follows the
    values, and torch autograd through the oracle's formulation follows the reference's gradients (what the GPU test holds
    csrc/composite.hip to)."""
    g = golden("g19_composite_grads")
    for inf in (0, 1):
        t = "inf%d_" % inf
        sigma = T(g[t + "sigma"]).clone().requires_grad_(True)
        rd = T(g[t + "rays_d"]).clone().requires_grad_(True)
        w, tl = O.cal_integrate_weight(sigma, T(g[t + "dists"]), rd, infinity=bool(inf))
        np.testing.assert_allclose(w.detach().numpy(), g[t + "weights"], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(tl.detach().numpy(), g[t + "T_left"], rtol=1e-6)
        dif, spc, tnt = (T(g[t + k]) for k in ("diffuse", "specular", "tint"))
        acc = lambda v: torch.sum(w * v, 1)
        dif_o, spec_o, tint_o, depth = acc(dif), acc(tnt * spc), acc(tnt), acc(T(g[t + "z_vals"])[..., None])
        rgb = torch.clamp(dif_o + spec_o, 0, 1)
        loss = ((rgb * T(g[t + "cw_rgb"])).sum() + (depth * T(g[t + "cw_depth"])).sum() + (tl * T(g[t + "cw_T"])).sum()
                + (dif_o * T(g[t + "cw_dif"])).sum() + (spec_o * T(g[t + "cw_spec"])).sum() + (tint_o * T(g[t + "cw_tint"])).sum()
                + 0.1 * (w * T(g[t + "cw_w"])).sum())
        loss.backward()
        np.testing.assert_allclose(sigma.grad.numpy(), g[t + "g_sigma"], rtol=1e-4, atol=1e-5 * float(np.abs(g[t + "g_sigma"]).max()))
        np.testing.assert_allclose(rd.grad.numpy(), g[t + "g_rays_d"], rtol=1e-4, atol=1e-5 * float(np.abs(g[t + "g_rays_d"]).max()))


def test_g20_render_batch_rays_gradients(golden):
    """G20 (round 6): the reference's own loss.backward() through its render_batch_rays (the C oracle's encoder underneath) --
    autograd through the oracle's restatement gives the same gradients of the table, every decoder parameter and both ray tensors
    (foreground and background / infinity; step 7000: the coarse-to-fine mask partly closed)."""
    g = golden("g20_render_grads")
    mn = T(g["tile_corner"]) + T(g["tile_size"]) / 2 - T(g["tile_size"])
    sz = T(g["tile_size"]) * 2
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    for tag in ("fg", "bg"):
        sd = {k[3:]: T(v).clone().requires_grad_(True) for k, v in g.items() if k.startswith("sd.")}
        F = T(g["features"]).clone().requires_grad_(True)
        o, d = T(g[f"{tag}_rays_o"]).clone().requires_grad_(True), T(g[f"{tag}_rays_d"]).clone().requires_grad_(True)
        fn = (lambda x: O.contract_bg(x, mn, sz)) if tag == "bg" else (lambda x: O.contract_fore(x, mn, sz))
        out = O.render_batch_rays(o, d, T(g[f"{tag}_z_vals"]), T(g[f"{tag}_dists"]), F, T(g["res"]), sd, O.TRAIN, fn, int(g["global_step"]),
                                  infinity=(tag == "bg"))
        cw = {k: T(g[f"{tag}_cw_{k}"]) for k in ("rgb", "depth", "T_left", "diffuse", "specular", "tint")}
        loss = sum((out[k] * cw[k]).sum() for k in cw) + 0.37 * out["l2_reg_specular"] + 0.1 * (out["depth"][:, 0] * out["T_left"]).sum()
        np.testing.assert_allclose(float(loss), float(g[f"{tag}_loss"]), rtol=1e-6)
        loss.backward()
        assert rel(F.grad, T(g[f"{tag}_g_features"])) < 1e-5
        assert rel(o.grad, T(g[f"{tag}_g_rays_o"])) < 1e-5 and rel(d.grad, T(g[f"{tag}_g_rays_d"])) < 1e-5
        for n, v in sd.items():
            ref = T(g[f"{tag}_g_sd.{n}"])
            if float(ref.abs().max()) > 0:
                assert rel(v.grad, ref) < 1e-5, n
