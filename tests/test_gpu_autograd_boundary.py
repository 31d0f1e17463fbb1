"""The fused kernels behind an autograd boundary (render.FusedRenderRays) and the reference-shaped HashGrid module
(hashgrid/__init__.py:32-596) that uses it: a caller that keeps the reference's loss code -- arbitrary terms on rgb / depth /
T_left / diffuse / specular / tint / l2_reg_specular through loss.backward() (tile.py:954-1011, criterions.py:122-196) --
reaches the fused kernels; and an UNCHANGED render_batch_rays call sequence (encoder op, decoder module, torch compositing)
runs on the HIP encoder + the HIP decoder op."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


def _rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def _inputs(rng, B, S_, bg):
    o = rng.uniform(-3, 3, (B, 3)).astype(np.float32)
    d = (rng.normal(size=(B, 3)) * rng.uniform(0.5, 1.5, (B, 1))).astype(np.float32)
    if bg:
        z = np.sort(rng.uniform(9, 70, (B, S_)), 1).astype(np.float32)
        dist = np.concatenate([np.diff(z, axis=1), np.full((B, 1), 1e-6, np.float32)], 1).astype(np.float32)
    else:
        z = np.sort(rng.uniform(0.2, 3.2, (B, S_)), 1).astype(np.float32)
        dist = np.concatenate([np.diff(z, axis=1), np.full((B, 1), 0.05, np.float32)], 1).astype(np.float32)
    return o, d, z, dist


def _loss(out, w):
    """A loss of the kind criterions.py builds: terms on colour, depth, transmittance, the decomposed colours and the regulariser."""
    return ((out["rgb"] * w["rgb"]).sum() + (out["depth"] * w["depth"]).sum() + (out["T_left"] * w["T"]).sum()
            + (out["diffuse"] * w["dif"]).sum() + (out["specular"] * w["spec"]).sum() + (out["tint"] * w["tint"]).sum()
            + 0.37 * out["l2_reg_specular"] + (out["depth"][:, 0] * out["T_left"]).sum() * 0.1)


@pytest.mark.parametrize("bg,S_,log2_T", [(False, 64, 12), (True, 40, 12), (False, 128, 22)])
def test_arbitrary_loss_through_loss_backward_vs_oracle_autograd(bg, S_, log2_T):
    """render_batch_rays as one differentiable op: gradients of the table, the decoder and BOTH ray tensors of an arbitrary
    loss against autograd through the oracle's render_batch_rays (f32 on the host)."""
    import scanerf_amd  # noqa
    from scanerf_amd import network, render
    rng = np.random.default_rng(21)
    B, Tn = 160, 2 ** log2_T
    o, d, z, dist = _inputs(rng, B, S_, bg)
    feat = (rng.normal(size=(16, Tn, 2)) * 0.5).astype(np.float32)
    sd = {k: v.clone().requires_grad_(True) for k, v in O.init_mlp(seed=5, bias_scale=0.05).items()}
    res = O.level_resolutions(torch.tensor([32, 32, 32]), torch.tensor([2048, 2048, 2048]))
    mn, sz = torch.tensor([-8.0, -8.0, -8.0]), torch.tensor([16.0, 16.0, 16.0])
    fn = (lambda x: O.contract_bg(x, mn, sz)) if bg else (lambda x: O.contract_fore(x, mn, sz))
    step = 7000
    w = {"rgb": T(rng.normal(size=(B, 3)).astype(np.float32)), "depth": T(rng.normal(size=(B, 1)).astype(np.float32)),
         "T": T(rng.normal(size=(B,)).astype(np.float32)), "dif": T(rng.normal(size=(B, 3)).astype(np.float32)),
         "spec": T(rng.normal(size=(B, 3)).astype(np.float32)), "tint": T(rng.normal(size=(B, 3)).astype(np.float32))}
    F = T(feat).requires_grad_(True)
    to, td = T(o).requires_grad_(True), T(d).requires_grad_(True)
    ref = O.render_batch_rays(to, td, T(z), T(dist), F, res, sd, O.TRAIN, fn, step, infinity=bg)
    _loss(ref, w).backward()
    gblob_ref = O.pack_blob({k: v.grad for k, v in sd.items()})

    Fg = T(feat).to(DEV).requires_grad_(True)
    blob = O.pack_blob({k: v.detach() for k, v in sd.items()}).to(DEV).requires_grad_(True)
    og, dg = T(o).to(DEV).requires_grad_(True), T(d).to(DEV).requires_grad_(True)
    out_ray, weights = render.fused_render_rays(og, dg, T(z).to(DEV), T(dist).to(DEV), Fg, blob, res.to(DEV).int().contiguous(),
                                                network.weight_feature(step, DEV), mn.tolist(), sz.tolist(),
                                                render.BG if bg else render.FORE, bg, None, network.skip_levels(step), True)
    out = render.render_batch_rays_dict(out_ray, weights, True)
    for k in ("rgb", "depth", "T_left", "diffuse", "specular", "tint", "weights", "l2_reg_specular"):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), ref[k].detach().numpy(), rtol=1e-4, atol=1e-6, err_msg=k)
    _loss(out, {k: v.to(DEV) for k, v in w.items()}).backward()
    assert Fg.grad.shape == Fg.shape and blob.grad.shape == blob.shape
    errs = {"table": _rel_l2(Fg.grad.cpu(), F.grad), "decoder": _rel_l2(blob.grad.cpu(), gblob_ref),
            "rays_o": _rel_l2(og.grad.cpu(), to.grad), "rays_d": _rel_l2(dg.grad.cpu(), td.grad)}
    print(f"autograd boundary (bg={bg}, S={S_}, T=2^{log2_T}): relative L2 vs the oracle's autograd {errs}")
    assert errs["table"] < 5e-5 and errs["decoder"] < 5e-5, errs
    assert errs["rays_o"] < 2e-3 and errs["rays_d"] < 2e-3, errs   # (the position path: trilinear kinks, as test_ray_gradients_vs_oracle_autograd)


def test_mse_loss_through_the_boundary_is_bit_equal_to_train_step_fused():
    """With the loss gradient of the training step's own loss kernel, FusedRenderRays.backward gives bit for bit the table
    and decoder gradients train_step_fused computes (same kernels, same records, same deterministic sums)."""
    import scanerf_amd  # noqa
    from scanerf_amd import network, render
    from scanerf_amd.tile_model import TileModel, train_step_fused
    torch.manual_seed(0)
    B, S_ = 4096, 64
    m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=14, seed=1)
    with torch.no_grad():
        m.features.mul_(200.0)
    o = torch.rand(B, 3, device=DEV) * 8 - 4
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1)
    tgt = torch.rand(B, 3, device=DEV)
    step = 20000
    z, dist = m.sample(o, d, S_)
    F = m.features.detach().clone().requires_grad_(True)
    blob = m.decoder.blob().detach().clone().requires_grad_(True)
    out_ray, _ = render.fused_render_rays(o, d, z, dist, F, blob, m.resolution, network.weight_feature(step, DEV), m.min_bbox.tolist(),
                                          m.bbox_size.tolist(), render.FORE, False, None, 0, False)
    loss, grad_out = render.photometric_loss_grad(out_ray.detach(), tgt, None, 0.01)
    out_ray.backward(grad_out)
    # the same through torch's own MSE graph: equal up to the rounding of the loss gradient
    F2 = m.features.detach().clone().requires_grad_(True)
    out2, _ = render.fused_render_rays(o, d, z, dist, F2, blob.detach(), m.resolution, network.weight_feature(step, DEV), m.min_bbox.tolist(),
                                       m.bbox_size.tolist(), render.FORE, False, None, 0, False)
    l2 = torch.nn.functional.mse_loss(out2[:, render.RGB], tgt) + 0.01 * out2[:, render.W_SPEC2].sum() / (3.0 * B)
    l2.backward()
    np.testing.assert_allclose(float(l2.detach()), float(loss), rtol=1e-6)
    assert _rel_l2(F2.grad, F.grad) < 1e-6
    dummy = torch.optim.SGD(m.decoder.parameters(), lr=0.0)
    train_step_fused(m, dummy, o, d, tgt, S_, step, table_lr=0.0, fused_adam=False, dec_step=False)
    assert torch.equal(m.features.grad, F.grad)
    assert torch.equal(m.decoder.params.grad, blob.grad)


def _hashgrid_from_g15(g, fused):
    import scanerf_amd  # noqa
    from scanerf_amd import network
    from scanerf_amd.hashgrid import HashGrid
    hg = HashGrid(DEV, T(g["tile_corner"]), T(g["tile_size"]), log2_hashmap_size=10, grid_resolution=[32, 2048], sampler_log2dim=4)
    assert np.array_equal(hg.HE.resolution.cpu().numpy(), g["res"]) and np.array_equal(hg.sampler_log2dim.cpu().numpy(), g["log2dim"])
    hg.occupied_grid = T(g["occ"]).to(DEV)
    with torch.no_grad():
        hg.HE.features.copy_(T(g["features"]).to(DEV))
    dec = network.ShallowMLP(32)
    dec.load_state_dict({k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")})
    hg.fused = fused
    return hg, dec.to(DEV)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_hashgrid_module_reproduces_reference_render_fore_and_bg_rays_golden_g15(golden, fused, tag):
    """The reference-shaped module on both routes (fused op / encoder op + decoder op + torch compositing) against the
    reference's own render_fore_rays / render_bg_rays outputs: dictionaries key by key, TRAIN and INFERENCE."""
    g = golden("g15_render_masks")
    hg, dec = _hashgrid_from_g15(g, fused)
    o, d = T(g["rays_o"]).to(DEV), T(g["rays_d"]).to(DEV)
    mask = T(g["occlusion_mask"]).to(DEV) if tag == "mask" else None
    S_, step = int(g["S"]), int(g["global_step"])
    for mode in (0, 1):
        with torch.no_grad():
            fo, ok = hg.render_fore_rays(o, d, S_, dec, mode, occlusion_mask=mask, global_step=step)
            assert ok and hg.last_render_route == ("fused" if fused else "ops")
            bo, ok = hg.render_bg_rays(o, d, S_, dec, mode, occlusion_mask=mask, global_step=step, bg_mode="IZ", invalid_underground=True)
            assert ok
        tol = dict(rtol=1e-4, atol=2e-6)
        for k in ("fore_valid", "pred_color", "pred_depth", "specular", "diffuse", "T_left"):
            np.testing.assert_allclose(fo[k].cpu().numpy().astype(np.float64), g[f"fg_{tag}_m{mode}_{k}"].astype(np.float64),
                                       **(dict(rtol=1e-4, atol=1e-4) if k == "pred_depth" else tol), err_msg=f"fg {k}")
        for k in ("valid", "rgb", "depth", "specular", "diffuse", "T_left"):
            np.testing.assert_allclose(bo[k].cpu().numpy().astype(np.float64), g[f"bg_{tag}_m{mode}_{k}"].astype(np.float64),
                                       **(dict(rtol=1e-4, atol=1e-4) if k == "depth" else tol), err_msg=f"bg {k}")
        if mode == 0:
            np.testing.assert_allclose(float(fo["l2_reg_specular"]), float(g[f"fg_{tag}_l2_reg_specular"]), rtol=1e-4)
            np.testing.assert_allclose(float(bo["l2_reg_specular"]), float(g[f"bg_{tag}_l2_reg_specular"]), rtol=1e-4)


def test_unchanged_call_sequence_trains_through_the_hip_ops_and_agrees_with_the_fused_route(golden):
    """tile.py:639-692's call sequence (render_fore_rays + render_bg_rays, merge, a loss with more than the MSE term,
    loss.backward()) on both routes of the module: same loss, gradients of table / decoder parameters within 1e-4."""
    g = golden("g15_render_masks")
    grads = {}
    for fused in (True, False):
        hg, dec = _hashgrid_from_g15(g, fused)
        o, d = T(g["rays_o"]).to(DEV), T(g["rays_d"]).to(DEV)
        tgt = torch.linspace(0, 1, o.numel(), device=DEV).reshape(-1, 3)
        fo, _ = hg.render_fore_rays(o, d, 16, dec, 0, global_step=6000)
        bo, _ = hg.render_bg_rays(o, d, 16, dec, 0, global_step=6000, bg_mode="IZ", invalid_underground=True)
        pred = fo["pred_color"] + fo["T_left"] * bo["rgb"]
        depth = fo["pred_depth"] + fo["T_left"] * bo["depth"]
        loss = ((pred - tgt) ** 2).mean() + 0.01 * (fo["l2_reg_specular"] + bo["l2_reg_specular"]) + 1e-3 * (depth ** 2).mean()
        loss.backward()
        grads[fused] = (float(loss.detach()), hg.HE.features.grad.clone(), {n: p.grad.clone() for n, p in dec.named_parameters()})
    np.testing.assert_allclose(grads[True][0], grads[False][0], rtol=1e-5)
    assert _rel_l2(grads[True][1], grads[False][1]) < 1e-4
    for n in grads[True][2]:
        assert _rel_l2(grads[True][2][n], grads[False][2][n]) < 1e-4, n


@pytest.mark.parametrize("table_dtype", [torch.float32, torch.bfloat16])
def test_boundary_with_a_ray_mask_equals_the_compacted_batch(table_dtype):
    """ray_valid through the autograd boundary: masked rays render as zeros with T_left = 1 and receive zero ray gradients;
    table / decoder gradients equal those of the batch with the masked rays removed (what hashgrid/__init__.py:419-434 does by
    boolean-mask indexing).  Also with a bf16 gather table (configs[2]): the gradient comes back in the table's dtype."""
    import scanerf_amd  # noqa
    from scanerf_amd import network, render
    rng = np.random.default_rng(3)
    B, S_ = 300, 48
    o, d, z, dist = (T(a).to(DEV) for a in _inputs(rng, B, S_, False))
    feat = T((rng.normal(size=(16, 2 ** 12, 2)) * 0.5).astype(np.float32)).to(DEV).to(table_dtype)
    res = O.level_resolutions(torch.tensor([32, 32, 32]), torch.tensor([2048, 2048, 2048])).to(DEV).int().contiguous()
    blob0 = network.xavier_blob(2, DEV, bias_scale=0.05)
    keep = T(rng.random(B) < 0.6).to(DEV)
    w = torch.randn(B, 16, device=DEV)
    w[:, 15] = 0
    box = ([-8.0] * 3, [16.0] * 3, render.FORE, False)
    wf = network.weight_feature(9000, DEV)

    def run(oo, dd, zz, di, mask, ww):
        F = feat.clone().requires_grad_(True)
        blob = blob0.clone().requires_grad_(True)
        og, dg = oo.clone().requires_grad_(True), dd.clone().requires_grad_(True)
        out, _ = render.fused_render_rays(og, dg, zz, di, F, blob, res, wf, *box, mask, 0, True)
        (out * ww).sum().backward()
        return out.detach(), F.grad, blob.grad, og.grad, dg.grad

    out_m, gF_m, gb_m, go_m, gd_m = run(o, d, z, dist, keep, w)
    out_c, gF_c, gb_c, go_c, gd_c = run(o[keep], d[keep], z[keep], dist[keep], None, w[keep])
    assert gF_m.dtype == table_dtype and gF_m.shape == feat.shape
    dead = ~keep
    assert bool((out_m[dead][:, [0, 1, 2, 3, 5, 6, 7, 8, 9, 10]] == 0).all()) and bool((out_m[dead][:, 4] == 1).all())
    assert float(go_m[dead].abs().max()) == 0.0 and float(gd_m[dead].abs().max()) == 0.0
    assert torch.equal(out_m[keep], out_c)
    tol = 1e-5 if table_dtype == torch.float32 else 2e-2   # (bf16: the returned gradient is rounded to the table's dtype)
    assert _rel_l2(gF_m.float(), gF_c.float()) < tol
    assert _rel_l2(gb_m, gb_c) < 1e-5
    assert _rel_l2(go_m[keep], go_c) < 1e-4 and _rel_l2(gd_m[keep], gd_c) < 1e-4


def test_boundary_plans_in_its_forward_launch_and_two_calls_do_not_share_a_workspace(monkeypatch):
    """Round 6: FusedRenderRays.forward lets the forward kernel count the backward's record ranges into a workspace the call owns
    (no separate count launch in backward).  Two calls whose backwards run AFTER both forwards (foreground + background of a tile:
    the per-stream workspace of the first would have been re-planned by the second) give the table / decoder / ray gradients of
    the plan-in-backward form, bit for bit."""
    import scanerf_amd  # noqa
    from scanerf_amd import network, render
    rng = np.random.default_rng(4)
    B, S_, Tn = 2048, 64, 2 ** 14
    res = O.level_resolutions(torch.tensor([32, 32, 32]), torch.tensor([2048, 2048, 2048])).to(DEV).int().contiguous()
    feat = T((rng.normal(size=(16, Tn, 2)) * 0.5).astype(np.float32)).to(DEV)
    blob = network.xavier_blob(3).to(DEV)
    wf = network.weight_feature(20000, DEV)
    fg = [T(a).to(DEV) for a in _inputs(rng, B, S_, False)]
    bg = [T(a).to(DEV) for a in _inputs(rng, B, S_, True)]
    grads = {}
    for in_fwd in (True, False):
        monkeypatch.setattr(render, "FORWARD_PLAN_IN_AUTOGRAD", in_fwd)
        F, bl = feat.clone().requires_grad_(True), blob.clone().requires_grad_(True)
        outs = []
        for (o, d, z, dist), mode, inf in ((fg, render.FORE, False), (bg, render.BG, True)):
            out, _ = render.fused_render_rays(o, d, z, dist, F, bl, res, wf, [-8.0] * 3, [16.0] * 3, mode, inf)
            outs.append(out)
        assert (outs[0].grad_fn.plan_ws is not None) == in_fwd
        loss = (outs[0][:, render.RGB] + outs[0][:, render.T_LEFT, None] * outs[1][:, render.RGB]).square().sum() + outs[1][:, render.DEPTH].sum() * 1e-3
        loss.backward()
        grads[in_fwd] = (F.grad.clone(), bl.grad.clone())
    assert torch.equal(grads[True][0], grads[False][0]) and torch.equal(grads[True][1], grads[False][1])
    assert float(grads[True][0].abs().max()) > 0


def test_surface_normals_of_the_op_by_op_route_follow_the_torch_decoder(golden):
    """render_batch_rays(out_normal=True) on the op-by-op route (hashgrid/__init__.py:576-584: -d(sigma)/d(position), normalised,
    composited with the weights): a first-order autograd.grad through the HIP decoder op and the encoder op's point gradient --
    against the same call with the torch decoder graph (ShallowMLP.use_hip = False)."""
    g = golden("g15_render_masks")
    res = {}
    for use_hip in (True, False):
        hg, dec = _hashgrid_from_g15(g, False)
        dec.use_hip = use_hip
        o, d = T(g["rays_o"]).to(DEV), T(g["rays_d"]).to(DEV)
        z, dist = hg.samplePoints(o, d, 16)
        v = torch.all(z != -1, dim=-1)
        out, ok = hg.render_batch_rays(o[v], d[v], z[v], dist[v], dec, 1, hg.contract_fore, out_normal=True, global_step=20000)
        assert ok and out["normal"].shape == (int(v.sum()), 3) and bool(torch.isfinite(out["normal"]).all())
        res[use_hip] = out["normal"].detach()
    assert float(res[True].abs().max()) > 1e-3
    assert _rel_l2(res[True], res[False]) < 1e-3


def test_surface_normals_against_the_reference_golden_g18(golden):
    """G18: the reference's own render_batch_rays(out_normal=True) (hashgrid/__init__.py:576-588: autograd of sigma w.r.t. the sample
    positions through ITS decoder, normalised, composited with ITS weights; the C oracle as its encoder) on G15's tile, table and
    decoder -- against the op-by-op route here: a first-order autograd.grad through the HIP decoder op and the HIP encoder's point
    gradient, the compositing op's weights."""
    g15, g18 = golden("g15_render_masks"), golden("g18_normals")
    hg, dec = _hashgrid_from_g15(g15, False)
    o, d, z, dist = (T(g18[k]).to(DEV) for k in ("rays_o", "rays_d", "z_vals", "dists"))
    out, ok = hg.render_batch_rays(o, d, z, dist, dec, 0, hg.contract_fore, out_normal=True, global_step=int(g18["global_step"]))
    assert ok and hg.last_render_route == "ops"
    np.testing.assert_allclose(out["rgb"].detach().cpu().numpy(), g18["rgb"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["depth"].detach().cpu().numpy(), g18["depth"], rtol=1e-4, atol=1e-5)
    n, ref = out["normal"].detach().cpu(), T(g18["normal"])
    assert float(ref.norm(dim=-1).max()) > 0.1 and _rel_l2(n, ref) < 2e-3, _rel_l2(n, ref)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("tag", ["fg", "bg"])
def test_gradients_against_the_reference_own_autograd_golden_g20(golden, fused, tag):
    """G20: the reference's own loss.backward() through ITS render_batch_rays (decoder module, compositing, contraction; the C oracle's
    encoder adjoint underneath) for a loss with a term on every output -- gradients of the hash table, of every decoder parameter
    and of both ray tensors, foreground and background / infinity, at a step where the coarse-to-fine mask is partly closed.
    Both routes of the module (the fused kernels behind render.FusedRenderRays; encoder op + decoder op + compositing op)."""
    import scanerf_amd  # noqa
    from scanerf_amd import network
    from scanerf_amd.hashgrid import HashGrid
    g = golden("g20_render_grads")
    hg = HashGrid(DEV, T(g["tile_corner"]), T(g["tile_size"]), log2_hashmap_size=10, grid_resolution=[32, 2048], sampler_log2dim=4)
    assert np.array_equal(hg.HE.resolution.cpu().numpy(), g["res"])
    with torch.no_grad():
        hg.HE.features.copy_(T(g["features"]).to(DEV))
    dec = network.ShallowMLP(32)
    dec.load_state_dict({k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")})
    dec = dec.to(DEV)
    hg.fused = fused
    o = T(g[f"{tag}_rays_o"]).to(DEV).requires_grad_(True)
    d = T(g[f"{tag}_rays_d"]).to(DEV).requires_grad_(True)
    z, dist = T(g[f"{tag}_z_vals"]).to(DEV), T(g[f"{tag}_dists"]).to(DEV)
    cfn, inf = (hg.contract_bg, True) if tag == "bg" else (hg.contract_fore, False)
    out, ok = hg.render_batch_rays(o, d, z, dist, dec, 0, cfn, out_normal=False, infinity=inf, global_step=int(g["global_step"]))
    assert ok and hg.last_render_route == ("fused" if fused else "ops")
    cw = {k: T(g[f"{tag}_cw_{k}"]).to(DEV) for k in ("rgb", "depth", "T_left", "diffuse", "specular", "tint")}
    loss = sum((out[k] * cw[k]).sum() for k in cw) + 0.37 * out["l2_reg_specular"] + 0.1 * (out["depth"][:, 0] * out["T_left"]).sum()
    np.testing.assert_allclose(float(loss.detach()), float(g[f"{tag}_loss"]), rtol=2e-4)
    loss.backward()
    assert _rel_l2(hg.HE.features.grad.cpu(), T(g[f"{tag}_g_features"])) < 2e-4
    for n, p in dec.named_parameters():
        ref = T(g[f"{tag}_g_sd.{n}"])
        if float(ref.abs().max()) > 0:
            assert _rel_l2(p.grad.cpu(), ref) < 2e-4, n
    assert _rel_l2(o.grad.cpu(), T(g[f"{tag}_g_rays_o"])) < 1e-3
    assert _rel_l2(d.grad.cpu(), T(g[f"{tag}_g_rays_d"])) < 1e-3
