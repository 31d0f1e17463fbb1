"""csrc/composite.hip: alpha compositing along rays and its adjoint as stand-alone ops (HashGrid.cal_integrate_weight + accumulate,
hashgrid/__init__.py:344-366, :564-574, :591-594) against the torch formulation in float64, and the decoder op on separate
feature / direction tensors against the same op on the concatenated input."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _torch_composite(sigma, dif, spec, tint, z, dists, rays_d, infinity):
    """The reference's sequence (hashgrid/__init__.py:344-366, :564-574, :591-594) in the dtype of its inputs."""
    delta = dists * rays_d.norm(dim=-1, keepdim=True)
    if infinity:
        delta = torch.cat([delta[:, :-1], torch.full_like(delta[:, :1], 1e10)], 1)
    alpha = 1.0 - torch.exp(-sigma * delta)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-6], 1), 1)[:, :-1]
    w = alpha * T
    acc = lambda v: (w[..., None] * v).sum(1)
    diffuse, tn, specular = acc(dif), acc(tint), acc(tint * spec)
    return {"rgb": torch.clamp(diffuse + specular, 0, 1), "depth": (w * z).sum(1), "T_left": T[:, -1], "diffuse": diffuse,
            "specular": specular, "tint": tn, "w_spec2": (w.detach()[..., None] * spec ** 2).sum((1, 2)), "weights": w}


def _inputs(B, S, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    sigma = (torch.rand(B, S, generator=g) ** 3 * 8).to(dtype)
    sigma[:, ::7] = 0.0
    dif, spec, tint = (torch.rand(B, S, 3, generator=g).to(dtype) for _ in range(3))
    z = torch.cumsum(torch.rand(B, S, generator=g) * 0.1 + 0.01, 1).to(dtype)
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((B, 1), 1e-6, dtype=dtype)], 1)
    rays_d = (torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) * (0.5 + torch.rand(B, 1, generator=g))).to(dtype)
    return sigma, dif, spec, tint, z, dists, rays_d


@pytest.mark.parametrize("S", [24, 33, 64, 128, 200])
@pytest.mark.parametrize("infinity", [False, True])
def test_composite_forward_and_backward_against_torch_float64(S, infinity):
    import scanerf_amd  # noqa: F401
    from scanerf_amd import render
    B = 257
    ins = _inputs(B, S, 100 + S)
    ref_in = [t.double().requires_grad_(i in (0, 1, 2, 3, 6)) for i, t in enumerate(ins)]
    ref = _torch_composite(*ref_in, infinity)
    dev_in = [t.to(DEV).requires_grad_(i in (0, 1, 2, 3, 6)) for i, t in enumerate(ins)]
    out, w = render.composite_rays(*dev_in, infinity)
    cols = {"rgb": render.RGB, "depth": render.DEPTH, "T_left": render.T_LEFT, "diffuse": render.DIFFUSE, "specular": render.SPECULAR,
            "tint": render.TINT, "w_spec2": render.W_SPEC2}
    for k, c in cols.items():
        np.testing.assert_allclose(out[:, c].detach().cpu().numpy(), ref[k].detach().numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
    np.testing.assert_allclose(w.detach().cpu().numpy(), ref["weights"].detach().numpy(), rtol=2e-5, atol=1e-7)
    # one scalar of random upstream gradients on every output (clamp active on some rays: rgb sums above 1 exist)
    g = torch.Generator().manual_seed(7)
    go = torch.randn(B, 16, generator=g)
    go[:, 15] = 0.0
    gw = torch.randn(B, S, generator=g) * 0.1
    assert float((ref["diffuse"] + ref["specular"]).max()) > 1.0
    loss = (out * go.to(DEV)).sum() + (w * gw.to(DEV)).sum()
    loss.backward()
    ref_out = torch.cat([ref["rgb"], ref["depth"][:, None], ref["T_left"][:, None], ref["diffuse"], ref["specular"], ref["tint"],
                         ref["w_spec2"][:, None], torch.zeros(B, 1, dtype=torch.float64)], 1)
    ((ref_out * go.double()).sum() + (ref["weights"] * gw.double()).sum()).backward()
    for i, name in ((0, "sigma"), (1, "diffuse"), (2, "specular"), (3, "tint"), (6, "rays_d")):
        a, b = dev_in[i].grad.cpu().double(), ref_in[i].grad
        scale = float(b.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 3e-5 * scale, (name, float((a - b).abs().max()) / scale)


def test_composite_is_bit_reproducible_and_handles_empty_batches():
    import scanerf_amd  # noqa: F401
    from scanerf_amd import render
    ins = [t.to(DEV) for t in _inputs(1000, 128, 3)]
    a = render.composite_rays(*ins, True)
    for _ in range(5):
        b = render.composite_rays(*ins, True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    e = [t[:0] for t in ins]
    out, w = render.composite_rays(*e, False)
    assert out.shape == (0, 16) and w.shape == (0, 128)


def test_decoder_on_separate_feature_and_direction_tensors_equals_the_concatenated_form():
    """network.ShallowMLP.forward_parts / decoder_op.decoder_apply_parts: the same kernels addressed with row strides 32 and 3
    instead of 35 + 35: outputs and every gradient (features, directions, blob) bit-equal to the op on torch.cat([features, dirs])."""
    import scanerf_amd  # noqa: F401
    from scanerf_amd import decoder_op, network
    g = torch.Generator().manual_seed(2)
    N = (1 << 15) + 37
    feats = torch.randn(N, 32, generator=g).to(DEV)
    dirs = torch.randn(N, 3, generator=g).to(DEV)
    blob = network.xavier_blob(3).to(DEV)
    wf = network.weight_feature(20000, DEV)
    gs = [torch.randn(N, k, generator=g).to(DEV) for k in (1, 3, 3, 3)]
    res = []
    for parts in (True, False):
        f, d, b = feats.clone().requires_grad_(True), dirs.clone().requires_grad_(True), blob.clone().requires_grad_(True)
        outs = decoder_op.decoder_apply_parts(f, d, b, wf) if parts else decoder_op.decoder_apply(torch.cat([f, d], -1), b, wf)
        sum((o * gg).sum() for o, gg in zip(outs, gs)).backward()
        res.append([o.detach() for o in outs] + [f.grad, d.grad, b.grad])
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("infinity", [False, True])
def test_composite_against_the_reference_own_autograd_golden_g19(golden, infinity):
    """G19: HashGrid.cal_integrate_weight + accumulate + the l2_reg_specular sum run by the REFERENCE's methods under torch autograd
    (hashgrid/__init__.py:344-366, :564-574, :591-594): its outputs and its gradients of a random functional of every output --
    against scanerf_composite_forward / _backward (f32 both sides: 1e-4 of the largest gradient)."""
    import scanerf_amd  # noqa: F401
    from scanerf_amd import render
    g = golden("g19_composite_grads")
    t = "inf%d_" % infinity
    G = lambda k: torch.from_numpy(g[t + k]).to(DEV)
    ins = [G(k).requires_grad_(k not in ("z_vals", "dists")) for k in ("sigma", "diffuse", "specular", "tint", "z_vals", "dists", "rays_d")]
    out, w = render.composite_rays(*ins, infinity)
    d = render.render_batch_rays_dict(out, w, True)
    for key, ref in (("rgb", "rgb"), ("depth", "depth"), ("T_left", "T_left"), ("diffuse", "diffuse_out"), ("specular", "specular_out"),
                     ("tint", "tint_out"), ("weights", "weights"), ("l2_reg_specular", "l2_reg_specular")):
        np.testing.assert_allclose(d[key].detach().cpu().numpy(), g[t + ref], rtol=1e-4, atol=1e-6, err_msg=key)
    loss = ((d["rgb"] * G("cw_rgb")).sum() + (d["depth"] * G("cw_depth")).sum() + (d["T_left"] * G("cw_T")).sum()
            + (d["diffuse"] * G("cw_dif")).sum() + (d["specular"] * G("cw_spec")).sum() + (d["tint"] * G("cw_tint")).sum()
            + 0.1 * (d["weights"] * G("cw_w")).sum() + 0.37 * d["l2_reg_specular"])
    loss.backward()
    for i, key in ((0, "g_sigma"), (1, "g_diffuse"), (2, "g_specular"), (3, "g_tint"), (6, "g_rays_d")):
        a, b = ins[i].grad.cpu(), torch.from_numpy(g[t + key])
        scale = float(b.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 1e-4 * scale, (key, float((a - b).abs().max()) / scale)
