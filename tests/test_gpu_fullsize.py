"""The headline configuration (BASELINE.json configs[1]: 65 536 rays x 128 samples, L=16, T=2^19) checked for
correctness AT ITS OWN SIZE: the fused training path of bench.py (plan -> forward -> backward emits the scatter
records -> accumulate) against (i) the reference-style atomic scatter on the same feature gradients, (ii) a
conservation law of the trilinear scatter, (iii) the exact-f32 decoder arithmetic, and (iv) torch autograd through
the CPU oracle on ray slices pushed through the FULL-SIZE launch (every other ray masked by ray_valid).

This is where 32-bit record offsets, workspace sizes (8.6 GB of records) and T-dependent bucket geometry live.
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
B, S_, LOG2_T = 65536, 128, 19


@pytest.fixture(scope="module")
def full():
    """One tile + one ray batch at the benchmark's size, the forward run once."""
    import scanerf_amd  # noqa: F401
    from scanerf_amd import network, render
    from scanerf_amd.tile_model import TileModel
    torch.manual_seed(19)
    m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=LOG2_T, seed=7)
    with torch.no_grad():
        m.features.mul_(300.0)  # xavier std of a 2^19 table leaves sigma ~ softplus(0): make the volume non-trivial
    o = torch.rand(B, 3, device=DEV) * 8 - 4
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1) * (0.5 + torch.rand(B, 1, device=DEV))
    z, dist = m.sample(o, d, S_)
    assert bool(torch.all(z != -1))  # fully occupied sampler grid, origins inside the tile
    step = 20000
    wf = network.weight_feature(step, DEV)
    m.packed.pack(m.decoder.blob(), wf)
    box = (m.min_bbox.tolist(), m.bbox_size.tolist(), render.FORE, False)
    gout = torch.zeros(B, 16, device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(3)
    gout[:, 0:3] = torch.randn(B, 3, device=DEV, generator=gen) / B
    gout[:, 3] = torch.randn(B, device=DEV, generator=gen) / B
    gout[:, 4] = torch.randn(B, device=DEV, generator=gen) / B
    gout[:, 14] = 0.37 / B
    return dict(m=m, o=o, d=d, z=z, dist=dist, wf=wf, box=box, gout=gout, step=step)


def _fused(full, valid, arith=None, want_dfeat=False):
    """plan -> forward (tile_T, xstash) -> backward (emits records) -> accumulate, as tile_model.train_step_fused does."""
    from scanerf_amd import render
    m, o, d, z, dist = (full[k] for k in ("m", "o", "d", "z", "dist"))
    render.set_arith(arith or render.DEFAULT_ARITH)
    try:
        m.packed.pack(m.decoder.blob(), full["wf"])
        T = m.features.shape[1]
        tile_T = torch.empty(B, (S_ + 15) // 16, device=DEV)
        xs = torch.empty(B * S_, 32, device=DEV)
        out, w = render.render_forward(o, d, z, dist, m.features, m.resolution, m.packed, *full["box"], ray_valid=valid,
                                       want_weights=True, tile_T=tile_T, xstash=xs)
        assert render.scatter_supported(B, S_, T)
        ws = render.scatter_plan(o, d, z, m.resolution, T, *full["box"], ray_valid=valid)
        gtab = torch.zeros_like(m.features)
        dfeat, gblob = render.render_backward(o, d, z, dist, m.features, m.resolution, m.packed, full["wf"], *full["box"], out,
                                              tile_T, full["gout"], ray_valid=valid, xstash=xs, scatter=(ws, gtab),
                                              want_dfeat=want_dfeat)
        render.scatter_accumulate(ws, gtab, B, S_)
        torch.cuda.synchronize()
        return out, w, dfeat, gtab, gblob
    finally:
        render.set_arith(render.DEFAULT_ARITH)


@pytest.mark.parametrize("arith,tol", [("h3", 1e-4), ("t16", 5e-4), ("t16s", 1e-4)])
def test_full_size_fused_scatter_vs_atomics_and_conservation(full, arith, tol):
    """All 65 536 rays valid: 5.4e8 records (h3: 16-byte records, 8.6 GB, byte offsets past 2^32; t16: 8-byte records with
    13-bit significands, scatter_common.h).  The table gradient of the fused path equals the reference-style atomic scatter
    (hashgrid_bg_kernel.cu:196-201) of the SAME dfeat, and per (level, feature) the sum over table entries equals the sum
    over samples of dfeat (the 8 trilinear weights sum to one)."""
    from scanerf_amd.hashgrid.lib.HASHGRID import embedding_bg_backward_cuda
    m, o, d, z = (full[k] for k in ("m", "o", "d", "z"))
    out, w, dfeat, gtab, gblob = _fused(full, None, arith, want_dfeat=True)
    assert torch.isfinite(out).all() and torch.isfinite(gtab).all() and torch.isfinite(gblob).all()
    # (ii) conservation, in float64
    lhs = gtab.double().sum(1).cpu().numpy()            # [16, 2]
    rhs = dfeat.double().sum(1).cpu().numpy()           # [16, 2]
    mag = dfeat.double().abs().sum(1).cpu().numpy()
    np.testing.assert_allclose(lhs, rhs, rtol=0, atol=float(2e-6 * mag.max()))
    # (i) the atomic kernel on the same dfeat
    pts = (((o[:, None, :] + z[:, :, None] * d[:, None, :]).reshape(-1, 3) - m._min_dev) / m._size_dev * 4.0 - 2.0).contiguous()
    g1 = torch.zeros_like(m.features)
    gin = dfeat.permute(1, 0, 2).contiguous()  # [N, 16, 2]
    from scanerf_amd.hashgrid.lib import HASHGRID as _HG
    _HG.TABLE_GRAD_ROUTE = "atomics"
    try:
        embedding_bg_backward_cuda(pts, gin, None, g1, m.features, m.resolution)
    finally:
        _HG.TABLE_GRAD_ROUTE = "binned"
    torch.cuda.synchronize()
    sc = float(g1.abs().max())
    assert sc > 0
    err = float((gtab - g1).abs().max()) / sc
    print(f"full size, {arith}: fused scatter vs atomics max err {err:.3e} of max, relative L2 {float((gtab - g1).norm() / g1.norm()):.3e}")
    assert err < tol, f"fused scatter vs atomics: {err:.3e} of max"
    # every table entry the atomics touched is touched by the fused path and vice versa (up to exact cancellations)
    nz1, nz2 = int((g1 != 0).sum()), int((gtab != 0).sum())
    assert abs(nz1 - nz2) <= 1e-4 * nz1, (nz1, nz2)


def test_full_size_h3_and_t16_vs_f32_arith(full):
    """(iii) the split-f16 decoder arithmetic against the exact-f32 MFMA kernels at full size: per-ray outputs to 1e-4
    (north_star); decoder and table gradients to 1e-4 of their maxima for h3 (every product split), to 1e-3 for t16
    (gradient products on one f16 MFMA per term; the measured figure is printed)."""
    out_f, w_f, _, gtab_f, gblob_f = _fused(full, None, "f32")
    for arith, tol in (("h3", 1e-4), ("t16", 1e-3), ("t16s", 1e-4)):
        out_h, w_h, _, gtab_h, gblob_h = _fused(full, None, arith)
        np.testing.assert_allclose(out_h[:, :5].cpu().numpy(), out_f[:, :5].cpu().numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(w_h.cpu().numpy(), w_f.cpu().numpy(), rtol=1e-4, atol=1e-7)
        for a, b, name in ((gblob_h, gblob_f, "decoder"), (gtab_h, gtab_f, "table")):
            sc = float(b.abs().max())
            err = float((a - b).abs().max()) / sc
            rel = float((a - b).norm() / b.norm())
            print(f"full size, {arith} vs f32, {name} gradient: max err {err:.3e} of max, relative L2 {rel:.3e}")
            assert err < tol, f"{name} gradient {arith} vs f32: {err:.3e} of max"


@pytest.mark.parametrize("first", [0, 21845, 43690, 63488])
def test_full_size_slice_vs_oracle_autograd(full, first):
    """(iv) 2 048 consecutive rays of the batch pushed through the FULL-SIZE launches with every other ray masked: outputs,
    table gradient and decoder gradient against torch autograd through the oracle on those rays."""
    n = 2048
    m, o, d, z, dist, gout = (full[k] for k in ("m", "o", "d", "z", "dist", "gout"))
    valid = torch.zeros(B, dtype=torch.bool, device=DEV)
    valid[first:first + n] = True
    out, w, _, gtab, gblob = _fused(full, valid)
    sl = slice(first, first + n)
    F = m.features.detach().cpu().clone().requires_grad_(True)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.decoder.ref_state_dict().items()}
    mn, sz = m.min_bbox, m.bbox_size
    ref = O.render_batch_rays(o[sl].cpu(), d[sl].cpu(), z[sl].cpu(), dist[sl].cpu(), F, m.resolution.cpu(), sd, O.TRAIN,
                              lambda x: O.contract_fore(x, mn, sz), full["step"])
    go = gout[sl].cpu()
    loss = (ref["rgb"] * go[:, 0:3]).sum() + (ref["depth"][:, 0] * go[:, 3]).sum() + (ref["T_left"] * go[:, 4]).sum() + \
        float(go[0, 14]) * ref["l2_reg_specular"] * (3 * n)
    loss.backward()
    got = out[sl].cpu().numpy()
    # 1e-4 relative (north_star) + an absolute floor: every weight carries the ~6e-8 ABSOLUTE rounding of
    # alpha = 1 - exp(-sigma delta) (hashgrid/__init__.py:352, a cancellation for small sigma delta) in the oracle and here
    # alike, and an output sums 128 of them: differences of a few 1e-6 between two correct f32 evaluations of a colour in
    # [0,1] are that noise (first run of this test: 2 of 6 144 colours off by 1.6e-6 at values ~5e-4)
    np.testing.assert_allclose(got[:, 0:3], ref["rgb"].detach().numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(got[:, 3], ref["depth"][:, 0].detach().numpy(), rtol=1e-4, atol=5e-5)  # sum of w * z, z ~ 10
    np.testing.assert_allclose(got[:, 4], ref["T_left"].detach().numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(w[sl].cpu().numpy(), ref["weights"][..., 0].detach().numpy(), rtol=1e-4, atol=2e-7)
    rest = torch.ones(B, dtype=torch.bool)
    rest[sl] = False
    assert torch.all(out[rest.to(DEV)][:, :4] == 0)
    # gradients: the default backward (arith t16) does its gradient products on one f16 MFMA per term: rounding noise of
    # ~5e-4 of the largest element on every element, bounded here as 2e-3 of the maximum and 2e-3 in relative L2
    # (arith h3: 2e-3 relative with a floor of 2e-5 of the maximum, tests/test_gpu_parity.py)
    from scanerf_amd import render
    tol = dict(rtol=2e-3, atol=2e-5) if render.DEFAULT_ARITH in ("h3", "t16s") else dict(rtol=2e-3, atol=2e-3)
    gF, gT = F.grad.numpy(), gtab.cpu().numpy()
    fs = np.abs(gF).max()
    gb_ref, gB = O.pack_blob({k: v.grad for k, v in sd.items()}).numpy(), gblob.cpu().numpy()
    bs = np.abs(gb_ref).max()
    l2t, l2b = np.linalg.norm(gT - gF) / np.linalg.norm(gF), np.linalg.norm(gB - gb_ref) / np.linalg.norm(gb_ref)
    print(f"slice {first}: table gradient max err {np.abs(gT - gF).max() / fs:.2e} of max, rel L2 {l2t:.2e}; "
          f"decoder gradient max err {np.abs(gB - gb_ref).max() / bs:.2e} of max, rel L2 {l2b:.2e}")
    np.testing.assert_allclose(gT / fs, gF / fs, **tol)
    np.testing.assert_allclose(gB / bs, gb_ref / bs, **tol)
    assert l2t < 2e-3 and l2b < 2e-3


def test_reference_default_table_size_scatter_adam_conservation_and_formats():
    """config/default.yaml:2 ships T = 2^24 with 16 384 rays x 128 samples: above 2^21 entries the table gradient goes through
    the stand-alone binned scatter ending in the sparse Adam (scanerf_embedding_bg_backward_binned_adam).  At that size, on one
    random point set with level-major gradients spanning 2^-9 .. 1: per (level, feature) the sum of the first moments x 10 (= the
    table gradient after one step from zero moments) equals the sum of the samples' gradients (the 8 trilinear weights sum to
    one) -- for the exact 16-byte records and for the 12-byte records that go behind the t16s backward; the two formats agree
    entry by entry to the records' rounding; the entries that moved are the entries with a gradient; nothing overflowed."""
    import scanerf_amd  # noqa: F401
    from scanerf_amd import render
    N, L, T = 16384 * 128, 16, 1 << 24
    gen = torch.Generator(device=DEV).manual_seed(24)
    res = torch.from_numpy(O.level_resolutions(torch.tensor([32, 32, 32]), torch.tensor([2048, 2048, 2048])).numpy()).to(DEV)
    pts = (torch.rand(N, 3, device=DEV, generator=gen) * 4 - 2).contiguous()
    dfe = (torch.randn(L, N, 2, device=DEV, generator=gen) * torch.exp2(-9 * torch.rand(L, N, 1, device=DEV, generator=gen))).contiguous()
    want = dfe.double().sum(1)                                   # [L][2]
    scale = dfe.double().abs().sum(1)
    m1s = {}
    for fmt in (0, 2):
        params, m1, m2, over = (torch.zeros(L, T, 2, device=DEV) for _ in range(4))
        render.scatter_table_grad_adam(pts, dfe, res, params, m1, m2, 1e-2, 0.9, 0.99, 1e-15, 0, overflow_grad=over, compact_records=fmt)
        torch.cuda.synchronize()
        assert not bool(over.any()), fmt
        got = m1.double().sum(1) * 10.0
        err = float(((got - want).abs() / scale).max())
        assert err <= 2e-6, (fmt, err)                           # (f32 moments: 0.1 g rounded once per entry)
        assert bool(torch.equal(m1 != 0, params != 0))           # sparse Adam: an entry moves iff it has a gradient
        m1s[fmt] = m1
        del params, m2, over
    diff = float((m1s[0] - m1s[2]).abs().max()) / float(m1s[0].abs().max())
    assert diff <= 4e-6, diff
    frac = float((m1s[0][8:] != 0).float().mean())               # hashed levels: 2.1e6 samples x 8 corners over 1.7e7 entries
    assert 0.3 < frac < 0.8, frac


def test_configs0_L8_render_on_the_hip_ops_path():
    """BASELINE.json configs[0]: single 8 m^3 tile, 4 096 random rays x 64 samples, L=8 hash grid (decoder in_channel 16),
    forward only -- at its own size on the HIP path.  The fused kernels hard-code 16 levels like the reference
    (hashgrid/__init__.py:62), so this configuration runs the reference's own structure: HIP occupancy sampler -> HIP hash
    encoder at L=8 (both through the binding names) -> torch decoder and compositing on the GPU
    (TileModel.render_fore_ops).  Checked against the oracle's render_batch_rays at 1e-4 (north_star)."""
    import scanerf_amd  # noqa: F401
    from scanerf_amd.tile_model import TileModel
    torch.manual_seed(0)
    Bc, Sc, L = 4096, 64, 8
    m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=19, seed=11, n_levels=L)
    assert tuple(m.features.shape) == (L, 2 ** 19, 2) and m.decoder.in_channel == 16
    with torch.no_grad():
        m.features.mul_(300.0)
    o = torch.rand(Bc, 3, device=DEV) * 8 - 4
    d = torch.nn.functional.normalize(torch.randn(Bc, 3, device=DEV), dim=-1) * (0.5 + torch.rand(Bc, 1, device=DEV))
    step = 2500  # inside the coarse-to-fine schedule: the level mask is not all ones
    with torch.no_grad():
        out = m.render_fore_ops(o, d, Sc, step, train=True)
    assert bool(out["valid"].all())
    # the fused path must refuse this table instead of mis-reading it
    with pytest.raises(RuntimeError):
        m.render_fore_fused(o, d, Sc, step)
    tile = O.Tile([-4, -4, -4], [8, 8, 8], log2_T=19, n_levels=L)
    assert np.array_equal(tile.res.numpy(), m.resolution.cpu().numpy())
    z_ref, d_ref = O.sample_points_grid(o.cpu().numpy(), d.cpu().numpy(), tile.occ_corner, tile.occ_size, tile.occ, tile.log2dim, Sc)
    z, dist = m.sample(o, d, Sc)
    assert np.array_equal(z.cpu().numpy(), z_ref) and np.array_equal(dist.cpu().numpy(), d_ref)
    sd = {k: v.detach().cpu() for k, v in m.decoder.ref_state_dict().items()}
    assert tuple(sd["Spatial_MLP.mlp.0.weight"].shape) == (64, 16)
    with torch.no_grad():
        ref = O.render_batch_rays(o.cpu(), d.cpu(), torch.from_numpy(z_ref), torch.from_numpy(d_ref), m.features.detach().cpu(),
                                  tile.res, sd, O.TRAIN, lambda x: O.contract_fore(x, tile.min_bbox, tile.bbox_size), step)
    for key in ("rgb", "diffuse", "specular"):
        np.testing.assert_allclose(out[key].cpu().numpy(), ref[key].numpy(), rtol=1e-4, atol=1e-6, err_msg=key)
    np.testing.assert_allclose(out["depth"].cpu().numpy(), ref["depth"][:, 0].numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(out["T_left"].cpu().numpy(), ref["T_left"].numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(out["weights"].cpu().numpy(), ref["weights"][..., 0].numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(float(out["l2_reg_specular"]), float(ref["l2_reg_specular"]), rtol=1e-4)
    assert float(ref["weights"].sum(1).max()) > 0.5  # the volume is not empty


# ------------------------------------------------------------------ BASELINE.json configs[2] at its own size
@pytest.fixture(scope="module")
def cfg2():
    """configs[2] as bench.py --workload configs2 builds it: 65 536 rays x 128 samples, T = 2^19,
    sphere-shell occupancy (r = 3 m, 0.5 m thick) at log2dim 7, bf16 gather table over the fp32 master, fused sparse Adam."""
    import scanerf_amd  # noqa: F401
    from scanerf_amd.tile_model import TileModel, sphere_shell_occupancy
    torch.manual_seed(23)

    def make():
        m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=LOG2_T, seed=5, sampler_log2dim=7, table_dtype=torch.bfloat16)
        m.set_occupancy(sphere_shell_occupancy(m, 3.0, 0.5))
        with torch.no_grad():
            m.features.mul_(300.0)
        return m

    o = torch.rand(B, 3, device=DEV) * 8 - 4
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1) * (0.5 + torch.rand(B, 1, device=DEV))
    tgt = torch.rand(B, 3, device=DEV)
    return dict(make=make, o=o, d=d, tgt=tgt, step=20000)


def _cfg2_gradients(m, c, valid_extra=None):
    """sampler -> valid-ray compaction -> forward on the bf16 gather table -> photometric loss -> backward emitting the records
    -> accumulate into a gradient table (the pieces of train_step_fused, with the gradient table and dfeat kept)."""
    from scanerf_amd import network, render
    z, dist = m.sample(c["o"], c["d"], S_)
    valid = render.ray_valid(z)
    n, co, cd, ct, cz, cdist = render.compact_rays(valid, c["o"], c["d"], c["tgt"], z, dist)
    co, cd, ct, cz, cdist = (t[:n].contiguous() for t in (co, cd, ct, cz, cdist))
    wf = network.weight_feature(c["step"], DEV)
    m.packed.pack(m.decoder.blob(), wf)
    box = (m.min_bbox.tolist(), m.bbox_size.tolist(), render.FORE, False)
    table = m.gather_table()
    assert table.dtype == torch.bfloat16
    T = m.features.shape[1]
    tile_T = torch.empty(n, (S_ + 15) // 16, device=DEV)
    xs = torch.empty(n * S_, 32, device=DEV)
    rv = valid_extra(n) if valid_extra else None
    out, w = render.render_forward(co, cd, cz, cdist, table, m.resolution, m.packed, *box, ray_valid=rv, want_weights=True, tile_T=tile_T, xstash=xs)
    loss, gout = render.photometric_loss_grad(out, ct, rv, 0.01)
    ws = render.scatter_plan(co, cd, cz, m.resolution, T, *box, ray_valid=rv)
    gtab = torch.zeros_like(m.features)
    dfeat, gblob = render.render_backward(co, cd, cz, cdist, table, m.resolution, m.packed, wf, *box, out, tile_T, gout, ray_valid=rv,
                                          xstash=xs, scatter=(ws, gtab), want_dfeat=True)
    render.scatter_accumulate(ws, gtab, n, S_)
    torch.cuda.synchronize()
    return dict(n=n, valid=valid, o=co, d=cd, tgt=ct, z=cz, dist=cdist, out=out, w=w, loss=loss, gout=gout, dfeat=dfeat, gtab=gtab, gblob=gblob, wf=wf)


def test_configs2_full_size_under_the_default_arithmetic(cfg2):
    """configs[2] AT ITS OWN SIZE under the default arithmetic (t16s; round 3 tested the configuration at 4 096 x 64, T = 2^14,
    pinned to h3).  (a) compacting the valid rays == masking them inside the kernels; (b) per (level, feature) the table
    gradient's sum over entries == the feature gradients' sum over samples; (c) the fused sparse Adam moves an entry iff it has a
    gradient, and the resident bf16 gather table follows the fp32 master; (d) a 2 048-ray slice of the compacted batch, pushed
    through the full-size launches with the other rays masked, against torch autograd through the oracle on the bf16-rounded table."""
    from scanerf_amd import render
    from scanerf_amd.tile_model import train_step_fused
    assert render.DEFAULT_ARITH == "t16s" and render.arith_name() == "t16s"
    c = cfg2
    # ---- (b) conservation + what has a gradient
    m = c["make"]()
    g = _cfg2_gradients(m, c)
    frac = g["n"] / B
    assert 0.2 < frac < 0.7, frac            # the shell is met by ~40 % of a random batch: compaction really changes the launch
    assert torch.isfinite(g["gtab"]).all() and torch.isfinite(g["gblob"]).all()
    lhs, rhs = g["gtab"].double().sum(1).cpu().numpy(), g["dfeat"].double().sum(1).cpu().numpy()
    mag = g["dfeat"].double().abs().sum(1).cpu().numpy()
    np.testing.assert_allclose(lhs, rhs, rtol=0, atol=float(2e-6 * mag.max()))
    # ---- (a) + (c): one whole training step, compacted and masked, from the same state
    res = {}
    for name, kw in (("compact", {"compact_rays": True}), ("masked", {"compact_rays": False})):
        mm = c["make"]()
        opt = torch.optim.Adam(mm.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
        before = mm.features.detach().clone()
        loss = float(train_step_fused(mm, opt, c["o"], c["d"], c["tgt"], S_, c["step"], **kw))
        torch.cuda.synchronize()
        res[name] = (loss, mm.features.detach().clone(), mm.decoder.blob().detach().clone(), mm._half_table.clone())
        moved = mm.features.detach() != before
        if name == "compact":
            np.testing.assert_allclose(loss, float(g["loss"][0]), rtol=1e-6)
            # same rays, same arithmetic, same fixed-point image: the entries the epilogue moved are the entries with a gradient
            # (an entry whose gradient is below ~1e-19 can have a second moment that underflows: allowed to stay, never the reverse)
            has = g["gtab"] != 0
            assert not bool((moved & ~has).any()), int((moved & ~has).sum())
            assert int((has & ~moved).sum()) <= 1e-6 * has.numel(), (int((has & ~moved).sum()), int(has.sum()))
            assert 0.05 < float(moved[8:].float().mean()) < 0.99   # (hashed levels: 27 000 rays x 128 samples x 8 corners over 5e5 entries)
        # the resident bf16 gather table is the fp32 master rounded, everywhere
        assert bool(torch.equal(mm._half_table, mm.features.detach().to(torch.bfloat16)))
        del mm, opt
    np.testing.assert_allclose(res["compact"][0], res["masked"][0], rtol=2e-5)
    # "masked" differs from "compact" in which rays share a workgroup (the backward's power-of-two gradient scale is per
    # workgroup) and in the record order: updated tables agree except where Adam turns rounding noise on a ~zero gradient into
    # an lr-sized step
    dfe = (res["compact"][1] - res["masked"][1]).abs() / res["masked"][1].abs().max()
    db = float((res["compact"][2] - res["masked"][2]).abs().max() / res["masked"][2].abs().max())
    n_off = int((dfe > 2e-4).sum())
    print(f"configs[2] full size: {g['n']} of {B} rays valid; compact vs masked: {n_off} of {dfe.numel()} entries differ by more than 2e-4 of max, "
          f"largest {float(dfe.max()):.2e}; decoder {db:.2e}")
    assert n_off <= 1e-5 * dfe.numel() and float(dfe.max()) < 0.05 and db < 2e-4, (n_off, float(dfe.max()), db)
    # ---- (d) slice vs the oracle's autograd
    n_sl, first = 2048, g["n"] // 3

    def only_slice(n):
        v = torch.zeros(n, dtype=torch.bool, device=DEV)
        v[first:first + n_sl] = True
        return v

    s = _cfg2_gradients(m, c, only_slice)
    sl = slice(first, first + n_sl)
    F = m.gather_table().float().cpu().clone().requires_grad_(True)       # the bf16-rounded table the kernels gathered from
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.decoder.ref_state_dict().items()}
    ref = O.render_batch_rays(s["o"][sl].cpu(), s["d"][sl].cpu(), s["z"][sl].cpu(), s["dist"][sl].cpu(), F, m.resolution.cpu(), sd, O.TRAIN,
                              lambda x: O.contract_fore(x, m.min_bbox, m.bbox_size), c["step"])
    loss_ref = torch.nn.functional.mse_loss(ref["rgb"], s["tgt"][sl].cpu()) + 0.01 * ref["l2_reg_specular"]
    loss_ref.backward()
    np.testing.assert_allclose(float(s["loss"][0]), float(loss_ref.detach()), rtol=1e-4)
    got = s["out"][sl].cpu().numpy()
    np.testing.assert_allclose(got[:, 0:3], ref["rgb"].detach().numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(got[:, 4], ref["T_left"].detach().numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(s["w"][sl].cpu().numpy(), ref["weights"][..., 0].detach().numpy(), rtol=1e-4, atol=2e-7)
    gF, gT = F.grad.numpy(), s["gtab"].cpu().numpy()
    gb_ref, gB = O.pack_blob({k: v.grad for k, v in sd.items()}).numpy(), s["gblob"].cpu().numpy()
    fs, bs = np.abs(gF).max(), np.abs(gb_ref).max()
    l2t, l2b = np.linalg.norm(gT - gF) / np.linalg.norm(gF), np.linalg.norm(gB - gb_ref) / np.linalg.norm(gb_ref)
    print(f"configs[2] slice: table gradient max err {np.abs(gT - gF).max() / fs:.2e} of max, rel L2 {l2t:.2e}; decoder max err "
          f"{np.abs(gB - gb_ref).max() / bs:.2e}, rel L2 {l2b:.2e}")
    np.testing.assert_allclose(gT / fs, gF / fs, rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(gB / bs, gb_ref / bs, rtol=2e-3, atol=2e-5)
    assert l2t < 1e-4 and l2b < 1e-4


# ------------------------------------------------------------------ BASELINE.json configs[4], the legs one GPU runs
def test_configs4_four_resident_tiles_round_robin_equals_each_alone():
    """configs[4] keeps 4 tiles resident per GPU and steps them round-robin (bench.py --tiles-per-gpu 4; the reference swaps them
    through host memory, tile.py:574-636).  The tiles share every scratch buffer (plan workspace, record stream, packed-decoder
    workspace per model): two rounds over the four tiles must leave every tile EXACTLY where two steps of that tile alone leave
    it -- tables, Adam moments, decoders, bit for bit -- at the benchmark's size under the default arithmetic."""
    import scanerf_amd  # noqa: F401
    from scanerf_amd.tile_model import TileModel, train_step_fused
    torch.manual_seed(41)
    o = torch.rand(B, 3, device=DEV) * 8 - 4
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1) * (0.5 + torch.rand(B, 1, device=DEV))
    tgt = torch.rand(B, 3, device=DEV)

    def make(t):
        m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=LOG2_T, seed=100 + t)
        return m, torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)

    def state(m):
        return [m.features.detach().clone(), m.exp_avg.clone(), m.exp_avg_sq.clone(), m.decoder.blob().detach().clone()]

    rr = [make(t) for t in range(4)]
    for i in range(8):
        m, opt = rr[i % 4]
        train_step_fused(m, opt, o, d, tgt, S_, 20000 + i // 4)
    torch.cuda.synchronize()
    for t in range(4):
        m, opt = make(t)
        init = m.features.detach().clone()
        for k in range(2):
            train_step_fused(m, opt, o, d, tgt, S_, 20000 + k)
        torch.cuda.synchronize()
        for a_, b_, what in zip(state(rr[t][0]), state(m), ("table", "exp_avg", "exp_avg_sq", "decoder")):
            assert torch.equal(a_, b_), (t, what, float((a_ - b_).abs().max()))
        assert float((m.features.detach() - init).abs().max()) > 0
        del m, opt


def test_configs4_full_hd_frame_is_bit_reproducible_and_a_crop_matches_the_oracle_loop(tmp_path):
    """configs[4]'s render leg as bench.py --workload configs4-render builds it: 4 tiles (f16 tables, T = 2^19, shell occupancy
    at log2dim 7) + blended backgrounds, one 1920 x 1080 view, 128 + 128 samples.  Two renders of the frame agree bit for bit;
    a 24 x 32 crop across a shell's silhouette equals the oracle's restatement of rendering.py:286-544 on those rays (1e-4)."""
    import scanerf_amd  # noqa: F401
    from scanerf_amd import renderer as R
    from scanerf_amd import tile_model as tm
    from test_gpu_render_time import oracle_render_loop
    H, W, ntile = 1080, 1920, 4
    tiles = []
    for t in range(ntile):
        m = tm.TileModel([-4.0 * ntile + 8.0 * t, -4, -4], [8, 8, 8], DEV, log2_T=LOG2_T, seed=t, sampler_log2dim=7)
        m.set_occupancy(tm.sphere_shell_occupancy(m, 3.0, 0.5))
        with torch.no_grad():
            m.features.mul_(300.0)
        R.export_tile(str(tmp_path / f"tile{t}"), m)
        tiles.append(R.load_tile(str(tmp_path / f"tile{t}")))
        del m
    rend = R.TileSetRenderer(DEV, tiles)
    K = np.float32([1600.0, 0, W / 2, 0, 1600.0, H / 2, 0, 0, 1])
    c2w = np.float32([[1.0, 0, 0, 0.0], [0, 1, 0, 0.5], [0, 0, 1, -14.0]])
    f1 = rend.render(H, W, K, c2w, num_sample=128, num_bg_sample=128)
    f2 = rend.render(H, W, K, c2w, num_sample=128, num_bg_sample=128)
    for a_, b_ in zip(f1, f2):
        assert torch.equal(a_, b_)
        assert torch.isfinite(a_).all()
    dif, spec, depth, transp = f1
    opaque = (transp[..., 0] < 0.5)
    assert 0.02 < float(opaque.float().mean()) < 0.5
    # a crop across a silhouette: the row with the most opaque pixels, centred on its first opaque pixel
    r0 = int(opaque.sum(1).argmax())
    c0 = int(opaque[r0].float().argmax())
    r0, c0 = min(max(r0 - 12, 0), H - 24), min(max(c0 - 16, 0), W - 32)
    o_all, d_all = rend.compute_rays(H, W, K, c2w)
    idx = (torch.arange(r0, r0 + 24, device=DEV)[:, None] * W + torch.arange(c0, c0 + 32, device=DEV)[None, :]).reshape(-1)
    o, d = o_all[idx].cpu().numpy(), d_all[idx].cpu().numpy()
    ref = oracle_render_loop(rend, o, d, 128, 128)
    assert ref["T"].min() < 0.5 < ref["T"].max(), "the crop must contain both opaque and empty pixels"
    n = idx.numel()
    crop = lambda a: a.reshape(H * W, -1)[idx].cpu().numpy()
    np.testing.assert_allclose(crop(transp), ref["T"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(crop(dif), ref["dif"], rtol=1e-4, atol=3e-6)
    np.testing.assert_allclose(crop(spec), ref["spec"], rtol=1e-4, atol=3e-6)
    np.testing.assert_allclose(crop(depth), ref["depth"], rtol=1e-4, atol=1e-3 * max(1.0, float(np.abs(ref["depth"]).max()) / 100))
    img_a = np.clip(crop(dif) + crop(spec), 0, 1).reshape(24, 32, 3) * 255
    img_b = np.clip(ref["dif"] + ref["spec"], 0, 1).reshape(24, 32, 3) * 255
    assert O.psnr(img_a, img_b) > 80.0 and n == 768


def test_configs1_three_routes_agree_at_full_size():
    """configs[1] at its own size through the reference-shaped classes (round 5): hashgrid.HashGrid.render_fore_rays + a torch
    loss + loss.backward() on the FUSED op (render.FusedRenderRays) and on the OP-BY-OP route (row-mapped encoder op, decoder
    op of csrc/decoder.hip on 8.4e6 samples, torch compositing, level-synchronous binned scatter with 12-byte records): same
    loss, table and decoder-parameter gradients within 1e-4 relative L2 of each other; the op-by-op table gradient obeys the
    trilinear scatter's conservation law per level (sum over a level's entries = sum of the incoming feature gradients)."""
    import scanerf_amd  # noqa: F401
    from scanerf_amd import network
    from scanerf_amd.hashgrid import HashGrid
    torch.manual_seed(23)
    o = torch.rand(B, 3, device=DEV) * 8 - 4
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1) * (0.5 + torch.rand(B, 1, device=DEV))
    tgt = torch.rand(B, 3, device=DEV)
    res = {}
    for fused in (True, False):
        torch.manual_seed(5)
        hg = HashGrid(DEV, torch.tensor([-4.0, -4, -4]), torch.tensor([8.0, 8, 8]), log2_hashmap_size=LOG2_T, grid_resolution=[32, 2048],
                      sampler_log2dim=4)
        with torch.no_grad():
            hg.HE.features.mul_(300.0)
        dec = network.init_model(network.ShallowMLP(32), "xavier").to(DEV)
        hg.fused = fused
        out, ok = hg.render_fore_rays(o, d, S_, dec, 0, global_step=20000)
        assert ok and hg.last_render_route == ("fused" if fused else "ops") and bool(out["fore_valid"].all())
        loss = torch.nn.functional.mse_loss(out["pred_color"], tgt) + 0.01 * out["l2_reg_specular"] + 1e-3 * (out["pred_depth"] ** 2).mean()
        loss.backward()
        torch.cuda.synchronize()
        res[fused] = (float(loss), hg.HE.features.grad.clone(), {n: p.grad.clone() for n, p in dec.named_parameters()},
                      out["pred_color"].detach().clone())
        del hg, dec, out, loss
        torch.cuda.empty_cache()
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=2e-5)
    np.testing.assert_allclose(res[True][3].cpu().numpy(), res[False][3].cpu().numpy(), rtol=1e-4, atol=2e-6)
    e_tab = rel(res[False][1], res[True][1])
    e_dec = max(rel(res[False][2][n], res[True][2][n]) for n in res[True][2])
    print(f"configs[1] routes: loss {res[True][0]:.6f}; op-by-op vs fused: table gradient {e_tab:.2e}, worst decoder parameter {e_dec:.2e} (relative L2)")
    assert e_tab < 1e-4 and e_dec < 1e-4
    assert float((res[False][1] != 0).float().mean()) > 0.3
    lv = res[False][1].double().sum(dim=(1, 2)), res[True][1].double().sum(dim=(1, 2))     # per-level conservation, both routes
    np.testing.assert_allclose(lv[0].cpu().numpy(), lv[1].cpu().numpy(), rtol=1e-3, atol=1e-9 * float(res[True][1].abs().sum()))
