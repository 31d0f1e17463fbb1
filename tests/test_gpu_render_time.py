"""GPU parity for the render-time ops: the multi-tile novel-view loop of
rendering.py:286-544 driven through the reference's HASHGRID binding names, every stage compared
against the oracle's restatement of hashgrid/src/rendering_kernel.cu.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def g(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV).contiguous()


def _scene(rng, T=2 ** 10):
    corners = np.float32([[-4, -2, -2], [-1, -2, -2], [2, -2, -2]])   # x-overlaps: [-1,0] and [2,3]
    sizes = np.float32([[4, 4, 4], [4, 4, 4], [4, 4, 4]])
    l2d = np.int32([[3, 3, 3], [4, 3, 3], [3, 3, 3]])
    grids = [rng.random(tuple(2 ** k for k in l)) < 0.35 for l in l2d]
    starts = np.cumsum([0] + [gr.size for gr in grids[:-1]]).astype(np.int64)
    occ = np.concatenate([gr.reshape(-1) for gr in grids])
    tables = (rng.normal(size=(3, 16, T, 2)) * 0.6).astype(np.float16)
    params = []
    for b in range(3):
        sd = O.init_mlp(seed=20 + b, bias_scale=0.05)
        sd["sigma_layer.mlp.0.bias"] = sd["sigma_layer.mlp.0.bias"] + 4.0  # visible densities (alpha ~ 0.1 per sample)
        params.append(O.pack_blob(sd).numpy())
    params = np.stack(params)
    res1 = O.level_resolutions(torch.tensor([32, 32, 32]), torch.tensor([2048, 2048, 2048])).numpy()
    res = np.stack([res1, res1, res1]).astype(np.int32)
    return dict(corners=corners, sizes=sizes, l2d=l2d, starts=starts, occ=occ, tables=tables, params=params, res=res, T=T)


def _rays(rng, B):
    o = np.stack([rng.uniform(-9, -5, B), rng.uniform(-1.5, 1.5, B), rng.uniform(-1.5, 1.5, B)], 1).astype(np.float32)
    d = np.stack([np.ones(B), rng.normal(0, 0.12, B), rng.normal(0, 0.12, B)], 1).astype(np.float32)
    d *= rng.uniform(0.7, 1.3, (B, 1)).astype(np.float32)
    o[: B // 8] = [0.5, 0.2, -0.3]       # some cameras inside tile 1
    o[B // 8: B // 6] = [-20, 30, 0]     # some rays miss everything
    return o, d.astype(np.float32)


def oracle_render_loop(rnd, o, d, num_sample, num_bg_sample):
    """rendering.py:286-544's loop on the oracle's restatement of the render-time kernels, over the renderer's own scene arrays
    (numpy rays o, d [B,3]) -> {"dif", "spec", "depth", "T"}: what TileSetRenderer.render_rays must reproduce."""
    B = o.shape[0]
    corners, sizes = rnd.block_corner.cpu().numpy(), rnd.block_size.cpu().numpy()
    occ, fake = rnd.occupied_grid.cpu().numpy(), rnd.fake_occupied_grid.cpu().numpy()
    starts, l2d = rnd.grid_starts.cpu().numpy(), rnd.grid_log2dim.cpu().numpy()
    tabs, par, res = rnd.feature_tables.cpu().numpy(), rnd.params.cpu().numpy(), rnd.resolution.cpu().numpy()
    inter = O.ray_block_intersection(o, d, corners, sizes)
    tb = np.argsort(inter[..., 0], axis=-1, kind="stable").astype(np.int32)
    max_tracing = int((inter != 1e7).astype(np.float32).mean(-1).sum(-1).max())
    T_, dF, sF, zF = np.ones((B, 1), np.float32), np.zeros((B, 3), np.float32), np.zeros((B, 3), np.float32), np.zeros((B, 1), np.float32)
    ti, zs = np.zeros(B, np.int32), np.zeros(B, np.float32)
    for _ in range(max_tracing):
        running = (ti < max_tracing) & (T_[:, 0] > 1e-5)
        if running.sum() == 0:
            break
        z, dd = O.render_sample_points(o, d, corners, sizes, fake, starts, l2d, num_sample, tb, inter, ti, zs)
        bi = O.prepare_points(z, running, inter)
        pd, ps, pa = O.pts_inference(o, d, z, dd, bi, tabs, par, res, occ, starts, l2d, corners, sizes)
        O.accumulate_color(pd, ps, pa, T_, z, dF, sF, zF)
    ob, bw = O.update_outgoing_bidx(o, d, corners, sizes, tb, inter, 0.12, False)
    with np.errstate(invalid="ignore", divide="ignore"):
        bwn = np.nan_to_num(bw / bw.sum(-1, keepdims=True))
    bd, bs, bz = np.zeros((B, 3), np.float32), np.zeros((B, 3), np.float32), np.zeros((B, 1), np.float32)
    for i in range(int((bwn > 0).sum(-1).max())):
        zb = O.render_inverse_z_sampling(inter, ob[:, i], num_bg_sample, 1e6)
        pd, ps, pa = O.bg_pts_inference_v2(o, d, zb, ob, i, corners, sizes, res, tabs, par)
        t1, td, ts, tz = np.ones((B, 1), np.float32), np.zeros((B, 3), np.float32), np.zeros((B, 3), np.float32), np.zeros((B, 1), np.float32)
        O.accumulate_color(pd, ps, pa, t1, zb, td, ts, tz)
        bd += td * bwn[:, i:i + 1]; bs += ts * bwn[:, i:i + 1]; bz += tz * bwn[:, i:i + 1]
    return {"dif": dF + T_ * bd, "spec": sF + T_ * bs, "depth": zF + T_ * bz, "T": T_}


def test_render_loop_stage_by_stage():
    import scanerf_amd  # noqa
    from scanerf_amd import hashgrid as H
    rng = np.random.default_rng(21)
    sc = _scene(rng)
    B, S = 700, 64
    o, d = _rays(rng, B)
    nb = 3
    C, Z, OCC, ST, L2 = g(sc["corners"]), g(sc["sizes"]), g(sc["occ"]), g(sc["starts"]), g(sc["l2d"])
    RO, RD = g(o), g(d)
    TAB, PAR, RES = g(sc["tables"]), g(sc["params"]), g(sc["res"])

    # ---- ray_block_intersection (bit-exact) + tracing order
    inter = torch.full((B, nb, 2), 1e7, device=DEV)
    H.ray_block_intersection(RO, RD, C, Z, inter)
    inter_ref = O.ray_block_intersection(o, d, sc["corners"], sc["sizes"])
    assert np.array_equal(inter.cpu().numpy(), inter_ref)
    tb_ref = np.argsort(inter_ref[..., 0], axis=-1, kind="stable").astype(np.int32)
    TB = g(tb_ref)
    max_tracing = int((inter_ref != 1e7).astype(np.float32).mean(-1).sum(-1).max())
    assert max_tracing == 3

    # ---- helpers with no caller in rendering.py but on the binding surface
    last = torch.full((B,), -1, dtype=torch.int32, device=DEV)
    H.get_last_block(TB, last, inter)
    assert np.array_equal(last.cpu().numpy(), O.get_last_block(tb_ref, inter_ref))

    tracing_idx, z_start = np.zeros(B, np.int32), np.zeros(B, np.float32)
    TI, ZS = g(tracing_idx), g(z_start)
    transp, dif, spec, depth = (np.ones((B, 1), np.float32), np.zeros((B, 3), np.float32), np.zeros((B, 3), np.float32),
                                np.zeros((B, 1), np.float32))
    TR, DI, SP, DE = g(transp), g(dif), g(spec), g(depth)
    n_overlap = 0
    for step in range(max_tracing):
        running_ref = (tracing_idx < max_tracing) & (transp[:, 0] > 1e-5)
        # ---- sample_points (bit-exact, incl. the in/out state)
        z = torch.full((B, S), -1.0, device=DEV)
        dd = torch.full((B, S), -1.0, device=DEV)
        H.sample_points(RO, RD, C, Z, OCC, ST, L2, TB, inter, TI, ZS, z, dd)
        z_ref, d_ref = O.render_sample_points(o, d, sc["corners"], sc["sizes"], sc["occ"], sc["starts"], sc["l2d"], S, tb_ref,
                                              inter_ref, tracing_idx, z_start)
        assert np.array_equal(z.cpu().numpy(), z_ref) and np.array_equal(dd.cpu().numpy(), d_ref), f"step {step}"
        assert np.array_equal(TI.cpu().numpy(), tracing_idx) and np.array_equal(ZS.cpu().numpy(), z_start)
        # ---- prepare_points (bit-exact)
        bi = torch.full((B, S, 4), -1, dtype=torch.int16, device=DEV)
        H.prepare_points(z, g(running_ref), inter, bi)
        bi_ref = O.prepare_points(z_ref, running_ref, inter_ref)
        assert np.array_equal(bi.cpu().numpy(), bi_ref)
        n_overlap += int((bi_ref[..., 1] != -1).sum())
        # ---- pts_inference (MFMA decoder vs the scalar restatement of decoder.h)
        pd, ps, pa = (torch.zeros(B, S, 3, device=DEV), torch.zeros(B, S, 3, device=DEV), torch.zeros(B, S, 1, device=DEV))
        H.pts_inference(RO, RD, z, dd, bi, TAB, PAR, RES, OCC, ST, L2, C, Z, pd, ps, pa)
        rd_, rs_, ra_ = O.pts_inference(o, d, z_ref, d_ref, bi_ref, sc["tables"], sc["params"], sc["res"], sc["occ"],
                                        sc["starts"], sc["l2d"], sc["corners"], sc["sizes"])
        np.testing.assert_allclose(pa.cpu().numpy(), ra_, rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(pd.cpu().numpy(), rd_, rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(ps.cpu().numpy(), rs_, rtol=1e-4, atol=2e-6)
        assert ra_.max() > 0.05 and transp.min() < 0.5 if step == max_tracing - 1 else True
        # ---- accumulate_color
        H.accumulate_color(pd, ps, pa, TR, z, DI, SP, DE)
        O.accumulate_color(rd_, rs_, ra_, transp, z_ref, dif, spec, depth)
        np.testing.assert_allclose(TR.cpu().numpy(), transp, rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(DI.cpu().numpy(), dif, rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(SP.cpu().numpy(), spec, rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(DE.cpu().numpy(), depth, rtol=1e-4, atol=2e-5)
    assert n_overlap > 100, "the scene must exercise multi-tile blending"

    # ---- background: exit tiles + blend weights (bit-exact), inverse-z, bg inference, accumulation
    ob = torch.full((B, 4), -1, dtype=torch.int16, device=DEV)
    bw = torch.zeros((B, 4), device=DEV)
    H.update_outgoing_bidx(RO, RD, C, Z, TB, inter, ob, bw, 0.12, False)
    ob_ref, bw_ref = O.update_outgoing_bidx(o, d, sc["corners"], sc["sizes"], tb_ref, inter_ref, 0.12, False)
    assert np.array_equal(ob.cpu().numpy(), ob_ref) and np.array_equal(bw.cpu().numpy(), bw_ref)
    ob2 = torch.full((B, 4), -1, dtype=torch.int16, device=DEV)
    bw2 = torch.zeros((B, 4), device=DEV)
    H.update_outgoing_bidx_v2(RO, RD, C, Z, TB, inter, ob2, bw2)
    ob2_ref, bw2_ref = O.update_outgoing_bidx_v2(o, sc["corners"], sc["sizes"])
    assert np.array_equal(ob2.cpu().numpy(), ob2_ref) and np.array_equal(bw2.cpu().numpy(), bw2_ref)
    Sb = 48
    for i in range(2):
        zb = torch.full((B, Sb), -1.0, device=DEV)
        H.inverse_z_sampling(inter, ob[:, i].contiguous(), zb, 1e6)
        zb_ref = O.render_inverse_z_sampling(inter_ref, ob_ref[:, i], Sb, 1e6)
        assert np.array_equal(zb.cpu().numpy(), zb_ref)
        pd, ps, pa = (torch.zeros(B, Sb, 3, device=DEV), torch.zeros(B, Sb, 3, device=DEV), torch.zeros(B, Sb, 1, device=DEV))
        H.bg_pts_inference_v2(RO, RD, zb, ob, i, C, Z, RES, TAB, PAR, pd, ps, pa)
        rd_, rs_, ra_ = O.bg_pts_inference_v2(o, d, zb_ref, ob_ref, i, sc["corners"], sc["sizes"], sc["res"], sc["tables"],
                                              sc["params"])
        np.testing.assert_allclose(pa.cpu().numpy(), ra_, rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(pd.cpu().numpy(), rd_, rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(ps.cpu().numpy(), rs_, rtol=1e-4, atol=2e-6)
        if i == 0:
            assert (ob_ref[:, 0] != -1).sum() > B // 2 and ra_.max() > 0.05
    # ---- bg_pts_inference (v1: hashgrid/binding.cpp:31, rendering_kernel.cu:872-1008 -- every outgoing tile of the ray blended per
    # sample on ONE set of z_vals; no caller in rendering.py, on the binding surface).  Slots behind a -1 must be ignored.
    zb = torch.full((B, Sb), -1.0, device=DEV)
    H.inverse_z_sampling(inter, ob[:, 0].contiguous(), zb, 1e6)
    zb_ref = O.render_inverse_z_sampling(inter_ref, ob_ref[:, 0], Sb, 1e6)
    ob_v1, bw_v1 = ob_ref.copy(), np.abs(bw_ref) + 0.1
    ob_v1[::7, 0] = -1            # rays whose FIRST slot is empty: the loop breaks at once -> zeros, whatever follows
    ob_v1[1::5, 1] = ob_v1[1::5, 0]   # two tiles on many rays
    ob_v1[2::9, 2] = 0            # ... and a third behind a second that may be -1 (then ignored)
    bw_v1[3::11] = 0.0            # zero total weight: sums stay unnormalised (= 0)
    pd, ps, pa = (torch.full((B, Sb, 3), 5.0, device=DEV), torch.full((B, Sb, 3), 5.0, device=DEV), torch.full((B, Sb, 1), 5.0, device=DEV))
    H.bg_pts_inference(RO, RD, zb, g(ob_v1), g(bw_v1.astype(np.float32)), C, Z, RES, TAB, PAR, pd, ps, pa)
    rd_, rs_, ra_ = O.bg_pts_inference(o, d, zb_ref, ob_v1, bw_v1.astype(np.float32), sc["corners"], sc["sizes"], sc["res"], sc["tables"],
                                       sc["params"])
    assert ra_.max() > 0.05 and (ra_[::7] == 0).all()
    np.testing.assert_allclose(pa.cpu().numpy(), ra_, rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(pd.cpu().numpy(), rd_, rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(ps.cpu().numpy(), rs_, rtol=1e-4, atol=2e-6)


def _inference_case(B=1000, S=72, seed=5, empty_tile=None):
    """fg (overlapping tiles, occupancy skips, ragged end) and bg inputs of the two inference ops -> run() = their six outputs"""
    import scanerf_amd  # noqa
    from scanerf_amd import hashgrid as H
    rng = np.random.default_rng(seed)
    sc = _scene(rng)
    nb = 3
    if empty_tile is not None:   # a tile whose density head answers -300 everywhere: softplus -> 0, opacity exactly 0
        sd = O.init_mlp(seed=20 + empty_tile, bias_scale=0.05)
        sd["sigma_layer.mlp.0.weight"] = sd["sigma_layer.mlp.0.weight"] * 0.0
        sd["sigma_layer.mlp.0.bias"] = sd["sigma_layer.mlp.0.bias"] * 0.0 - 300.0
        sc["params"][empty_tile] = O.pack_blob(sd).numpy()
    o, d = _rays(rng, B)
    C, Z, OCC, ST, L2 = g(sc["corners"]), g(sc["sizes"]), g(sc["occ"]), g(sc["starts"]), g(sc["l2d"])
    RO, RD, TAB, PAR, RES = g(o), g(d), g(sc["tables"]), g(sc["params"]), g(sc["res"])
    inter = torch.full((B, nb, 2), 1e7, device=DEV)
    H.ray_block_intersection(RO, RD, C, Z, inter)
    TB = torch.argsort(inter[..., 0], dim=-1, stable=True).int().contiguous()
    TI, ZS = torch.zeros(B, dtype=torch.int32, device=DEV), torch.zeros(B, device=DEV)
    z, dd = torch.full((B, S), -1.0, device=DEV), torch.full((B, S), -1.0, device=DEV)
    H.sample_points(RO, RD, C, Z, OCC, ST, L2, TB, inter, TI, ZS, z, dd)
    bi = torch.full((B, S, 4), -1, dtype=torch.int16, device=DEV)
    H.prepare_points(z, torch.ones(B, dtype=torch.bool, device=DEV), inter, bi)
    ob = torch.full((B, 4), -1, dtype=torch.int16, device=DEV)
    bw = torch.zeros((B, 4), device=DEV)
    H.update_outgoing_bidx(RO, RD, C, Z, TB, inter, ob, bw, 0.12, False)
    zb = torch.full((B, S), -1.0, device=DEV)
    H.inverse_z_sampling(inter, ob[:, 0].contiguous(), zb, 1e6)

    def run():
        # (both ops write EVERY sample -- zeros where no tile applies, rendering_kernel.cu:569-571 -- whatever the arrays held)
        fg = [torch.full((B, S, 3), 7.0, device=DEV), torch.full((B, S, 3), 7.0, device=DEV), torch.full((B, S, 1), 7.0, device=DEV)]
        H.pts_inference(RO, RD, z, dd, bi, TAB, PAR, RES, OCC, ST, L2, C, Z, *fg)
        bg = [torch.full((B, S, 3), 7.0, device=DEV), torch.full((B, S, 3), 7.0, device=DEV), torch.full((B, S, 1), 7.0, device=DEV)]
        H.bg_pts_inference_v2(RO, RD, zb, ob, 0, C, Z, RES, TAB, PAR, *bg)
        return [t.cpu().numpy() for t in fg + bg]

    run.ctx = dict(H=H, RO=RO, RD=RD, z=z, dd=dd, bi=bi, inter=inter, TAB=TAB, PAR=PAR, RES=RES, OCC=OCC, ST=ST, L2=L2, C=C, Z=Z, B=B, S=S)
    return run


def test_pipelined_group_loop_gives_the_bits_of_the_plain_one(monkeypatch):
    """The 32-sample-tile chunk kernel's software pipeline (next group's corner loads in flight during a group's decoder) reorders
    memory traffic only: fg and bg outputs equal SCANERF_RENDER_PIPE=0 bit for bit."""
    from conftest import need_experiments
    need_experiments("the 32-sample-tile kernel without its software pipeline")
    from scanerf_amd.hashgrid.lib import HASHGRID as _HG
    run = _inference_case()
    monkeypatch.setattr(_HG, "INFER_ARITH", "h3")
    monkeypatch.delenv("SCANERF_RENDER_PIPE", raising=False)
    piped = run()
    monkeypatch.setenv("SCANERF_RENDER_PIPE", "0")
    plain = run()
    assert piped[2].max() > 0.05 and piped[5].max() > 0.05
    for a, b in zip(piped, plain):
        assert np.array_equal(a, b)
    # and launch after launch (no atomics on this path: any difference is a hazard the generated code does not cover);
    # other work in between, so that the tables are not always warm
    monkeypatch.delenv("SCANERF_RENDER_PIPE", raising=False)
    scratch = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    for rep in range(12):
        scratch.fill_(rep)
        again = run()
        for a, b in zip(piped, again):
            assert np.array_equal(a, b), f"launch {rep} differs"


@pytest.mark.parametrize("B,S", [(1000, 72), (3000, 16)])
def test_inference_that_derives_its_slot_lists_equals_prepare_points_then_pts_inference(B, S):
    """pts_inference_tracing (one launch, no block_idxs array) against prepare_points + pts_inference: bit for bit, with a third
    of the rays stopped; and every sample written whatever the arrays held."""
    c = _inference_case(B, S, seed=13).ctx
    H = c["H"]
    running = (torch.arange(B, device=DEV) % 3 != 1).contiguous()
    bi = torch.full((B, S, 4), -1, dtype=torch.int16, device=DEV)
    H.prepare_points(c["z"], running, c["inter"], bi)
    tail = (c["TAB"], c["PAR"], c["RES"], c["OCC"], c["ST"], c["L2"], c["C"], c["Z"])
    two = [torch.full((B, S, 3), 7.0, device=DEV), torch.full((B, S, 3), 7.0, device=DEV), torch.full((B, S, 1), 7.0, device=DEV)]
    H.pts_inference(c["RO"], c["RD"], c["z"], c["dd"], bi, *tail, *two)
    one = [torch.full((B, S, 3), -3.0, device=DEV), torch.full((B, S, 3), -3.0, device=DEV), torch.full((B, S, 1), -3.0, device=DEV)]
    H.pts_inference_tracing(c["RO"], c["RD"], c["z"], c["dd"], running, c["inter"], *tail, *one)
    assert float(two[2].max()) > 0.05 and float((two[2][running] > 0).float().mean()) > 0.01
    assert float(two[2][~running].abs().max()) == 0.0
    for a, b in zip(one, two):
        assert torch.equal(a, b)
    # SKIP_UNSAMPLED: rays whose first depth is -1 (they got no sample) are left unwritten by the inference and unread by the
    # accumulation; the per-ray results are those of the plain pair
    z = c["z"]
    unsampled = z[:, 0] == -1.0
    assert 0 < int(unsampled.sum()) < B and bool((z[unsampled] == -1.0).all())
    skp = [torch.full((B, S, 3), -3.0, device=DEV), torch.full((B, S, 3), -3.0, device=DEV), torch.full((B, S, 1), -3.0, device=DEV)]
    H.pts_inference_tracing(c["RO"], c["RD"], z, c["dd"], running, c["inter"], *tail, *skp, sample_major=H.SKIP_UNSAMPLED)
    for a, b in zip(skp, one):
        assert torch.equal(a[~unsampled], b[~unsampled])
        assert bool((a[unsampled] == -3.0).all())

    def acc(arrs, flag):
        tr = torch.linspace(0.2, 1.0, B, device=DEV).reshape(B, 1).contiguous()
        out = [torch.zeros(B, 3, device=DEV), torch.zeros(B, 3, device=DEV), torch.zeros(B, 1, device=DEV)]
        H.accumulate_color(*arrs, tr, z, *out, sample_major=flag)
        return [tr] + out
    for a, b in zip(acc(skp, H.SKIP_UNSAMPLED), acc(one, 0)):
        assert torch.equal(a, b)


def test_tiles_of_zero_opacity_skip_the_directional_layers_and_change_nothing(monkeypatch):
    """Samples whose opacity 1 - exp(-sigma delta) is exactly zero (tile 1's density head answers -300) leave zeros; the
    16-sample-tile kernel skips the directional layers of tiles that hold only such samples, the 32-sample-tile kernel does not:
    equal outputs, exact zeros where tile 1 is the only tile, the other tiles untouched."""
    from scanerf_amd.hashgrid.lib import HASHGRID as _HG
    run = _inference_case(1000, 72, seed=17, empty_tile=1)
    t16 = run()
    monkeypatch.setattr(_HG, "INFER_ARITH", "h3")
    h3 = run()
    monkeypatch.setattr(_HG, "INFER_ARITH", "t16")
    bi = run.ctx["bi"].cpu().numpy()
    only1 = (bi[..., 0] == 1) & (bi[..., 1] == -1)
    assert only1.sum() > 1000 and t16[2].max() > 0.05
    for a, b in zip(t16[:3], h3[:3]):
        assert np.all(a[only1] == 0.0) and np.all(b[only1] == 0.0)
    for a, b in zip(t16, h3):
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("B,S", [(1000, 72), (517, 128), (3000, 16)])
def test_sixteen_sample_tile_kernel_against_the_32_sample_one(monkeypatch, B, S):
    """k_pts_inference_t16 (default: 16-sample tiles, four waves per SIMD, the chunk's SH operands in LDS) against
    k_pts_inference_chunks (SCANERF_RENDER_ARITH=h3): both evaluate the split-f16 decoder, with different k-step groupings, so
    they agree to f32 rounding; the SH rows change where the harmonics are evaluated, not their bits (SCANERF_RENDER_SH_ROWS=0:
    equal bit for bit; S = 16 makes a chunk's ray range too wide for the rows, so that case runs without them anyway); and the
    default repeats launch after launch."""
    from conftest import experiments_build
    from scanerf_amd.hashgrid.lib import HASHGRID as _HG
    run = _inference_case(B, S, seed=11)
    monkeypatch.delenv("SCANERF_RENDER_SH_ROWS", raising=False)
    t16 = run()
    norows = t16
    if experiments_build():   # (the switch exists in a `make EXP=1` library only; S = 16 runs without the rows in every build)
        monkeypatch.setenv("SCANERF_RENDER_SH_ROWS", "0")
        norows = run()
        monkeypatch.delenv("SCANERF_RENDER_SH_ROWS", raising=False)
    monkeypatch.setattr(_HG, "INFER_ARITH", "h3")
    h3 = run()
    monkeypatch.setattr(_HG, "INFER_ARITH", "t16")
    assert t16[2].max() > 0.05 and t16[5].max() > 0.05
    for a, b, c in zip(t16, norows, h3):
        assert np.array_equal(a, b)
        np.testing.assert_allclose(a, c, rtol=2e-5, atol=2e-6)
    scratch = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    for rep in range(6):
        scratch.fill_(rep)
        for a, b in zip(t16, run()):
            assert np.array_equal(a, b), f"launch {rep} differs"


def test_ray_firsthit_block():
    """rendering_kernel.cu:705-813 through the HASHGRID binding name, bit-exact against the oracle's restatement: the tile
    with the nearest far bound among those whose occupancy the ray touches, else the last tile crossed, else -1."""
    import scanerf_amd  # noqa
    from scanerf_amd import hashgrid as H
    rng = np.random.default_rng(23)
    sc = _scene(rng)
    # sparse grids so that many rays cross a tile without touching an occupied cell (the "last tile" rule), tile 2 empty
    grids = [rng.random(tuple(2 ** k for k in l)) < p for l, p in zip(sc["l2d"], (0.02, 0.01, 0.0))]
    sc["occ"] = np.concatenate([gr.reshape(-1) for gr in grids])
    B, nb = 3000, 3
    o, d = _rays(rng, B)
    d[::5, 0] *= -1.0                     # rays that leave the scene backwards
    d[1::7, 1:] = 0.0                     # axis-aligned rays (safe_divide branch of the DDA)
    C, Z, OCC, ST, L2 = g(sc["corners"]), g(sc["sizes"]), g(sc["occ"]), g(sc["starts"]), g(sc["l2d"])
    RO, RD = g(o), g(d)
    inter = torch.full((B, nb, 2), 1e7, device=DEV)
    H.ray_block_intersection(RO, RD, C, Z, inter)
    TB = torch.argsort(inter[..., 0], dim=-1).int().contiguous()   # rendering.py:301 tracing order
    hit = torch.full((B,), -1, dtype=torch.int16, device=DEV)
    H.ray_firsthit_block(RO, RD, C, Z, OCC, ST, L2, TB, inter, hit)
    inter_ref = O.ray_block_intersection(o, d, sc["corners"], sc["sizes"])
    assert np.array_equal(inter.cpu().numpy(), inter_ref)
    hit_ref = O.ray_firsthit_block(o, d, sc["corners"], sc["sizes"], sc["occ"], sc["starts"], sc["l2d"], TB.cpu().numpy(), inter_ref)
    got = hit.cpu().numpy()
    assert np.array_equal(got, hit_ref)
    # the three outcomes all occur: a touched tile, the last tile crossed without touching anything, no tile at all
    crossed = (inter_ref[..., 0] != 1e7).any(1)
    assert (got == -1).sum() > 0 and ((got != -1) & crossed).sum() > B // 2
    assert np.array_equal(got == -1, ~crossed)
    assert (got == 2).sum() > 0   # tile 2 is empty: it can only be chosen by the last-tile rule


def test_process_occupied_grid_and_sort_by_key():
    import scanerf_amd  # noqa
    from scanerf_amd import hashgrid as H
    rng = np.random.default_rng(22)
    sc = _scene(rng)
    C, Z, OCC, ST, L2 = g(sc["corners"]), g(sc["sizes"]), g(sc["occ"]), g(sc["starts"]), g(sc["l2d"])
    tgt = OCC.clone()
    tgt_ref = sc["occ"].astype(np.uint8).copy()
    for b in range(3):
        total = int(np.prod(2 ** sc["l2d"][b]))
        H.process_occupied_grid(b, total, C, Z, OCC, ST, L2, tgt)
        O.process_occupied_grid(b, total, sc["corners"], sc["sizes"], sc["occ"], sc["starts"], sc["l2d"], tgt_ref)
    assert np.array_equal(tgt.cpu().numpy().astype(np.uint8), tgt_ref)
    assert tgt_ref.sum() > sc["occ"].sum()
    # sort_by_key: thrust sort_by_key + unique_by_key semantics (rendering_kernel.cu:452-463)
    keys = torch.tensor([3, 1, 3, 2, 1, 1], dtype=torch.int16, device=DEV)
    vals = torch.arange(6, dtype=torch.int32, device=DEV)
    starts = torch.arange(6, dtype=torch.int32, device=DEV)
    n = H.sort_by_key(keys, vals, starts)
    assert n == 3 and keys[:3].tolist() == [1, 2, 3] and starts[:3].tolist() == [0, 3, 4]


def test_renderer_end_to_end_and_tile_formats(tmp_path):
    """f2/f3: export two trained-tile directories (feature.npz + decoder.pth), load them into the multi-tile
    renderer, render a small view and compare the image with the oracle driven through the same loop."""
    import scanerf_amd  # noqa
    from scanerf_amd import renderer as R
    from scanerf_amd.tile_model import TileModel
    rng = np.random.default_rng(23)
    tiles = []
    for b, cx in enumerate((-4.0, 2.0)):  # tiles [-4,4]x.. and [2,10]x..: 2 m overlap in x
        m = TileModel([cx, -4, -4], [8, 8, 8], DEV, log2_T=10, seed=30 + b, sampler_log2dim=4)
        with torch.no_grad():
            m.features.mul_(60.0)
            m.decoder.sigma_layer_mlp_0_bias.add_(3.0)
        m.occupied_grid = g(rng.random((16, 16, 16)) < 0.4)
        R.export_tile(str(tmp_path / f"tile{b}"), m)
        tiles.append(R.load_tile(str(tmp_path / f"tile{b}")))
    assert tiles[0]["features"].dtype == np.float16 and tiles[0]["blob"].shape == (13994,)
    rnd = R.TileSetRenderer(DEV, tiles)
    H, W = 20, 28
    K = np.float32([[30, 0, 14], [0, 30, 10], [0, 0, 1]])
    c2w = np.float32([[0, 0, 1, -9], [0, 1, 0, 0.3], [-1, 0, 0, 0.2]])  # looking down +x from x = -9
    dif, spec, depth, transp = rnd.render(H, W, K, c2w, num_sample=64, num_bg_sample=32)
    # (default: ray-block work arrays [B/32,S,32] between the ops; the reference's [B,S] layout and [S,B] give the same image --
    # the per-sample arithmetic is the same, the per-ray accumulation runs in sample order instead of as a wave scan)
    for lay in (0, 1):
        for a_, b_ in zip((dif, spec, depth, transp), rnd.render(H, W, K, c2w, num_sample=64, num_bg_sample=32, layout=lay)):
            np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), rtol=1e-5, atol=1e-6)
    ragged = rnd.render_rays(*(t[:333].contiguous() for t in rnd.compute_rays(H, W, K, c2w)), num_sample=64, num_bg_sample=32)
    for a_, b_ in zip((dif, spec, depth, transp), ragged):   # 333 rays: padded to a multiple of 32 inside
        assert torch.equal(a_.reshape(H * W, -1)[:333], b_)
    # opt-in skip of the backgrounds of saturated rays (not the reference's behaviour: rendering.py:534-536 adds
    # transparency * background for every ray): identical wherever the transmittance is above 1e-5, colour within 1e-5 and
    # depth within 1e-5 * sample_range elsewhere
    sk = rnd.render(H, W, K, c2w, num_sample=64, num_bg_sample=32, skip_saturated_background=True)
    sat = (transp <= 1e-5)[..., 0]
    for a_, b_, bound in zip((dif, spec, depth), sk[:3], (1e-5, 1e-5, 1e-5 * 1e6)):
        assert torch.equal(a_[~sat], b_[~sat])
        if bool(sat.any()):
            assert float((a_[sat] - b_[sat]).abs().max()) <= bound * 1.001
    # ---- the same loop on the oracle
    o, d = (t.cpu().numpy() for t in rnd.compute_rays(H, W, K, c2w))
    o_ref, d_ref = O.compute_ray_forward(np.stack([np.zeros(H * W), np.tile(np.arange(W), H), np.repeat(np.arange(H), W)], 1),
                                         K.reshape(1, 9), c2w.reshape(1, 12))
    assert np.array_equal(o, o_ref) and np.array_equal(d, d_ref)
    B = H * W
    ref = oracle_render_loop(rnd, o, d, 64, 32)
    assert ref["T"].min() < 0.5 < ref["T"].max(), "view must contain both opaque and empty pixels"
    np.testing.assert_allclose(transp.cpu().numpy().reshape(B, 1), ref["T"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(dif.cpu().numpy().reshape(B, 3), ref["dif"], rtol=1e-4, atol=3e-6)
    np.testing.assert_allclose(spec.cpu().numpy().reshape(B, 3), ref["spec"], rtol=1e-4, atol=3e-6)
    np.testing.assert_allclose(depth.cpu().numpy().reshape(B, 1), ref["depth"], rtol=1e-4, atol=1e-3)
    img_a = np.clip((dif + spec).cpu().numpy(), 0, 1) * 255
    img_b = np.clip(ref["dif"] + ref["spec"], 0, 1).reshape(H, W, 3) * 255
    assert O.psnr(img_a, img_b) > 80.0  # tools/utils.py:53-55 PSNR of the HIP render vs the oracle render


def test_packed_decoders_follow_their_owner_not_an_address():
    """Two renderers built one after the other over the same tile geometry but different decoders: the second must render
    with ITS decoders even when the caching allocator hands its blobs the address the first one's had (the packed MFMA
    images are owned by the renderer; the raw-tensor convenience cache is keyed on the tensor object, not its address)."""
    import gc
    import scanerf_amd  # noqa
    from scanerf_amd import hashgrid as H
    from scanerf_amd.hashgrid.lib import HASHGRID as HG
    rng = np.random.default_rng(24)
    sc = _scene(rng)
    B, S, nb = 64, 32, 3
    o, d = _rays(rng, B)
    RO, RD = g(o), g(d)
    C, Z, OCC, ST, L2, TAB, RES = (g(sc[k]) for k in ("corners", "sizes", "occ", "starts", "l2d", "tables", "res"))
    z = torch.linspace(2.0, 9.0, S, device=DEV).repeat(B, 1).contiguous()
    dd = torch.full((B, S), 0.25, device=DEV)
    bi = torch.full((B, S, 4), -1, dtype=torch.int16, device=DEV)
    bi[..., 0] = 0   # every sample in tile 0: the reversed parameter set below puts another decoder there

    def infer(params):
        pd, ps, pa = torch.zeros(B, S, 3, device=DEV), torch.zeros(B, S, 3, device=DEV), torch.zeros(B, S, 1, device=DEV)
        H.pts_inference(RO, RD, z, dd, bi, TAB, params, RES, OCC, ST, L2, C, Z, pd, ps, pa)
        torch.cuda.synchronize()
        return torch.cat([pd, ps, pa], -1).clone()

    p1 = g(sc["params"])
    addr = p1.data_ptr()
    out1 = infer(p1)
    del p1
    gc.collect()
    assert not HG._images, "the cache entry must die with its tensor"
    p2 = g(sc["params"][::-1].copy())       # same shape, different decoders; very likely the same address
    out2 = infer(p2)
    ref2 = infer(HG.PackedDecoders(p2))      # owner-held images
    assert torch.equal(out2, ref2)
    assert not torch.equal(out1, out2), f"second parameter set rendered with the first one's decoders (addr reuse: {p2.data_ptr() == addr})"
    # an in-place update through the version counter is seen; alternating two live parameter sets does not thrash
    p3 = g(sc["params"])
    out3 = infer(p3)
    assert torch.equal(out3, out1) and len(HG._images) == 2
    assert torch.equal(infer(p2), out2) and torch.equal(infer(p3), out3) and len(HG._images) == 2
    p3.mul_(0.5)
    assert not torch.equal(infer(p3), out3)


def test_sort_tracing_blocks_equals_stable_argsort():
    """The renderer's per-ray tile order (rendering.py:301) in one launch: identical to torch.argsort(near, stable=True),
    with misses (1e7), equal distances and nb from 1 to 64."""
    import scanerf_amd  # noqa
    from scanerf_amd import hashgrid as H
    gen = torch.Generator(device=DEV).manual_seed(3)
    for B, nb in ((1, 1), (1000, 2), (70001, 4), (513, 9), (300, 64)):
        inter = torch.rand(B, nb, 2, device=DEV, generator=gen) * 20
        inter[torch.rand(B, nb, device=DEV, generator=gen) < 0.4] = 1e7            # tiles the ray misses
        dup = torch.rand(B, device=DEV, generator=gen) < 0.3
        if nb > 1:
            inter[dup, nb - 1, 0] = inter[dup, 0, 0]                               # equal entry distances
        got = H.sort_tracing_blocks(inter.contiguous())
        want = torch.argsort(inter[..., 0], dim=-1, stable=True).int()
        assert got.dtype == torch.int32 and torch.equal(got, want), (B, nb)


def test_background_of_rays_with_exactly_zero_transmittance_is_skipped_without_changing_a_bit(tmp_path):
    """Round 6: a ray whose foreground transmittance is EXACTLY 0.0 (an opacity rounded to 1: the density head answers +40 here)
    gets no background samples -- `dif + 0 * bgd` is `dif`: colour, depth and transmittance are the bits of the unskipped render."""
    import scanerf_amd  # noqa
    from scanerf_amd import renderer as R
    from scanerf_amd.tile_model import TileModel
    rng = np.random.default_rng(5)
    m = TileModel([-4.0, -4, -4], [8, 8, 8], DEV, log2_T=10, seed=3, sampler_log2dim=4)
    with torch.no_grad():
        m.features.mul_(60.0)
        m.decoder.sigma_layer_mlp_0_bias.add_(40.0)
    m.occupied_grid = g(rng.random((16, 16, 16)) < 0.5)
    R.export_tile(str(tmp_path / "tile0"), m)
    rnd = R.TileSetRenderer(DEV, [R.load_tile(str(tmp_path / "tile0"))])
    H, W = 24, 32
    K = np.float32([[30, 0, 16], [0, 30, 12], [0, 0, 1]])
    c2w = np.float32([[0, 0, 1, -9], [0, 1, 0, 0.3], [-1, 0, 0, 0.2]])
    on = rnd.render(H, W, K, c2w, num_sample=64, num_bg_sample=32)
    rnd.skip_zero_transmittance_background = False
    off = rnd.render(H, W, K, c2w, num_sample=64, num_bg_sample=32)
    zero = int((on[3] == 0).sum())
    assert 50 < zero < H * W, zero      # some rays are exactly opaque, some see the background
    for a_, b_ in zip(on, off):
        assert torch.equal(a_, b_)
