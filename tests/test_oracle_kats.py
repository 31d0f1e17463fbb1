"""Known-answer tests for the C half of the oracle (restatement of the reference's .cu text;
the reference ships no tests, so these KATs are derived from the source: SURVEY.md 8c).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import oracle as O


@pytest.mark.parametrize("xyz,i19,i24", [
    ((0, 0, 0), 0, 0), ((1, 0, 0), 1, 1), ((0, 1, 0), 489905, 3635633), ((0, 0, 1), 153493, 153493),
    ((1, 1, 1), 339493, 3485221), ((31, 17, 5), 303415, 10789175), ((8191, 8191, 8191), 455131, 979419)])
def test_hash_kats(xyz, i19, i24):
    # primes 1, 0x9E3779B1, 0x30025795: hashgrid/src/hashgrid_bg_kernel.cu:17
    assert O.hash_index(*xyz, 2 ** 19) == i19
    assert O.hash_index(*xyz, 2 ** 24) == i24


def test_encoder_constant_table_gives_constant():
    rng = np.random.default_rng(0)
    pts = rng.uniform(-2, 2, (257, 3)).astype(np.float32)
    res = O.level_resolutions(torch.tensor([32, 32, 32]), torch.tensor([2048, 2048, 2048])).numpy()
    feat = np.zeros((16, 2 ** 10, 2), np.float32)
    feat[..., 0], feat[..., 1] = 0.75, -1.5
    out = O.embedding_forward(pts, feat, res)
    np.testing.assert_allclose(out[..., 0], 0.75, rtol=1e-6)
    np.testing.assert_allclose(out[..., 1], -1.5, rtol=1e-6)


def test_encoder_hits_grid_vertex_exactly():
    # p01*(res-1) integral => only corner 000 has weight 1
    res = np.array([[33, 33, 33]], np.int32)
    T = 2 ** 12
    feat = np.random.default_rng(1).normal(size=(1, T, 2)).astype(np.float32)
    ijk = np.array([[3, 7, 30], [0, 0, 0], [32, 32, 32]])
    pts = (ijk / 32.0 * 4.0 - 2.0).astype(np.float32)
    out = O.embedding_forward(pts, feat, res)
    for n, (i, j, k) in enumerate(ijk):
        np.testing.assert_array_equal(out[n, 0], feat[0, O.hash_index(i, j, k, T)])


def test_encoder_backward_is_adjoint_and_gradpoint_matches_fd():
    rng = np.random.default_rng(2)
    N, L, T = 50, 4, 2 ** 8
    res = np.array([[5, 6, 7], [9, 9, 9], [17, 13, 11], [40, 40, 40]], np.int32)
    pts = rng.uniform(-1.9, 1.9, (N, 3)).astype(np.float32)
    feat = rng.normal(size=(L, T, 2)).astype(np.float32)
    g = rng.normal(size=(N, L, 2)).astype(np.float32)
    gp, gf = O.embedding_backward(pts, g, feat, res)
    # <g, E(feat')> is linear in feat': adjoint identity
    f2 = rng.normal(size=feat.shape).astype(np.float32)
    lhs = np.sum(g.astype(np.float64) * O.embedding_forward(pts, f2, res))
    np.testing.assert_allclose(lhs, np.sum(gf.astype(np.float64) * f2), rtol=1e-4)
    # finite differences on points (float64 would be cleaner; eps sized for fp32)
    eps = 1e-3
    for ax in range(3):
        d = np.zeros((1, 3), np.float32)
        d[0, ax] = eps
        fd = np.sum(g * (O.embedding_forward(pts + d, feat, res) - O.embedding_forward(pts - d, feat, res)), (1, 2)) / (2 * eps)
        # a +-eps step that crosses a cell face sees the kink of the trilinear basis: allow a few outliers
        bad = np.abs(gp[:, ax] - fd) > 5e-2 + 5e-2 * np.abs(fd)
        assert bad.mean() <= 0.1, (ax, bad.sum())


def test_world_space_encoder_equals_bg_encoder_on_same_lattice():
    # box [-2,2]^3 => grid=4/(res-1): same cell/offset as the contracted variant up to rounding
    rng = np.random.default_rng(3)
    pts = rng.uniform(-1.99, 1.99, (200, 3)).astype(np.float32)
    res = np.array([[17, 17, 17], [65, 65, 65]], np.int32)  # 4/(res-1) exact in fp32
    feat = rng.normal(size=(2, 2 ** 9, 2)).astype(np.float32)
    a = O.embedding_forward(pts, feat, res)
    b = O.embedding_forward(pts, feat, res, corner=np.float32([-2, -2, -2]), size=np.float32([4, 4, 4]))
    np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-5)


def _tile():
    return O.Tile([-4, -4, -4], [8, 8, 8], sampler_log2dim=4)


def test_sampler_full_grid_axis_ray():
    t = _tile()
    S = 64
    o = np.float32([[-10, 0.3, 0.2]])
    d = np.float32([[1, 0, 0]])
    z, dist = O.sample_points_grid(o, d, t.occ_corner, t.occ_size, t.occ, t.log2dim, S)
    b = O.ray_aabb_intersection(o, d, t.bbox_center.numpy(), t.occ_size.numpy())
    np.testing.assert_allclose(b[0], [6.0, 14.0])
    assert np.all(z != -1)
    assert z[0, 0] == b[0, 0]
    np.testing.assert_allclose(dist.sum(), b[0, 1] - b[0, 0], rtol=1e-5)
    assert np.all(np.diff(z[0]) > 0)
    # 16 cells of 0.5 crossed, S/16 = 4 samples each
    np.testing.assert_allclose(dist[0], 0.125, rtol=1e-5)


def test_sampler_empty_grid_and_miss_keep_sentinel():
    t = _tile()
    o = np.float32([[-10, 0, 0], [-10, 50, 0]])
    d = np.float32([[1, 0, 0], [1, 0, 0]])
    z, dist = O.sample_points_grid(o, d, t.occ_corner, t.occ_size, torch.zeros_like(t.occ), t.log2dim, 16)
    assert np.all(z == -1) and np.all(dist == -1)
    z, dist = O.sample_points_grid(o, d, t.occ_corner, t.occ_size, t.occ, t.log2dim, 16)
    assert np.all(z[1] == -1) and np.all(z[0] != -1)


def test_sampler_sparse_grid_apportions_exactly_S_inside_occupied_cells():
    t = _tile()
    rng = np.random.default_rng(5)
    occ = torch.from_numpy(rng.random(tuple(t.occ.shape)) < 0.3)
    B, S = 300, 32
    o = rng.uniform(-4, 4, (B, 3)).astype(np.float32)
    d = rng.normal(size=(B, 3)).astype(np.float32)
    z, dist = O.sample_points_grid(o, d, t.occ_corner, t.occ_size, occ, t.log2dim, S)
    hit = z[:, 0] != -1
    assert hit.sum() > 100
    assert np.all((z[hit] != -1).all(1)) and np.all(dist[hit] > 0)
    assert np.all(np.diff(z[hit], axis=1) > 0)
    # every sample's left endpoint lies in an occupied cell (nudged inwards by half a step)
    p = o[hit, None, :] + (z[hit] + 0.5 * dist[hit])[..., None] * d[hit, None, :]
    cell = np.floor((p - t.occ_corner.numpy()) / (t.occ_size.numpy() / np.array(t.occ.shape))).astype(int)
    cell = np.clip(cell, 0, np.array(t.occ.shape) - 1)
    assert occ.numpy()[cell[..., 0], cell[..., 1], cell[..., 2]].mean() > 0.999


def test_aabb_kats():
    c, s = np.float32([0, 0, 0]), np.float32([2, 2, 2])
    o = np.float32([[-3, 0, 0], [0, 0, 0], [-3, 5, 0], [3, 0, 0], [0.5, 0.5, -4]])
    d = np.float32([[1, 0, 0], [0, 1, 0], [1, 0, 0], [1, 0, 0], [0, 0, 2]])
    b = O.ray_aabb_intersection(o, d, c, s)
    np.testing.assert_allclose(b, [[2, 4], [0, 1], [-1, -1], [-1, -1], [1.5, 2.5]])
    # v2 layout [B,K,2]
    b2 = O.ray_aabb_intersection(o, d, np.float32([[0, 0, 0], [10, 0, 0]]), np.float32([[2, 2, 2], [2, 2, 2]]))
    np.testing.assert_allclose(b2[:, 0], b)
    np.testing.assert_allclose(b2[0, 1], [12, 14])


def test_compositing_kat_sigma_zero():
    S = 16
    w, tl = O.cal_integrate_weight(torch.zeros(2, S, 1), torch.ones(2, S), torch.ones(2, 3), infinity=False)
    assert torch.all(w == 0)
    np.testing.assert_allclose(tl.numpy(), (1 + 1e-6) ** (S - 1), rtol=1e-6)


def test_sparse_adam_skips_zero_grad_and_uses_step_plus_one():
    rng = np.random.default_rng(7)
    p = rng.normal(size=(16, 8)).astype(np.float32)
    g = rng.normal(size=(16, 8)).astype(np.float32)
    g[::2] = 0
    m = rng.normal(size=(16, 8)).astype(np.float32) * 0.1
    v = np.abs(rng.normal(size=(16, 8))).astype(np.float32) * 0.1
    p0, m0, v0 = p.copy(), m.copy(), v.copy()
    O.adam_step(p, g, m, v, 1e-3, 0.9, 0.99, 1e-15, 4)
    assert np.array_equal(p[::2], p0[::2]) and np.array_equal(m[::2], m0[::2]) and np.array_equal(v[::2], v0[::2])
    t = 5.0
    mi = 0.9 * m0 + 0.1 * g
    vi = 0.99 * v0 + 0.01 * g * g
    ref = p0 - (1e-3 / (1 - 0.9 ** t)) * mi / (np.sqrt(vi / (1 - 0.99 ** t)) + 1e-15)
    np.testing.assert_allclose(p[1::2], ref[1::2], rtol=1e-5)


def test_half_conversion_matches_numpy():
    xs = np.float32([0, 1, -1, 65504, 1e-8, 6.1e-5, 3.14159, 1e6, -2.5e-7, 0.1])
    for x in xs:
        h = O.lib().orc_float2half(float(x))
        assert h == int(np.float16(x).view(np.uint16)), x
        assert np.float32(O.lib().orc_half2float(h)) == np.float32(np.float16(x))


def test_insideout_and_background_samplers():
    o = np.float32([[-3, 0, 0]])
    d = np.float32([[1, 0, 0]])
    z, zb, missed = O.sample_insideout_block(o, d, 5, 4, np.float32([0, 0, 0]), np.float32([2, 2, 2]), 100.0)
    assert missed == 0
    np.testing.assert_allclose(z[0], [2, 2.5, 3, 3.5, 4], rtol=1e-6)
    np.testing.assert_allclose(1 / zb[0], np.linspace(1 / 4, 1 / 100, 4), rtol=1e-5)
    zz = O.background_sampling(np.float32([1.0]), np.float32([5.0]), 5, 2.0)
    np.testing.assert_allclose(zz[0], [4, 4.5, 5, 5.5, 6], rtol=1e-6)


def test_compute_ray_backward_is_adjoint():
    rng = np.random.default_rng(11)
    C, B = 3, 40
    Ks = np.tile(np.float32([100, 0, 16, 0, 90, 12, 0, 0, 1]), (C, 1))
    locs = np.stack([rng.integers(0, C, B), rng.integers(0, 32, B), rng.integers(0, 24, B)], 1).astype(np.int32)
    M = rng.normal(size=(C, 12)).astype(np.float32)
    go, gd = rng.normal(size=(B, 3)).astype(np.float32), rng.normal(size=(B, 3)).astype(np.float32)
    g = O.compute_ray_backward(go, gd, Ks, locs, C)
    dM = rng.normal(size=(C, 12)).astype(np.float32)
    o1, d1 = O.compute_ray_forward(locs, Ks, M + dM)
    o0, d0 = O.compute_ray_forward(locs, Ks, M)
    lhs = np.sum(go * (o1 - o0)) + np.sum(gd * (d1 - d0))  # forward is linear in C2W
    np.testing.assert_allclose(lhs, np.sum(g * dM), rtol=1e-3)


def test_ray_firsthit_block_known_answers():
    """rendering_kernel.cu:705-813 from the source text: two 2 m tiles side by side on x, rays along +x through both.
    (tile 0 occupancy, tile 1 occupancy) -> expected hit: the touched tile with the smallest far bound; if none is touched
    the last tile crossed; -1 for a ray that meets no tile."""
    corners = np.float32([[0, 0, 0], [2, 0, 0]])
    sizes = np.float32([[2, 2, 2], [2, 2, 2]])
    l2d = np.int32([[1, 1, 1], [1, 1, 1]])
    starts = np.int64([0, 8])
    o = np.float32([[-1, 0.5, 0.5], [-1, 0.5, 0.5], [-1, 5.0, 0.5]])
    d = np.float32([[1, 0, 0], [-1, 0, 0], [1, 0, 0]])   # through both / away from both / beside both
    inter = O.ray_block_intersection(o, d, corners, sizes)
    assert inter[0].tolist() == [[1.0, 3.0], [3.0, 5.0]] and np.all(inter[1:] == 1e7)
    tb = np.argsort(inter[..., 0], axis=-1, kind="stable").astype(np.int32)
    full, empty = np.ones(8, np.uint8), np.zeros(8, np.uint8)
    for occ0, occ1, want in ((full, full, 0), (empty, full, 1), (full, empty, 0), (empty, empty, 1)):
        hit = O.ray_firsthit_block(o, d, corners, sizes, np.concatenate([occ0, occ1]), starts, l2d, tb, inter)
        assert hit.tolist() == [want, -1, -1], (occ0[0], occ1[0], hit)
    # only the cell the ray does NOT cross is occupied: the tile is crossed but not touched
    one = np.zeros(8, np.uint8)
    one[(1 << 2) | (1 << 1) | 1] = 1   # cell (1,1,1): y,z in [1,2), the ray runs at y = z = 0.5
    hit = O.ray_firsthit_block(o, d, corners, sizes, np.concatenate([one, full]), starts, l2d, tb, inter)
    assert hit.tolist() == [1, -1, -1]
