"""Golden vectors G13-G15 from the reference's importable Python half (round 5).

Run ONCE in the build container (where /root/reference exists):

    python tests/golden/make_golden_sampling.py

G13  PyHashGridBG.__init__'s per-axis level resolutions (hashgrid/PyHashGridBG.py:53-62) for cubic and NON-cubic boxes, built
     exactly as HashGrid.__init__ builds its arguments (hashgrid/__init__.py:35,56-57: base / finest = bbox_size /
     bbox_size.min() * grid_resolution, .int()).
G14  HashGrid.inverse_z_sampling (hashgrid/__init__.py:306-337) with invalid_underground in {False, True}; the CUDA-only
     ray_aabb_intersection it calls is replaced by this repo's C oracle of that op (in-place `bounds`, as the binding).
G15  HashGrid.render_fore_rays / render_bg_rays (hashgrid/__init__.py:413-509): the valid-mask logic (all(z != -1) & occlusion
     mask; bg valid from inverse_z_sampling & occlusion mask), the zero / one fill of invalid rays and the scatter of the
     rendered rows back -- with this repo's C oracle standing in for the three CUDA-only ops (sample_points_grid,
     ray_aabb_intersection, the hash encoder).  TRAIN and INFERENCE, with and without an occlusion mask.

Only DATA is written (inputs + the reference's outputs); no reference text is copied.  Nothing here runs on the GPU box.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, ROOT, _stub_modules  # noqa: E402

sys.dont_write_bytecode = True


def main():
    _stub_modules()
    sys.path.insert(0, REF)
    import network  # noqa
    import hashgrid as ref_hashgrid  # noqa
    import importlib
    ref_bg = importlib.import_module("hashgrid.PyHashGridBG")
    sys.path.insert(0, ROOT)
    from oracle import oracle

    def save(name, **kw):
        kw = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in kw.items()}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **kw)
        print("wrote", name, {k: v.shape for k, v in kw.items()})

    # ---- G13: level resolutions, per axis
    g13 = {}
    cases = [((8.0, 8.0, 8.0), (32, 2048)), ((8.0, 4.0, 16.0), (32, 2048)), ((10.0, 6.0, 7.5), (32, 8192)),
             ((3.0, 9.0, 4.5), (16, 512)), ((12.0, 12.0, 5.0), (32, 4096))]
    for i, (tile_size, (gb, gf)) in enumerate(cases):
        bbox_size = torch.tensor(tile_size) * 2                                       # hashgrid/__init__.py:50
        fin = (bbox_size / bbox_size.min() * gf).int()                                # :56
        base = (bbox_size / bbox_size.min() * gb).int()                               # :57
        he = ref_bg.PyHashGridBG("cpu", torch.zeros(3), bbox_size, 16, 2, 4, base, fin, "uniform")
        g13[f"tile_size{i}"] = np.array(tile_size, np.float32)
        g13[f"grid_resolution{i}"] = np.array([gb, gf])
        g13[f"resolution{i}"] = he.resolution
    g13["n"] = np.array(len(cases))
    save("g13_resolutions", **g13)

    # ---- a HashGrid without its native-dependent constructor (as make_golden.py), native ops -> this repo's C oracle
    hgm = ref_hashgrid.HashGrid.__new__(ref_hashgrid.HashGrid)
    torch.nn.Module.__init__(hgm)
    hgm.device = torch.device("cpu")
    corner, size = torch.tensor([-4.0, -3.0, -5.0]), torch.tensor([8.0, 8.0, 8.0])
    hgm.bbox_center = corner + size / 2.0
    hgm.bbox_size = size * 2
    hgm.min_bbox = hgm.bbox_center - hgm.bbox_size / 2.0

    def aabb(rays_o, rays_d, center, sz, bounds):  # cuda/binding.cpp ray_aabb_intersection: fills `bounds` in place
        bounds.copy_(torch.from_numpy(oracle.ray_aabb_intersection(rays_o.numpy(), rays_d.numpy(), center.numpy(), sz.numpy())))

    def spg(rays_o, rays_d, z_vals, dists, block_corner, block_size, occ, log2dim):  # sample_points_grid, in place
        z, d = oracle.sample_points_grid(rays_o.numpy(), rays_d.numpy(), block_corner.numpy(), block_size.numpy(), occ,
                                         log2dim, z_vals.shape[1])
        z_vals.copy_(torch.from_numpy(z))
        dists.copy_(torch.from_numpy(d))

    ref_hashgrid.ray_aabb_intersection = aabb
    ref_hashgrid.sample_points_grid = spg

    # ---- G14: inverse_z_sampling
    g = torch.Generator().manual_seed(14)
    B = 48
    ro = (torch.rand(B, 3, generator=g) - 0.5) * 8 + hgm.bbox_center          # inside the tile
    ro[40:] = hgm.bbox_center + torch.tensor([30.0, 2.0, 1.0])                 # outside the 2x box
    rd = torch.randn(B, 3, generator=g) * (0.5 + torch.rand(B, 1, generator=g))
    rd[40:44] = torch.tensor([1.0, 0.1, 0.0])                                  # pointing away: they miss the box -> far = 0.1
    rd[44:] = torch.tensor([-1.0, 0.02, 0.01])                                 # through the box from outside
    rd[:6, 1] = -torch.abs(rd[:6, 1]) * 4 - 1.0                                # steep downwards: leave through the floor
    g14 = {"rays_o": ro, "rays_d": rd, "bbox_center": hgm.bbox_center, "bbox_size": hgm.bbox_size, "S": np.array(24)}
    for ug in (False, True):
        z, d, v = hgm.inverse_z_sampling(ro, rd, 24, invalid_underground=ug)
        g14["z_ug%d" % ug], g14["dists_ug%d" % ug], g14["valid_ug%d" % ug] = z.contiguous(), d, v
    assert g14["valid_ug1"].sum() < B and g14["valid_ug1"].sum() > 0
    save("g14_inverse_z", **g14)

    # ---- G15: render_fore_rays / render_bg_rays
    torch.manual_seed(15)
    mlp = network.ShallowMLP(32)
    network.init_model(mlp, "xavier")
    T = 2 ** 10
    fin = (hgm.bbox_size / hgm.bbox_size.min() * 2048).int()
    base = (hgm.bbox_size / hgm.bbox_size.min() * 32).int()
    res = oracle.level_resolutions(base, fin, 16)
    feats = torch.randn(16, T, 2) * 0.5

    class HE(torch.nn.Module):
        def forward(self, x):
            return oracle.encode_bg(x.reshape(-1, 3).contiguous(), feats, res).reshape(*x.shape[:-1], 32)

    hgm.HE = HE()
    hgm.sampler_log2dim = torch.tensor([4, 4, 4], dtype=torch.int32)
    occ = torch.rand(16, 16, 16) < 0.25                                       # sparse: some rays meet no occupied cell
    occ[:, :, :4] = False
    hgm.occupied_grid = occ
    B, S = 40, 16
    ro = (torch.rand(B, 3) - 0.5) * 8 + hgm.bbox_center
    rd = torch.randn(B, 3) * (0.5 + torch.rand(B, 1))
    rd[:5, 1] = -torch.abs(rd[:5, 1]) * 4 - 1.0
    mask = (torch.rand(B, 1) < 0.7)
    g15 = {"rays_o": ro, "rays_d": rd, "features": feats, "res": res, "occ": occ, "log2dim": hgm.sampler_log2dim,
           "tile_corner": corner, "tile_size": size, "occlusion_mask": mask, "global_step": np.array(6000), "S": np.array(S)}
    sd = {k: v for k, v in mlp.state_dict().items()}
    g15.update({"sd." + k: v for k, v in sd.items()})
    for tag, m in (("nomask", None), ("mask", mask)):
        for mode in (0, 1):
            with torch.no_grad():
                fo, ok = hgm.render_fore_rays(ro, rd, S, mlp, mode, occlusion_mask=m, global_step=6000)
                assert ok
                bo, ok = hgm.render_bg_rays(ro, rd, S, mlp, mode, occlusion_mask=m, global_step=6000, bg_mode="IZ",
                                            invalid_underground=True)
                assert ok
            for k in ("fore_valid", "pred_color", "pred_depth", "specular", "diffuse", "T_left"):
                g15[f"fg_{tag}_m{mode}_{k}"] = fo[k]
            for k in ("valid", "rgb", "depth", "specular", "diffuse", "T_left"):
                g15[f"bg_{tag}_m{mode}_{k}"] = bo[k]
            if mode == 0:
                g15[f"fg_{tag}_l2_reg_specular"] = fo["l2_reg_specular"]
                g15[f"bg_{tag}_l2_reg_specular"] = bo["l2_reg_specular"]
    nv = int(g15["fg_nomask_m0_fore_valid"].sum())
    assert 0 < nv < B, nv
    save("g15_render_masks", **g15)

    # ---- G16: HashGrid.pruning_tile_grid (hashgrid/__init__.py:138-213), the coarse-to-fine occupancy pruning, run by the
    # reference itself (its lattice, its run batching, its threshold) with this repo's C encoder standing in for the CUDA one.
    # A table with a smooth density bump so that some cells survive and some do not; same level and one 2x split.
    torch.manual_seed(16)
    pr = ref_hashgrid.HashGrid.__new__(ref_hashgrid.HashGrid)
    torch.nn.Module.__init__(pr)
    pr.device = torch.device("cpu")
    pcorner, psize = torch.tensor([0.0, 0.0, 0.0]), torch.tensor([4.0, 4.0, 4.0])
    pr.bbox_center = pcorner + psize / 2.0
    pr.bbox_size = psize * 2
    pr.min_bbox = pr.bbox_center - pr.bbox_size / 2.0
    pr.finest_resolution = (pr.bbox_size / pr.bbox_size.min() * 128).int()
    pr.base_resolution = (pr.bbox_size / pr.bbox_size.min() * 8).int()
    pres = oracle.level_resolutions(pr.base_resolution, pr.finest_resolution, 16)
    pfeat = torch.randn(16, 2 ** 10, 2) * 2.0

    class PHE(torch.nn.Module):
        def forward(self, x):
            return oracle.encode_bg(x.reshape(-1, 3).contiguous().float(), pfeat, pres).reshape(*x.shape[:-1], 32)

    pr.HE = PHE()
    pmlp = network.ShallowMLP(32)
    network.init_model(pmlp, "xavier")
    with torch.no_grad():
        pmlp.sigma_layer.mlp[0].bias.fill_(-1.0)
    g16 = {"tile_corner": pcorner, "tile_size": psize, "features": pfeat, "res": pres, "grid_resolution": np.array([8, 128])}
    g16.update({"sd." + k: v for k, v in pmlp.state_dict().items()})
    occ0 = torch.rand(8, 8, 8) < 0.6
    g16["occ0"] = occ0
    for tag, sub, step, th in (("same", False, 6000, 0.4), ("split", True, 12000, 0.35)):
        pr.sampler_log2dim = torch.tensor([3, 3, 3], dtype=torch.int32)
        pr.occupied_grid = occ0.clone()
        with torch.no_grad():
            pr.pruning_tile_grid(step, pmlp, sub_split=sub, pruning_th=th, batch_size=4096)
        g16[f"{tag}_grid"], g16[f"{tag}_log2dim"] = pr.occupied_grid, pr.sampler_log2dim
        g16[f"{tag}_step"], g16[f"{tag}_th"] = np.array(step), np.array(th)
        frac = float(pr.occupied_grid.float().mean())
        assert 0.02 < frac < 0.9, (tag, frac)
        print("G16", tag, "occupied fraction", frac, "of", tuple(pr.occupied_grid.shape))
    save("g16_pruning", **g16)

    # ---- G18 (round 6): render_batch_rays(out_normal=True) (hashgrid/__init__.py:576-588): surface normals = -d(sigma)/d(sample
    # position), normalised, composited with the weights -- the reference's own autograd through ITS decoder and compositing, with
    # this repo's C oracle as the encoder (its point gradient is the adjoint the CUDA op returns).  The G15 tile, table and decoder.
    zz = torch.full((ro.shape[0], S), -1.0)
    dd = torch.full((ro.shape[0], S), -1.0)
    hgm.samplePoints = None   # (not used: the sampler op is called directly)
    spg(ro, rd, zz, dd, hgm.bbox_center - hgm.bbox_size / 4.0, hgm.bbox_size / 2.0, occ.numpy(), hgm.sampler_log2dim.numpy())
    v = torch.all(zz != -1, dim=-1)
    rov = ro[v].clone().requires_grad_(True)
    out, ok = hgm.render_batch_rays(rov, rd[v], zz[v], dd[v], mlp, 0, hgm.contract_fore, out_normal=True, infinity=False, global_step=20000)
    assert ok and bool(torch.isfinite(out["normal"]).all())
    save("g18_normals", rays_o=ro[v], rays_d=rd[v], z_vals=zz[v], dists=dd[v], normal=out["normal"], rgb=out["rgb"], depth=out["depth"],
         global_step=np.array(20000))

    # ---- G19 (round 6): the compositing sequence of render_batch_rays -- cal_integrate_weight + accumulate x 4 + the detached-weight
    # l2_reg_specular sum (hashgrid/__init__.py:344-366, :564-574, :591-594) -- run by the REFERENCE's own methods under torch
    # autograd: outputs AND the gradients of a random linear functional of every output w.r.t. sigma, the colours and rays_d.
    # Pins csrc/composite.hip's forward and adjoint (scanerf_composite_forward / _backward).
    gg = torch.Generator().manual_seed(19)
    Bc, Sc = 37, 48
    g19 = {}
    for inf in (False, True):
        sigma = (torch.rand(Bc, Sc, 1, generator=gg) ** 3 * 8).requires_grad_(True)
        dif, spc, tnt = (torch.rand(Bc, Sc, 3, generator=gg).requires_grad_(True) for _ in range(3))
        zc = torch.cumsum(torch.rand(Bc, Sc, generator=gg) * 0.1 + 0.01, 1)
        dc = torch.cat([zc[:, 1:] - zc[:, :-1], torch.full((Bc, 1), 1e-6)], 1)
        rdc = (torch.randn(Bc, 3, generator=gg) * (0.5 + torch.rand(Bc, 1, generator=gg))).requires_grad_(True)
        w, T_left = hgm.cal_integrate_weight(sigma, zc, dc, rdc, infinity=inf)
        depth = hgm.accumulate(w, zc[..., None])
        tint_o, dif_o = hgm.accumulate(w, tnt), hgm.accumulate(w, dif)
        spec_o = hgm.accumulate(w, tnt * spc)
        rgb = torch.clamp(dif_o + spec_o, 0, 1)
        l2 = torch.mean(hgm.accumulate(w.detach(), (spc - 0) ** 2))
        cw = {k: torch.randn(*v.shape, generator=gg) for k, v in (("rgb", rgb), ("depth", depth), ("T", T_left), ("dif", dif_o),
                                                                   ("spec", spec_o), ("tint", tint_o), ("w", w))}
        loss = ((rgb * cw["rgb"]).sum() + (depth * cw["depth"]).sum() + (T_left * cw["T"]).sum() + (dif_o * cw["dif"]).sum()
                + (spec_o * cw["spec"]).sum() + (tint_o * cw["tint"]).sum() + 0.1 * (w * cw["w"]).sum() + 0.37 * l2)
        loss.backward()
        t = "inf%d_" % inf
        g19.update({t + "sigma": sigma, t + "diffuse": dif, t + "specular": spc, t + "tint": tnt, t + "z_vals": zc, t + "dists": dc,
                    t + "rays_d": rdc, t + "weights": w, t + "T_left": T_left, t + "depth": depth, t + "tint_out": tint_o,
                    t + "diffuse_out": dif_o, t + "specular_out": spec_o, t + "rgb": rgb, t + "l2_reg_specular": l2,
                    t + "g_sigma": sigma.grad, t + "g_diffuse": dif.grad, t + "g_specular": spc.grad, t + "g_tint": tnt.grad,
                    t + "g_rays_d": rdc.grad})
        g19.update({t + "cw_" + k: v for k, v in cw.items()})
    save("g19_composite_grads", **g19)

    # ---- G20 (round 6): the GRADIENTS of render_batch_rays as the reference's own autograd gives them (its decoder module, its
    # compositing, its contraction; the C oracle's encoder adjoint underneath) for a loss with a term on every output -- w.r.t. the
    # hash table, every decoder parameter and both ray tensors; foreground (contract_fore) and background (contract_bg, infinity).
    # Pins the fused backward kernels' results (render.FusedRenderRays) to the reference, not only to the oracle's autograd.
    g20 = {"features": feats.detach().clone(), "res": res, "tile_corner": corner, "tile_size": size, "global_step": np.array(7000)}
    g20.update({"sd." + k: v for k, v in sd.items()})
    gg = torch.Generator().manual_seed(20)
    for tag, cfn, inf in (("fg", hgm.contract_fore, False), ("bg", hgm.contract_bg, True)):
        Bq, Sq = 24, 32
        roq = ((torch.rand(Bq, 3, generator=gg) - 0.5) * 6 + hgm.bbox_center).requires_grad_(True)
        # (unit directions x [0.5, 1.5]: with |o - centre| <= 3 and z <= 3.2 every foreground sample stays inside the 2x box, the
        # encoder's domain [-2, 2] -- hashgrid_bg_kernel.cu does not clamp, and neither does anything here)
        rdq = (torch.nn.functional.normalize(torch.randn(Bq, 3, generator=gg), dim=-1) * (0.5 + torch.rand(Bq, 1, generator=gg))).requires_grad_(True)
        if inf:
            zq = torch.sort(9 + torch.rand(Bq, Sq, generator=gg) * 60, 1).values
            dq = torch.cat([zq[:, 1:] - zq[:, :-1], torch.full((Bq, 1), 1e-6)], 1)
        else:
            zq = torch.sort(0.2 + torch.rand(Bq, Sq, generator=gg) * 3, 1).values
            dq = torch.cat([zq[:, 1:] - zq[:, :-1], torch.full((Bq, 1), 0.05)], 1)
        feats.grad = None
        feats.requires_grad_(True)
        mlp.zero_grad()
        with torch.no_grad():
            chk = cfn((roq[:, None, :] + zq[..., None] * rdq[:, None, :]).reshape(-1, 3))[0]
            assert float(chk.abs().max()) <= 2.0, float(chk.abs().max())
        out, ok = hgm.render_batch_rays(roq, rdq, zq, dq, mlp, 0, cfn, out_normal=False, infinity=inf, global_step=7000)
        assert ok
        cw = {k: torch.randn(*out[k].shape, generator=gg) for k in ("rgb", "depth", "T_left", "diffuse", "specular", "tint")}
        loss = sum((out[k] * cw[k]).sum() for k in cw) + 0.37 * out["l2_reg_specular"] + 0.1 * (out["depth"][:, 0] * out["T_left"]).sum()
        loss.backward()
        g20.update({f"{tag}_rays_o": roq, f"{tag}_rays_d": rdq, f"{tag}_z_vals": zq, f"{tag}_dists": dq, f"{tag}_loss": loss,
                    f"{tag}_g_features": feats.grad.clone(), f"{tag}_g_rays_o": roq.grad, f"{tag}_g_rays_d": rdq.grad})
        g20.update({f"{tag}_cw_{k}": v for k, v in cw.items()})
        g20.update({f"{tag}_g_sd.{k}": p_.grad.clone() for k, p_ in mlp.named_parameters()})
        feats.requires_grad_(False)
    save("g20_render_grads", **g20)


if __name__ == "__main__":
    main()
