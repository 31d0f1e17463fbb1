"""Capture golden vectors from the reference's importable Python half.

Run ONCE in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

It imports the reference's pure-Python modules (network.py, hashgrid.HashGrid's
torch methods, camera.py, consensus.py, scheduler.py) with the missing native /
third-party names stubbed in sys.modules, feeds them seeded inputs and writes the
inputs + outputs to tests/golden/*.npz.  Only DATA is written; no reference text
is copied.  The native hash encoder the reference would call (CUDA only) is
replaced by this repo's C oracle, so G6 pins the composition
contract -> encode -> MLP -> composite of the Python half, not the encoder.

Nothing here runs on the GPU box (the reference does not travel).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)


def _stub_modules():
    ed = types.ModuleType("easydict")

    class EasyDict(dict):
        __getattr__ = dict.get
        __setattr__ = dict.__setitem__

    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed

    def raising(*a, **k):
        raise RuntimeError("native op stub (CUDA-only in the reference)")

    cu = types.ModuleType("cuda")
    for n in ["ray_aabb_intersection", "sample_points_contract", "voxelize_mesh", "sample_points_grid",
              "compute_ray_forward", "compute_ray_backward"]:
        setattr(cu, n, raising)
    sys.modules["cuda"] = cu
    hl = types.ModuleType("hashgrid.lib")
    hl.__path__ = []
    hg = types.ModuleType("hashgrid.lib.HASHGRID")
    hg.Sampler = type("Sampler", (), {})
    for n in ["ray_block_intersection", "sample_points", "prepare_points", "sort_by_key", "pts_inference",
              "accumulate_color", "ray_firsthit_block", "inverse_z_sampling", "bg_pts_inference",
              "get_last_block", "update_outgoing_bidx", "update_outgoing_bidx_v2", "bg_pts_inference_v2",
              "process_occupied_grid", "embedding_forward_cuda", "embedding_backward_cuda",
              "embedding_bg_forward_cuda", "embedding_bg_backward_cuda"]:
        setattr(hg, n, raising)
    sys.modules["hashgrid.lib"] = hl
    sys.modules["hashgrid.lib.HASHGRID"] = hg
    tl = types.ModuleType("tools")
    tl.__path__ = []
    tl.tools = types.ModuleType("tools.tools")
    sys.modules["tools"] = tl
    sys.modules["tools.tools"] = tl.tools


def main():
    _stub_modules()
    sys.path.insert(0, REF)
    import network  # noqa
    import camera  # noqa
    import consensus  # noqa
    import scheduler  # noqa
    import hashgrid as ref_hashgrid  # noqa
    from oracle import oracle  # this repo's oracle: provides the CPU encoder for G6

    def save(name, **kw):
        kw = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in kw.items()}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **kw)
        print("wrote", name, {k: v.shape for k, v in kw.items()})

    # ---- G1: ShallowMLP (network.py:151-190), xavier weights + small non-zero biases
    torch.manual_seed(0)
    mlp = network.ShallowMLP(32)
    network.init_model(mlp, "xavier")
    with torch.no_grad():
        for n, p in mlp.named_parameters():
            if n.endswith("bias"):
                p.copy_(0.05 * torch.randn_like(p))
    x = torch.cat([0.3 * torch.randn(64, 32), torch.randn(64, 3) * torch.rand(64, 1) * 2], -1)
    wf = torch.linspace(0.2, 1.0, 32)[None, :]
    with torch.no_grad():
        o = mlp(x, weight_feature=wf)
        sig_only = mlp.inference_sigma(x[:, :32])
    sd = {k: v for k, v in mlp.state_dict().items()}
    save("g1_mlp", x=x, weight_feature=wf, sigma=o["sigma"], diffuse=o["diffuse"], specular=o["specular"],
         tint=o["tint"], sigma_only=sig_only, **{"sd." + k: v for k, v in sd.items()})

    # ---- G2: sh_encoding deg 3 (network.py:38-77)
    d = torch.nn.functional.normalize(torch.randn(16, 3), dim=-1)
    save("g2_sh", dirs=d, sh=network.sh_encoding(3, d))

    # ---- hashgrid.HashGrid pure-torch methods, without its native-dependent ctor
    hgm = ref_hashgrid.HashGrid.__new__(ref_hashgrid.HashGrid)
    torch.nn.Module.__init__(hgm)
    hgm.device = torch.device("cpu")
    corner = torch.tensor([-4.0, -4.0, -4.0])
    size = torch.tensor([8.0, 8.0, 8.0])
    hgm.bbox_center = corner + size / 2.0
    hgm.bbox_size = size * 2
    hgm.min_bbox = hgm.bbox_center - hgm.bbox_size / 2.0

    # ---- G3: cal_integrate_weight + accumulate (hashgrid/__init__.py:344-366)
    sigma = torch.rand(8, 16, 1) * 3
    z = torch.sort(torch.rand(8, 16) * 5 + 0.5, dim=-1)[0]
    dists = torch.rand(8, 16) * 0.3 + 0.01
    rd = torch.randn(8, 3)
    attr = torch.rand(8, 16, 3)
    g3 = {"sigma": sigma, "z": z, "dists": dists, "rays_d": rd, "attr": attr}
    for inf in (False, True):
        w, tl = hgm.cal_integrate_weight(sigma, z, dists.clone(), rd, infinity=inf)
        g3["weights_inf%d" % inf] = w
        g3["T_left_inf%d" % inf] = tl
        g3["acc_inf%d" % inf] = hgm.accumulate(w, attr)
    save("g3_composite", **g3)

    # ---- G4: contraction (hashgrid/__init__.py:394-411)
    p = (torch.rand(64, 3) - 0.5) * 40
    save("g4_contract", pts=p, min_bbox=hgm.min_bbox, bbox_size=hgm.bbox_size,
         fore=hgm.contract_fore(p)[0], bg=hgm.contract_bg(p)[0])

    # ---- G5: weight_feature (hashgrid/__init__.py:228-235)
    steps = [0, 1, 2500, 5000, 7777, 10000, 40000]
    save("g5_weight_feature", steps=np.array(steps), w=torch.stack([hgm.weight_feature(s) for s in steps], 0))

    # ---- G6: render_batch_rays end to end (hashgrid/__init__.py:512-596), fg + bg, TRAIN + INFERENCE
    T = 2 ** 12
    res = oracle.level_resolutions(torch.tensor([32, 32, 32]), torch.tensor([2048, 2048, 2048]), 16)
    feats = torch.randn(16, T, 2) * 0.5

    class HE(torch.nn.Module):  # stands in for PyHashGridBG (native): this repo's C oracle encoder
        def forward(self, x):
            return oracle.encode_bg(x.reshape(-1, 3).contiguous(), feats, res).reshape(*x.shape[:-1], 32)

    hgm.HE = HE()
    B, S = 12, 24
    ro = (torch.rand(B, 3) - 0.5) * 6
    rdir = torch.randn(B, 3) * (0.5 + torch.rand(B, 1))
    zf = torch.sort(torch.rand(B, S) * 3 + 0.2, dim=-1)[0]
    df = torch.cat([zf[:, 1:] - zf[:, :-1], torch.full((B, 1), 0.05)], -1)
    zb = torch.sort(torch.rand(B, S) * 60 + 9, dim=-1)[0]
    db = torch.cat([zb[:, 1:] - zb[:, :-1], torch.full((B, 1), 1e-6)], -1)
    g6 = {"rays_o": ro, "rays_d": rdir, "z_fg": zf, "d_fg": df, "z_bg": zb, "d_bg": db, "features": feats,
          "res": res, "global_step": np.array(2500)}
    for tag, zz, dd, fn, inf in (("fg", zf, df, hgm.contract_fore, False), ("bg", zb, db, hgm.contract_bg, True)):
        for mode in (0, 1):
            with torch.no_grad():
                out, ok = hgm.render_batch_rays(ro, rdir, zz, dd.clone(), mlp, mode, fn, out_normal=False,
                                                infinity=inf, global_step=2500)
            assert ok
            for k, v in out.items():
                g6["%s_m%d_%s" % (tag, mode, k)] = v
    save("g6_render_batch", **g6)

    # ---- G7: camera algebra + ray generation (camera.py:84-95, :259-281)
    se3 = torch.randn(5, 6) * 0.1
    SE3 = camera.lie.se3_to_SE3(se3)
    c2w = torch.cat([torch.linalg.qr(torch.randn(5, 3, 3))[0], torch.randn(5, 3, 1)], -1)
    w2c = camera.pose.invert(c2w)
    comp = camera.pose.compose([SE3, w2c])
    H, W = 6, 8
    ks = torch.tensor([[100.0, 0, 4.2, 0, 110.0, 2.9, 0, 0, 1]]).repeat(5, 1).reshape(5, 3, 3)
    ray_idx = torch.tensor([0, 7, 13, 22, 47])
    cen, ray = camera.get_center_and_ray_v2(H, W, comp, ks, ray_idx)
    save("g7_camera", se3=se3, SE3=SE3, c2w=c2w, w2c=w2c, composed=comp, composed_inv=camera.pose.invert(comp),
         ks=ks, ray_idx=ray_idx, H=np.array(H), W=np.array(W), center=cen, ray=ray)

    # ---- G8: ConsensusManager (consensus.py:18-21,40-50,70-76)
    cm = consensus.ConsensusManager.__new__(consensus.ConsensusManager)
    M = 10
    cm.cfg = types.SimpleNamespace(TILEIDX=0, RHO=0.05)
    cm.device = torch.device("cpu")
    cm.tile = types.SimpleNamespace(num_camera=M)
    cm.poses = types.SimpleNamespace(se3_refine=torch.randn(M, 6) * 0.01)
    cm.shared_se3 = cm.poses.se3_refine.clone()
    cm.delta_se3 = torch.randn(M, 6) * 0.001
    delta0 = cm.delta_se3.clone()
    cm.overlap_flags = torch.zeros(M, dtype=torch.bool)
    cm.rho = torch.ones(6) * 0.05
    shared = torch.randn(M, 6) * 0.01
    ov = torch.tensor([1, 4, 5])
    cm.update(shared, ov)
    save("g8_consensus", se3_refine=cm.poses.se3_refine, delta0=delta0, shared=shared, overlap_idxs=ov,
         delta1=cm.delta_se3, flags=cm.overlap_flags, rho=cm.rho, loss=cm.camera_loss())

    # ---- G9: Scheduler eta(step) (scheduler.py:15-52)
    sch = scheduler.Scheduler("grid", 1e-2, 1e-4, 40000)
    opt = types.SimpleNamespace(param_groups=[{"lr": 0.0}])
    etas = []
    for s in (0, 100, 5000, 20000, 39999):
        sch.step(s, opt)
        etas.append(opt.param_groups[0]["lr"])
    save("g9_scheduler", steps=np.array([0, 100, 5000, 20000, 39999]), eta=np.array(etas, np.float64),
         start=np.array(1e-2), end=np.array(1e-4), iters=np.array(40000))

    # ---- G10: torch.optim.Adam (the live optimiser, tile.py:301) for two steps, all grads non-zero
    pr = torch.nn.Parameter(torch.randn(32, 8))
    p0 = pr.detach().clone()
    adam = torch.optim.Adam([pr], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    grads = [torch.randn(32, 8), torch.randn(32, 8)]
    snaps = []
    for g in grads:
        pr.grad = g.clone()
        adam.step()
        snaps.append(pr.detach().clone())
    save("g10_adam", p0=p0, g0=grads[0], g1=grads[1], p1=snaps[0], p2=snaps[1])


if __name__ == "__main__":
    main()
