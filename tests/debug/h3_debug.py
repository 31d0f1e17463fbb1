"""Compare the two decoder arithmetics (f32 MFMA vs f16 split) of the fused forward; run h3 twice (determinism)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import scanerf_amd
from oracle import oracle as O
from scanerf_amd import network, render
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/
from test_gpu_parity import _render_inputs, g, DEV

junk = torch.randn(64 << 20, device=DEV)  # dirty the allocator's pool
del junk
for S_, bg in ((128, False),):
    rng = np.random.default_rng(9)
    B, T = 4096, 2 ** 13
    o, d, z, dist, feat = _render_inputs(rng, B, S_, T, bg)
    sd = O.init_mlp(seed=3, bias_scale=0.05)
    res = O.level_resolutions(torch.tensor([32, 32, 32]), torch.tensor([2048, 2048, 2048]))
    mn, sz = torch.tensor([-8.0, -8.0, -8.0]), torch.tensor([16.0, 16.0, 16.0])
    step = 2500
    pk = render.PackedDecoder(DEV).pack(O.pack_blob(sd).to(DEV), network.weight_feature(step, DEV))
    outs = {}
    RUNS = ["f32"] + ["h3%d" % i for i in range(int(os.environ.get("NRUN", 3)))]
    for ar in RUNS:
        render.set_arith(ar[:2] if ar.startswith("h3") else ar)
        out, w = render.render_forward(g(o), g(d), g(z), g(dist), g(feat), g(res.numpy()), pk, mn.tolist(), sz.tolist(),
                                       render.BG if bg else render.FORE, infinity=bg)
        outs[ar] = (out.cpu().numpy(), w.cpu().numpy())
    ref = outs["f32"]
    for ar in RUNS[1:]:
        out, w = outs[ar]
        e = np.abs(out[:, :14] - ref[0][:, :14]) / (1e-6 + 1e-4 * np.abs(ref[0][:, :14]))
        ew = np.abs(w - ref[1]) / (1e-7 + 1e-4 * np.abs(ref[1]))
        bad = np.unique(np.where(e > 1)[0])
        badw = np.unique(np.where(ew > 1)[0])
        if len(bad) or ar == RUNS[1]: print(f"S={S_} bg={bg} {ar} vs f32: out max {e.max():.3f} bad rays {bad[:10]} (n={len(bad)})  weights max {ew.max():.3f} bad rays {badw[:10]} (n={len(badw)})")
        for r in bad[:3]:
            cols = np.where(e[r] > 1)[0]
            ws = np.where(ew[r] > 0.5)[0]
            print(f"    ray {r}: bad cols {cols}  weights off at samples {ws[:16]}")
            t0 = (ws[0] // 32) * 32
            np.set_printoptions(precision=5, linewidth=250)
            print("      ratio w_h3/w_f32 over the tile:", (w[r, t0:t0 + 32] / ref[1][r, t0:t0 + 32]))
            if t0 + 32 < w.shape[1]:
                print("      next tile:", (w[r, t0 + 32:t0 + 40] / ref[1][r, t0 + 32:t0 + 40]))
    print("   h3 run-to-run identical:", all(np.array_equal(outs[RUNS[1]][0], outs[r][0]) for r in RUNS[2:]))
