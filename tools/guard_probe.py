"""Are the wait-state guards of the forward kernel (closed MFMA regions, operand guard, store guard: csrc/common.h, render_h3.h)
still needed once the kernel holds no packed-f32 arithmetic?  Launch-to-launch comparison of the training forward (x-stash,
tile_T, plan counts) per table type, back to back and with the instruction caches swept in between.  SCANERF_LIB selects the
build: tools/build_variant.py <tag> render="-DH3_OPAQUE_ADDR=1 [-DSCANERF_GUARDS=1 -DH3_REGIONS=1] [-fslp-vectorize ...]" writes
<pkg>/lib/debug/libscanerf_hip_<tag>.so)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd  # noqa
from scanerf_amd import _capi, render
from scanerf_amd.tile_model import TileModel, train_step_fused
DEV = "cuda:0"
N = int(os.environ.get("N", 200))
B, S = 16384, 64
for dt in (torch.bfloat16, torch.float32, torch.float16):
    torch.manual_seed(9)
    o = torch.rand(B, 3, device=DEV) * 8 - 4
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1) * (0.5 + torch.rand(B, 1, device=DEV))
    tgt = torch.rand(B, 3, device=DEV)
    m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=14, seed=1, table_dtype=dt)
    with torch.no_grad():
        m.features.mul_(30.0)
    opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    for i in range(2):
        train_step_fused(m, opt, o, d, tgt, S, 20000 + i)
    z, dist = m.sample(o, d, S)
    m.packed.pack(m.decoder.blob(), m.weight_feature(20000))
    box = (m.min_bbox.tolist(), m.bbox_size.tolist(), render.FORE, False)
    table = m.gather_table()
    for sweep in (False, True):
        ref, bad, bad_rays = None, 0, 0
        for it in range(N):
            if sweep:
                _capi.check(_capi.lib().scanerf_icache_sweep(_capi.stream()), "sweep")
            tile_T = torch.empty(B, render.tile_T_columns(S), device=DEV)
            xs = torch.empty(B * S, 32, device=DEV)
            out = render.render_forward(o, d, z, dist, table, m.resolution, m.packed, *box, want_weights=False, tile_T=tile_T, xstash=xs,
                                        plan=render.forward_plan_supported(B, S, table.shape[1]))[0]
            if ref is None:
                ref = (out.clone(), xs.clone(), tile_T.clone())
            elif not (torch.equal(out, ref[0]) and torch.equal(xs, ref[1]) and torch.equal(tile_T, ref[2])):
                bad += 1
                bad_rays += int((out != ref[0]).any(1).sum())
        print(f"{os.environ.get('TAG', '')} table {dt} sweep {sweep}: {bad} of {N - 1} launches differ ({bad_rays} rays in all)", flush=True)
