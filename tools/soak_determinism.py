"""Soak: the default training step (and the fg+bg iteration with pose gradients) at full size, RUNS runs of STEPS iterations from
the same state; every run must end in bit-identical table / moments / decoder (and ray gradients)."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd  # noqa
from scanerf_amd import _capi
from scanerf_amd.tile_model import TileModel, train_step_fgbg, train_step_fused
DEV = "cuda:0"
SWEEP = os.environ.get("SWEEP", "0")   # 1: the instruction caches are swept after every library call in every run but the first
ARITHS = os.environ.get("ARITH", "")  # e.g. "h3": also set the arithmetic (scanerf_amd.render.set_arith)
if ARITHS:
    from scanerf_amd import render as _r
    _r.set_arith(ARITHS)
B, S = int(os.environ.get("B", 65536)), 128
STEPS, RUNS = int(os.environ.get("STEPS", 40)), int(os.environ.get("RUNS", 3))
torch.manual_seed(1)
o = torch.rand(B, 3, device=DEV) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1) * (0.5 + torch.rand(B, 1, device=DEV))
tgt = torch.rand(B, 3, device=DEV)
for name in ("fused", "fgbg+pose"):
    digests = []
    for run in range(RUNS):
        _capi.SWEEP_ICACHE = SWEEP == "1" and run > 0
        m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=19, seed=1)
        with torch.no_grad():
            m.features.mul_(100.0)
        opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
        h = hashlib.sha256()
        for i in range(STEPS):
            step = 4000 + 100 * i      # walks through the coarse-to-fine schedule as well
            if name == "fused":
                train_step_fused(m, opt, o, d, tgt, S, step)
            else:
                _, g_o, g_d = train_step_fgbg(m, opt, o, d, tgt, S, S, step, pose_grads=True)
                if i % 10 == 9:
                    h.update(g_o.cpu().numpy().tobytes()); h.update(g_d.cpu().numpy().tobytes())
        torch.cuda.synchronize()
        for t in (m.features.detach(), m.exp_avg, m.exp_avg_sq, m.decoder.blob().detach()):
            h.update(t.cpu().numpy().tobytes())
        digests.append(h.hexdigest()[:16])
    print(f"{name}: {RUNS} runs x {STEPS} steps of {B} rays: digests {digests} -> {'identical' if len(set(digests)) == 1 else 'DIFFERENT'}", flush=True)
    _capi.SWEEP_ICACHE = False
    assert len(set(digests)) == 1
