"""What is the wrong value?  (the forward kernel built with H3_OPAQUE_ADDR=1 writes, in ~1 of 500 cold launches,
one wrong encoder output for lanes 48-63 of a tile.)  For every faulty x-stash element: the right value, the wrong value, and the
candidates -- the 8 corner values of the half-wave's LAST level (15: the loads whose destination registers the copy reuses)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scanerf_amd  # noqa
from scanerf_amd import render
from scanerf_amd.tile_model import TileModel, train_step_fused
DEV = "cuda:0"
torch.manual_seed(11)
B, S = 8192, 128
RUNS = int(os.environ.get("RUNS", 300))
o = torch.rand(B, 3, device=DEV) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1) * (0.5 + torch.rand(B, 1, device=DEV))
tgt = torch.rand(B, 3, device=DEV)
BUSY = int(os.environ.get("BUSY", "0"))  # number of 4096^3 f32 matrix products queued right before every step
BUSY_A = torch.randn(4096, 4096, device=DEV) if BUSY else None
BUSY_C = torch.empty(4096, 4096, device=DEV) if BUSY else None
KEEP = []
_fwd = render.render_forward


def fwd_keep(*a, **k):
    r = _fwd(*a, **k)
    torch.cuda.synchronize()
    KEEP.append((k["xstash"].clone(), a[2].clone()))
    return r


render.render_forward = fwd_keep
stashes = []
REUSE = bool(int(os.environ.get("REUSE", "0")))  # one model whose state is reset in place: the same addresses in every run
m0 = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=16, seed=1)
init = (m0.features.detach().clone() * 100.0, m0.decoder.params.detach().clone())
for run in range(RUNS):
    KEEP.clear()
    if REUSE:
        m = m0
        with torch.no_grad():
            m.features.copy_(init[0]); m.decoder.params.copy_(init[1]); m.exp_avg.zero_(); m.exp_avg_sq.zero_()
        m.adam_step = 0
    else:
        m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=16, seed=1)
        with torch.no_grad():
            m.features.mul_(100.0)
    opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    if BUSY:  # keep the GPU busy (and its clocks up) until the step's kernels arrive: is it the idle period before the launch?
        for _ in range(BUSY):
            BUSY_C.copy_(BUSY_A @ BUSY_A)
    train_step_fused(m, opt, o, d, tgt, S, 20000)
    xs, z = KEEP[0]
    if len(stashes) < 3:
        stashes.append(xs)
        if len(stashes) == 3:   # the reference = the value two of the first three runs agree on, element by element
            a, b, c = stashes
            ref = torch.where(a == b, a, c)
            for k, t in enumerate(stashes):
                if not torch.equal(t, ref):
                    stashes.append(None); stashes[k] = None
                    xs = t
                    break
            else:
                continue
        else:
            continue
    if torch.equal(xs, ref):
        continue
    R3, X3 = ref.view(B, S, 32), xs.view(B, S, 32)
    dif = (X3 != R3).nonzero()
    print(f"run {run}: {dif.shape[0]} faulty elements; features {sorted(set(dif[:, 2].tolist()))}; ray {dif[0, 0].item()} samples {dif[:, 1].min().item()}..{dif[:, 1].max().item()}")
    for (ray, s, f) in dif[:4].tolist():
        good, bad = R3[ray, s, f].item(), X3[ray, s, f].item()
        where = (R3 == bad).nonzero()[:6].tolist()
        print(f"   (ray {ray}, sample {s}, feature {f}): right {good:+.6e} wrong {bad:+.6e}; the wrong value is the right value of (ray, sample, feature) {where}", flush=True)
print("done")
