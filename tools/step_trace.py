"""Digest every input and output of the training step's kernels calls over three iterations, in two runs from the same state in
one process; print the first call whose digests differ (tools/poison_empty.py: the first run of a process can differ from the
later ones although nothing in the state does)."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd  # noqa
from scanerf_amd import render
from scanerf_amd.tile_model import TileModel, train_step_fused
DEV = "cuda:0"
torch.manual_seed(11)
B, S = 8192, 128
o = torch.rand(B, 3, device=DEV) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1) * (0.5 + torch.rand(B, 1, device=DEV))
tgt = torch.rand(B, 3, device=DEV)


def dig(t):
    if t is None:
        return "none"
    if hasattr(t, "workspace"):
        t = t.workspace
    if not torch.is_tensor(t):
        return str(t)[:12]
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:6] + f"@{t.data_ptr() % 4096:04x}"


LOG = []


def wrap(mod, name):
    f = getattr(mod, name)

    def g(*a, **k):
        torch.cuda.synchronize()
        ins = [dig(x) for x in a] + [f"{kk}={dig(v)}" for kk, v in k.items()]
        r = f(*a, **k)
        torch.cuda.synchronize()
        outs = [dig(x) for x in (r if isinstance(r, tuple) else (r,))]
        after = [f"{kk}:{dig(v)}" for kk, v in k.items() if torch.is_tensor(v)]
        LOG[-1].append((name, ins, outs, after))
        return r
    setattr(mod, name, g)


for n in ("ray_valid", "compact_rays", "render_forward", "photometric_loss_grad", "render_backward", "scatter_accumulate_adam"):
    wrap(render, n)
for run in range(int(os.environ.get("RUNS", 4))):
    LOG.append([])
    m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=16, seed=1)
    with torch.no_grad():
        m.features.mul_(100.0)
    opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    for i in range(3):
        train_step_fused(m, opt, o, d, tgt, S, 20000 + i)
strip = lambda s: s.split("@")[0] if "@" in s else s
for run in range(1, len(LOG)):
    for c, (a, b) in enumerate(zip(LOG[0], LOG[run])):
        da = [strip(x) for x in a[1] + a[2] + a[3]]
        db = [strip(x) for x in b[1] + b[2] + b[3]]
        if da != db:
            print(f"run {run}: call {c} ({a[0]}) differs")
            print("   run 0 in :", a[1]); print(f"   run {run} in :", b[1])
            print("   run 0 out:", a[2], a[3]); print(f"   run {run} out:", b[2], b[3])
            break
    else:
        print(f"run {run}: identical to run 0")

# ---- detail: the first forward of each run against run 0's (out rows / x-stash rows that differ)
print("detail of the first render_forward per run")
KEEP = []
_orig_fwd = render.render_forward
def fwd_keep(*a, **k):
    r = _orig_fwd(*a, **k)
    torch.cuda.synchronize()
    if len(KEEP) < RUNS2 and not KEEP_BUSY[0]:
        KEEP_BUSY[0] = True
        KEEP.append((r[0].clone(), k["xstash"].clone(), k["tile_T"].clone(), a[5].data_ptr() % 4096))
    return r
RUNS2 = 8
KEEP_BUSY = [False]
render.render_forward = fwd_keep
for run in range(RUNS2):
    KEEP_BUSY[0] = False
    m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=16, seed=1)
    with torch.no_grad():
        m.features.mul_(100.0)
    opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    train_step_fused(m, opt, o, d, tgt, S, 20000)
o0, x0, t0, _ = KEEP[0]
for run in range(1, len(KEEP)):
    o1, x1, t1, off = KEEP[run]
    rows = (o1 != o0).any(1).nonzero().flatten()
    xr = (x1 != x0).view(B, S, 32)
    print(f"run {run}: resolutions @{off:04x}; out rows differing {rows.numel()} {rows[:8].tolist()}; x-stash: rays {int(xr.any(2).any(1).sum())}, "
          f"samples {int(xr.any(2).sum())}, per feature {xr.sum((0, 1)).tolist()}")
    if rows.numel():
        r = int(rows[0]); print("   out row", r, o0[r].tolist()[:8], o1[r].tolist()[:8])
