"""Which stage of the default training step differs between repeats from the same state?  The stages of train_step_fused
(sampler, forward with the scatter plan, loss, t16 backward with 8-byte records, accumulate + sparse Adam) run REPS times on
the same inputs; every stage's outputs are hashed.  All columns must be equal."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd  # noqa
from scanerf_amd import network, render
from scanerf_amd.tile_model import TileModel
DEV = "cuda:0"
torch.manual_seed(11)
B, S = int(os.environ.get("B", 8192)), 128
REPS = int(os.environ.get("REPS", 12))
o = torch.rand(B, 3, device=DEV) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1) * (0.5 + torch.rand(B, 1, device=DEV))
tgt = torch.rand(B, 3, device=DEV)


def dig(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().cpu().numpy().tobytes())
    return h.hexdigest()[:8]


rows = []
for rep in range(REPS):
    m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=int(os.environ.get("LOG2T", 16)), seed=1)
    with torch.no_grad():
        m.features.mul_(100.0)
    T = m.features.shape[1]
    row = {}
    with torch.no_grad():
        z, dist = m.sample(o, d, S)
        row["sample"] = dig(z, dist)
        valid = render.ray_valid(z)
        wf = m.weight_feature(20000)
        blob = m.decoder.blob()
        m.packed.pack(blob, wf, network.skip_levels(20000))
        row["pack"] = dig(m.packed.workspace)
        tile_T = torch.empty((B, render.tile_T_columns(S)), device=DEV)
        xstash = torch.empty((B * S, 32), device=DEV)
        box = (m.min_bbox.tolist(), m.bbox_size.tolist(), render.FORE, False)
        out, _, ws = render.render_forward(o, d, z, dist, m.features, m.resolution, m.packed, *box, ray_valid=valid,
                                           want_weights=False, tile_T=tile_T, xstash=xstash, plan=True)
        row["fwd out"] = dig(out)
        row["fwd T"] = dig(tile_T)
        row["fwd x"] = dig(xstash)
        head = 1 << 16
        row["plan"] = dig(ws[:head])
        loss, grad_out = render.photometric_loss_grad(out, tgt, valid, 0.01)
        row["loss"] = dig(loss, grad_out)
        gtab = m.overflow_grad()
        gblob = torch.zeros(network.PARAMSIZE, device=DEV)
        render.render_backward(o, d, z, dist, m.features, m.resolution, m.packed, wf, *box, out, tile_T, grad_out, ray_valid=valid,
                               grad_blob=gblob, xstash=xstash, scatter=(ws, gtab), want_dfeat=False, arith=render._capi.ARITH_T16)
        row["bwd gblob"] = dig(gblob)
        torch.cuda.synchronize()
        # the records as a multiset (their order within a bucket may differ legitimately? no: fixed cursors) -- hash the raw stream
        if os.environ.get("RECS"):   # (hundreds of MB through the host: slow; the order within a bucket differs legitimately)
            row["bwd recs"] = dig(ws)
        render.scatter_accumulate_adam(ws, m.features.data, m.exp_avg, m.exp_avg_sq, 1e-2, 0.9, 0.99, 1e-15, m.adam_step, B, S,
                                       half_table=m._half_table, overflow_grad=gtab)
        row["table"] = dig(m.features, m.exp_avg, m.exp_avg_sq)
    rows.append(row)
    if rep % 50 == 49:
        print(f"  {rep + 1} repeats", flush=True)
keys = list(rows[0].keys())
if not os.environ.get("QUIET"):
    print(" ".join(f"{k:>10s}" for k in keys))
    for r in rows:
        print(" ".join(f"{r[k]:>10s}" for k in keys))
from collections import Counter
for k in keys:
    if k != "bwd recs":
        c = Counter(r[k] for r in rows)
        if len(c) > 1:
            print(f"  stage {k}: {len(c)} distinct; odd runs {[i for i, r in enumerate(rows) if r[k] != c.most_common(1)[0][0]]}")
for k in keys:
    n = len(set(r[k] for r in rows))
    if n > 1 and k != "bwd recs":
        print(f"FIRST DIFFERING STAGE: {k} ({n} distinct of {REPS})")
        break
else:
    print("all stages identical")
