"""tests/test_gpu_determinism.py::test_whole_training_step_is_bit_reproducible with per-iteration digests of every piece of
state: which one differs first?"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd  # noqa
from scanerf_amd.tile_model import TileModel, train_step_fgbg, train_step_fused
DEV = "cuda:0"
torch.manual_seed(11)
B, S = 8192, 128
o = torch.rand(B, 3, device=DEV) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1) * (0.5 + torch.rand(B, 1, device=DEV))
tgt = torch.rand(B, 3, device=DEV)
TABLE = {"table_dtype": {"bf16": torch.bfloat16, "f16": torch.float16}[os.environ["TABLE"]]} if os.environ.get("TABLE") else {}  # resident half-precision gather table (configs[2])
POSE = bool(int(os.environ.get("POSE", "0")))  # pose_grads=True: Jacobian-stash forward, POSE backward
dig = lambda t: hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:6]
for fgbg in ((False, True) if os.environ.get("WHICH", "both") == "both" else (os.environ["WHICH"] == "fgbg",)):
    rows = []
    for rep in range(int(os.environ.get("REPS", 8))):
        m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=int(os.environ.get("LOG2T", 16)), seed=1, **TABLE)
        with torch.no_grad():
            m.features.mul_(100.0)
        opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
        row = []
        for i in range(3):
            r = (train_step_fgbg(m, opt, o, d, tgt, S, S, 20000 + i, pose_grads=POSE) if fgbg
                 else train_step_fused(m, opt, o, d, tgt, S, 20000 + i, pose_grads=POSE))
            loss = r[0] if POSE else r
            row += [dig(loss), dig(m.decoder.params.grad), dig(m.decoder.blob()), dig(m.features), dig(m.exp_avg_sq)]
            if POSE:
                row[-1] = dig(torch.cat([m.exp_avg_sq.flatten()[:1024], r[1].flatten(), r[2].flatten()]))  # (+ the ray gradients)
        rows.append(row)
    names = [f"{n}{i}" for i in range(3) for n in ("loss", "gblob", "dec", "table", "v")]
    print(("fgbg" if fgbg else "fused") + ("+pose" if POSE else ""))
    from collections import Counter
    major = [Counter(r[k] for r in rows).most_common(1)[0][0] for k in range(len(names))]
    odd = [(i, [names[k] for k in range(len(names)) if r[k] != major[k]]) for i, r in enumerate(rows)]
    odd = [(i, c) for i, c in odd if c]
    print("identical" if not odd else f"{len(odd)} of {len(rows)} runs differ from the majority: " + "; ".join(f"run {i}: {c}" for i, c in odd))
