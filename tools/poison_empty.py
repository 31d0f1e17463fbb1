"""Does any kernel of the training step read memory that nothing has written?  torch.empty is replaced by a version that fills
the new tensor with a pattern (zeros / NaN / 1e30 / -7 per repeat); the state after three iterations must not depend on it."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd  # noqa
from scanerf_amd.tile_model import TileModel, train_step_fgbg, train_step_fused
DEV = "cuda:0"
_empty = torch.empty
PATTERN = [0.0]


def poisoned(*a, **k):
    t = _empty(*a, **k)
    if t.is_cuda and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(PATTERN[0])
        else:
            t.view(torch.uint8).fill_(0 if PATTERN[0] == 0.0 else 0x5b)
    return t


torch.empty = poisoned
torch.manual_seed(11)
B, S = 8192, 128
o = torch.rand(B, 3, device=DEV) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1) * (0.5 + torch.rand(B, 1, device=DEV))
tgt = torch.rand(B, 3, device=DEV)
dig = lambda t: hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:6]
for name in ("fused", "fgbg", "fused+pose", "fgbg+pose"):
    res = {}
    for pat in (0.0, float("nan"), 1e30, -7.0, 0.0, float("nan")):
        PATTERN[0] = pat
        scanerf_amd._capi._workspaces.clear() if hasattr(scanerf_amd._capi, "_workspaces") else None
        m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=16, seed=1)
        with torch.no_grad():
            m.features.mul_(100.0)
        opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
        row = []
        for i in range(3):
            pose = name.endswith("pose")
            r = (train_step_fgbg(m, opt, o, d, tgt, S, S, 20000 + i, pose_grads=pose) if name.startswith("fgbg")
                 else train_step_fused(m, opt, o, d, tgt, S, 20000 + i, pose_grads=pose))
            loss = r[0] if pose else r
            row += [dig(loss), dig(m.decoder.blob()), dig(m.features)] + ([dig(r[1]), dig(r[2])] if pose else [])
        res.setdefault(tuple(row), []).append(pat)
    print(name, "->", len(res), "distinct results;", [v for v in res.values()])
    if len(res) > 1:
        rows = list(res.keys())
        k = next(i for i in range(len(rows[0])) if len(set(r[i] for r in rows)) > 1)
        per = 5 if name.endswith("pose") else 3
        print("   first differing entry: iteration", k // per, ("loss", "decoder", "table", "g_o", "g_d")[k % per])
