import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import scanerf_amd
from oracle import oracle as O
from scanerf_amd._capi import check, lib, stream, workspace
DEV="cuda:0"
rng = np.random.default_rng(11)
N, L, T = 30011, 16, 2 ** 13
res = O.level_resolutions(torch.tensor([32, 32, 32]), torch.tensor([2048, 2048, 2048])).numpy()
pts = rng.uniform(-2, 2, (N, 3)).astype(np.float32)
special = int(os.environ.get("SPECIAL", "50"))
pts[:special] = 2.0
feat = (rng.normal(size=(L, T, 2)) * 0.5).astype(np.float32)
gin = rng.normal(size=(N, L, 2)).astype(np.float32)
_, gf_ref = O.embedding_backward(pts, gin, feat, res)
need = lib().scanerf_embedding_bwd_workspace_bytes(N, L, T)
g = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
for ws_bytes in (need, 1 << 20):
    ws = workspace(DEV, need)
    gf = torch.zeros(L, T, 2, device=DEV)
    P, G, R = g(pts), g(gin), g(res)
    check(lib().scanerf_embedding_bg_backward_binned(ctypes.c_void_p(P.data_ptr()), ctypes.c_void_p(G.data_ptr()), ctypes.c_void_p(gf.data_ptr()),
          ctypes.c_void_p(R.data_ptr()), N, L, T, 0, ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws_bytes), stream()), "b")
    out = gf.cpu().numpy()
    bad = np.argwhere(np.abs(out - gf_ref) > 3e-4 + 1e-3 * np.abs(gf_ref))
    print("ws", ws_bytes, "bad", len(bad), "levels", np.unique(bad[:, 0]) if len(bad) else None)
    if len(bad):
        for b in bad[:12]:
            print(b, out[tuple(b)], gf_ref[tuple(b)])
        # which corners do the (2,2,2) points touch at level 15?
        for lv in np.unique(bad[:, 0])[:3]:
            r = int(res[lv, 0]); b = r - 1
            ids = sorted({O.hash_index(b + dx, b + dy, b + dz, T) for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)})
            print("level", lv, "res", r, "special corner ids", ids, "bad ids", sorted(set(bad[bad[:, 0] == lv][:, 1]))[:16])
