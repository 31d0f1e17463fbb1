"""Golden fixture for the per-tile export formats: files written by the REFERENCE's own writer.

Run ONCE in the build container (where /root/reference exists):

    python tests/golden/make_golden_formats.py

G12.  `HashGrid.export` (hashgrid/__init__.py:248-257) is called on a HashGrid built through the same `__new__` path
make_golden.py uses for the pure-torch methods (its constructor needs the CUDA-only extension), with a small table
(T = 2^6) -- it writes tests/golden/g12_tile/feature.npz.  The decoder goes to tests/golden/g12_tile/decoder.pth by the
statement tile.py:521 uses (torch.save of ShallowMLP.state_dict()).  The consumer side is captured too: the reference's
`tools.utils.extract_MLP_para` (tools/utils.py:399-410, loaded with its absent third-party imports cv2 / imageio /
easydict stubbed by name) reads that decoder.pth back, and the expected render-time blob is assembled from ITS return
values in the order rendering.py:101-112 gives ([bias, W^T flattened] per layer); g12_expected.npz also holds what
rendering.py:164-165 makes of block_corner / block_size.  Only DATA is written; nothing here runs on the GPU box.
"""
import importlib
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def main():
    import make_golden
    make_golden._stub_modules()
    sys.path.insert(0, REF)
    import network  # noqa
    import hashgrid as ref_hashgrid  # noqa

    torch.manual_seed(12)
    hgm = ref_hashgrid.HashGrid.__new__(ref_hashgrid.HashGrid)
    torch.nn.Module.__init__(hgm)
    hgm.device = torch.device("cpu")
    corner = torch.tensor([-3.0, 1.0, 5.0])
    size = torch.tensor([8.0, 4.0, 8.0])
    hgm.bbox_center = corner + size / 2.0
    hgm.bbox_size = size * 2                                  # hashgrid/__init__.py:50
    hgm.min_bbox = hgm.bbox_center - hgm.bbox_size / 2.0
    hgm.sampler_log2dim = torch.tensor([3, 2, 3], dtype=torch.int32)
    hgm.occupied_grid = torch.rand(8, 4, 8) < 0.4
    L, T = 16, 2 ** 6
    base = (hgm.bbox_size / hgm.bbox_size.min() * 4).int()
    fin = (hgm.bbox_size / hgm.bbox_size.min() * 64).int()
    from oracle import oracle
    res = oracle.level_resolutions(base, fin, L)
    hgm.HE = types.SimpleNamespace(features=torch.nn.Parameter(torch.randn(L, T, 2) * 0.7), resolution=res)

    out_dir = os.path.join(HERE, "g12_tile")
    os.makedirs(out_dir, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        hgm.export(tmp)                                       # the reference's writer
        shutil.copy(os.path.join(tmp, "feature.npz"), os.path.join(out_dir, "feature.npz"))

    mlp = network.ShallowMLP(32)
    network.init_model(mlp, "xavier")
    with torch.no_grad():
        for n, p in mlp.named_parameters():
            if n.endswith("bias"):
                p.copy_(0.05 * torch.randn_like(p))
    torch.save(mlp.state_dict(), os.path.join(out_dir, "decoder.pth"))   # tile.py:521

    # ---- the reference's reader of decoder.pth
    for name in ("cv2", "imageio"):
        sys.modules.setdefault(name, types.ModuleType(name))
    for k in [k for k in sys.modules if k == "tools" or k.startswith("tools.")]:
        del sys.modules[k]                                    # (make_golden stubs `tools`; here the real package is wanted)
    ref_utils = importlib.import_module("tools.utils")
    weights, bias = ref_utils.extract_MLP_para(os.path.join(out_dir, "decoder.pth"))
    parts = []
    for w, b in zip(weights, bias):                           # order of rendering.py:101-112
        parts += [b, w.transpose(1, 0).flatten()]
    blob = torch.cat(parts, 0)
    f = np.load(os.path.join(out_dir, "feature.npz"))
    bc, bs = f["block_corner"], f["block_size"]
    np.savez_compressed(os.path.join(HERE, "g12_expected.npz"), blob=blob.numpy(),
                        features_f32=hgm.HE.features.detach().numpy(), occupied_grid=hgm.occupied_grid.numpy(),
                        render_block_corner=bc + bs / 4.0, render_block_size=bs / 2.0,   # rendering.py:164-165
                        tile_corner=corner.numpy(), tile_size=size.numpy(), resolution=res.numpy(),
                        grid_log2dim=hgm.sampler_log2dim.numpy())
    print("wrote g12_tile/feature.npz", {k: (f[k].shape, f[k].dtype) for k in f.files})
    print("wrote g12_tile/decoder.pth, g12_expected.npz blob", blob.shape)


if __name__ == "__main__":
    main()
