"""On-disk formats either side of the hot path, byte-compatible with the reference's
readers and writers so tiles, cameras and checkpoints move between the two code bases unchanged.

  camera.log / refined_camera.log   7-line records                      load_data.py:60-100, tools/tools.py:66-78
  tiles/tile_info.txt               "# TILEID(1) BBOX_CORNER(3) ..."    preprocess/build_tiles.py:232-237, tile.py:102-110
  tiles/training_views.txt          two lines per tile                  preprocess/build_tiles.py:203-218, tile.py:95-100
  tile-<i>/cams.npz                 c2ws, ks, idxs                      tile.py:527-529
  checkpoint-<step>-<tile>.pt       dict of dicts                       tile.py:534-569, hashgrid/__init__.py:94-107,
                                                                        consensus.py:25-38
  *.ply (mesh for voxelize_mesh)    vertex x/y/z + face vertex lists    cuda/include/plyIO.h (tinyply)
(feature.npz / decoder.pth live in renderer.py next to the renderer that consumes them.)
"""
import os
import re

import numpy as np
import torch


# ---- camera logs -------------------------------------------------------------------------------------------------
def read_campara(path, return_shape=False):
    """-> Ks [N,3,3] f32, C2Ws [N,3,4] f32 (, H, W of the LAST record, as the reference returns them)."""
    with open(path, "r") as f:
        lines = f.readlines()
    Ks, C2Ws = [], []
    height = width = 0
    for i in range(0, len(lines) - 6, 7):
        item = lines[i:i + 7]
        fx, fy, cx, cy = (float(x) for x in re.split(r"\s+", item[1].strip()))
        width, height, _near, _far = (float(x) for x in re.split(r"\s+", item[2].strip()))
        rows = [[float(x) for x in re.split(r"\s+", item[r].strip())] for r in (3, 4, 5)]
        Ks.append(np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float32))
        C2Ws.append(np.array(rows, dtype=np.float32))
    Ks, C2Ws = np.stack(Ks, 0), np.stack(C2Ws, 0)
    return (Ks, C2Ws, int(height), int(width)) if return_shape else (Ks, C2Ws)


def write_campara(path, ks, c2ws, H, W):
    """The reference's writer, digit for digit (focal lengths %.2f, principal point as Python prints it, poses %.8f,
    near/far fixed to 0 / 1000)."""
    with open(path, "w") as f:
        for count, (k, c2w) in enumerate(zip(ks, c2ws)):
            f.write(f"{count}\n")
            f.write(f"{k[0, 0]:.2f} {k[1, 1]:.2f} {k[0, 2]} {k[1, 2]}\n")
            f.write(f"{W} {H} 0 1000\n")
            for r in range(3):
                f.write(f"{c2w[r, 0]:.8f} {c2w[r, 1]:.8f} {c2w[r, 2]:.8f} {c2w[r, 3]:.8f}\n")
            f.write("0 0 0 1\n")


# ---- tile tables ---------------------------------------------------------------------------------------------------
TILE_INFO_HEADER = "# TILEID(1) BBOX_CORNER(3) BBOX_SIZE(3) RESOLUTION(2) FLAG(1)\n"


def write_tile_info(path, corners, sizes, resolutions, flags=None):
    corners, sizes = np.asarray(corners, np.float64), np.asarray(sizes, np.float64)
    if sizes.ndim == 1:
        sizes = np.broadcast_to(sizes, corners.shape)
    resolutions = np.broadcast_to(np.asarray(resolutions), (corners.shape[0], 2))
    flags = np.zeros(corners.shape[0], np.int64) if flags is None else np.asarray(flags)
    with open(path, "w") as f:
        f.write(TILE_INFO_HEADER)
        for i in range(corners.shape[0]):
            c, s = corners[i], sizes[i]
            f.write(f"{i} {c[0]:.2f} {c[1]:.2f} {c[2]:.2f} {s[0]:.2f} {s[1]:.2f} {s[2]:.2f} "
                    f"{int(resolutions[i][0])} {int(resolutions[i][1])} {int(flags[i])}\n")


def read_tile_info(path, tile_idx=None):
    """-> list of dicts {idx, corner [3], size [3], resolution [base, finest], init_outside}, or the one with
    idx == tile_idx (what tile.py:102-110 extracts)."""
    with open(path, "r") as f:
        lines = [ln.strip().split(" ") for ln in f.readlines()[1:] if ln.strip()]
    tiles = [{"idx": int(l[0]), "corner": [float(l[1]), float(l[2]), float(l[3])],
              "size": [float(l[4]), float(l[5]), float(l[6])], "resolution": [int(l[7]), int(l[8])],
              "init_outside": int(l[9]) == 1} for l in lines]
    if tile_idx is None:
        return tiles
    for t in tiles:
        if t["idx"] == tile_idx:
            return t
    raise KeyError(f"tile {tile_idx} not in {path}")


def write_training_views(path, views_per_tile):
    """views_per_tile: list (tile id = position) of lists of global camera ids."""
    with open(path, "w") as f:
        for i, views in enumerate(views_per_tile):
            f.write(f"{i}\n")
            f.write(" ".join(str(int(v)) for v in views) + "\n")


def read_training_views(path, tile_idx=None):
    with open(path, "r") as f:
        lines = f.readlines()
    out = {}
    for i in range(0, len(lines) - 1, 2):
        out[int(lines[i].strip())] = [int(x) for x in lines[i + 1].strip().split(" ") if x]
    return out if tile_idx is None else out[tile_idx]


# ---- per-tile exports ----------------------------------------------------------------------------------------------
def write_cams(path, c2ws, ks, idxs):
    np.savez(path, c2ws=np.asarray(c2ws), ks=np.asarray(ks), idxs=np.asarray(idxs))


def read_cams(path):
    f = np.load(path)
    return f["c2ws"], f["ks"], f["idxs"]


# ---- checkpoints ---------------------------------------------------------------------------------------------------
def export_check_point(path, model, consensus, dec_opt, global_step, grid_opt_state=None):
    """checkpoint-<step>-<tile>.pt with the reference's keys.  `hashgrid` and `admm` hold numpy arrays, `decoder`
    the ShallowMLP state dict (reference parameter names), `optimizer` the decoder optimiser's state dict.  The
    table is stepped by the fused sparse Adam kernel, whose state (exp_avg, exp_avg_sq, step) is stored under
    `featureGrid_optimizer` in torch.optim.Adam's state-dict layout (one parameter, id 0)."""
    n = lambda t: t.detach().cpu().numpy()
    ckp = {"global_step": int(global_step),
           "hashgrid": {"occupied_grid": n(model.occupied_grid), "sampler_log2dim": n(model.log2dim),
                        "grid_resolution": n(2 ** model.log2dim).astype(np.int32), "features": n(model.features)},
           "decoder": {k: v.detach().cpu().clone() for k, v in model.decoder.ref_state_dict().items()},
           "optimizer": dec_opt.state_dict() if dec_opt is not None else None}
    if consensus is not None:
        ckp["admm"] = {"shared_se3": n(consensus.shared_se3), "delta_se3": n(consensus.delta_se3),
                       "overlap_flags": n(consensus.overlap_flags), "rho": n(consensus.rho)}
    ckp["featureGrid_optimizer"] = grid_opt_state if grid_opt_state is not None else {
        "state": {0: {"step": torch.tensor(float(model.adam_step)), "exp_avg": model.exp_avg.detach().cpu().clone(),
                      "exp_avg_sq": model.exp_avg_sq.detach().cpu().clone()}},
        "param_groups": [{"lr": getattr(model, "table_lr", 1e-2), "betas": (0.9, 0.99), "eps": 1e-15, "params": [0]}]}
    torch.save(ckp, path)
    return path


def load_check_point(path, model, consensus=None, dec_opt=None):
    """Inverse of export_check_point; also accepts a checkpoint written by the reference (dense torch Adam on the
    table: its exp_avg / exp_avg_sq / step seed the sparse Adam state).  Returns global_step."""
    ckp = torch.load(path, map_location="cpu", weights_only=False)
    dev = model.device
    hg = ckp["hashgrid"]
    with torch.no_grad():
        model.features.copy_(torch.as_tensor(hg["features"]).to(dev))
        if hasattr(model, "invalidate_gather_table"):
            model.invalidate_gather_table()
    model.log2dim = torch.as_tensor(hg["sampler_log2dim"]).int().to(dev)
    model.occupied_grid = torch.as_tensor(hg["occupied_grid"]).to(dev, torch.bool).contiguous()
    model._occ_full = bool(model.occupied_grid.all())
    model.decoder.load_ref_state_dict(ckp["decoder"])
    st = ckp.get("featureGrid_optimizer", {}).get("state", {})
    if 0 in st:
        model.exp_avg.copy_(st[0]["exp_avg"].to(dev))
        model.exp_avg_sq.copy_(st[0]["exp_avg_sq"].to(dev))
        model.adam_step = int(float(st[0]["step"]))
    if dec_opt is not None and ckp.get("optimizer") is not None:
        dec_opt.load_state_dict(ckp["optimizer"])
    if consensus is not None and "admm" in ckp:
        a = ckp["admm"]
        consensus.shared_se3 = torch.as_tensor(a["shared_se3"]).to(dev)
        consensus.delta_se3 = torch.as_tensor(a["delta_se3"]).to(dev)
        consensus.overlap_flags = torch.as_tensor(a["overlap_flags"]).to(dev)
        consensus.rho = torch.as_tensor(a["rho"]).to(dev)
    return int(ckp["global_step"])


# ---- PLY meshes (input of voxelize_mesh) ---------------------------------------------------------------------------
_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4",
              "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4",
              "float32": "f4", "float64": "f8"}


def read_ply(path):
    """Triangle mesh from an ascii or binary_little_endian PLY -> (vertices [V,3] f32, faces [F,3] i32).
    Reads what plyIO.h's read_plyFile reads: vertex x/y/z and the face list property (vertex_indices /
    vertex_index); other vertex properties are skipped, faces must be triangles."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elems = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").strip().split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elems.append({"name": tok[1], "count": int(tok[2]), "props": []})
            elif tok[0] == "property":
                elems[-1]["props"].append(("list", tok[2], tok[3], tok[4]) if tok[1] == "list" else ("scalar", tok[1], tok[2]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        verts = faces = None
        if fmt == "ascii":
            rows = f.read().decode("ascii", "replace").split("\n")
            pos = 0
        for e in elems:
            scalar_only = all(p[0] == "scalar" for p in e["props"])
            if fmt == "binary_little_endian":
                if scalar_only:
                    dt = np.dtype([(p[2], "<" + _PLY_TYPES[p[1]]) for p in e["props"]])
                    data = np.frombuffer(f.read(dt.itemsize * e["count"]), dtype=dt, count=e["count"])
                    if e["name"] == "vertex":
                        verts = np.stack([data["x"], data["y"], data["z"]], 1).astype(np.float32)
                else:
                    out = []
                    for _ in range(e["count"]):
                        idx = None
                        for p in e["props"]:
                            if p[0] == "list":
                                n = int(np.frombuffer(f.read(np.dtype(_PLY_TYPES[p[1]]).itemsize), "<" + _PLY_TYPES[p[1]])[0])
                                it = np.dtype("<" + _PLY_TYPES[p[2]])
                                v = np.frombuffer(f.read(it.itemsize * n), it)
                                if p[3] in ("vertex_indices", "vertex_index"):
                                    idx = v
                            else:
                                f.read(np.dtype(_PLY_TYPES[p[1]]).itemsize)
                        out.append(idx)
                    if e["name"] == "face":
                        faces = out
            else:
                chunk = [r.split() for r in rows[pos:pos + e["count"]]]
                pos += e["count"]
                if e["name"] == "vertex":
                    names = [p[2] for p in e["props"]]
                    ix = [names.index(c) for c in "xyz"]
                    verts = np.array([[float(r[i]) for i in ix] for r in chunk], np.float32).reshape(-1, 3)
                elif e["name"] == "face":
                    faces = []
                    for r in chunk:
                        col, idx = 0, None
                        for p in e["props"]:
                            if p[0] == "list":
                                n = int(r[col])
                                if p[3] in ("vertex_indices", "vertex_index"):
                                    idx = np.array(r[col + 1:col + 1 + n], np.int64)
                                col += 1 + n
                            else:
                                col += 1
                        faces.append(idx)
    if verts is None or faces is None:
        raise ValueError(f"{path}: needs a vertex and a face element")
    if any(fc is None or len(fc) != 3 for fc in faces):
        raise ValueError(f"{path}: only triangle faces are supported")
    return verts, np.asarray(faces, np.int32).reshape(-1, 3)


def write_ply(path, vertices, faces, binary=True):
    vertices, faces = np.asarray(vertices, np.float32), np.asarray(faces, np.int32)
    hdr = ("ply\nformat %s 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
           "element face %d\nproperty list uchar int vertex_indices\nend_header\n") % (
               "binary_little_endian" if binary else "ascii", len(vertices), len(faces))
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        if binary:
            f.write(vertices.astype("<f4").tobytes())
            rec = np.zeros(len(faces), dtype=[("n", "u1"), ("i", "<i4", (3,))])
            rec["n"], rec["i"] = 3, faces
            f.write(rec.tobytes())
        else:
            for v in vertices:
                f.write(("%r %r %r\n" % (float(v[0]), float(v[1]), float(v[2]))).encode("ascii"))
            for t in faces:
                f.write(("3 %d %d %d\n" % (t[0], t[1], t[2])).encode("ascii"))


def tile_dir(logdir, tile_idx):
    return os.path.join(logdir, f"tile-{tile_idx}")
