#!/usr/bin/env python3
"""bench.py -- training rays/s of the per-tile volume-rendering hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--path fused|ops] [--rays B] [--arith f32|h3|t16|t16s]

`value` / `ms_per_step` are measured under the fastest F32-EQUIVALENT decoder arithmetic (render.FP32_EQUIV_ARITH; gradient
error against the oracle ~5e-6 relative L2, reported live as `arith_evidence`); the reduced-precision "t16" step and the
other arithmetics are timed beside it under their own keys.

One "step" = one full training iteration of one tile on a synthetic ray batch
(BASELINE.json configs[1]: L=16, T=2^19 fp32 table, 2-hidden x 64 decoder, 65 536 rays x
128 samples): occupancy-grid sampling -> hash encode -> decoder -> compositing -> MSE +
0.01*l2_reg_specular -> backward -> fused sparse Adam on the table + Adam on the decoder.
Inputs (rays, targets, parameters) are resident in HBM before the timed region.

N > 1: one process per GPU, one independent tile per rank (tiles shard one per GPU:
admm_trainer.py:74-83) -> weak scaling, no data-path collective; the ADMM camera-consensus
exchange (RCCL all-reduce) runs every SYN_ITERS=100 steps and is also timed on its own.
Started under torchrun (RANK/WORLD_SIZE in the environment) this process IS one rank.  Started
plainly with --gpus N > 1 it is the launcher: like the reference's admm_trainer.py:312-337 it spawns
its own workers -- N child processes of this script, one per GPU, with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set -- BEFORE anything touches the GPU, relays rank 0's JSON line and exits
non-zero if any rank failed.  --dry-run-cpu runs the same launcher and collectives on gloo/CPU with
no kernels (the CPU tests use it; its line says "dry_run": true and is not a measurement).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# Requests per second the L2s exchange with the fabric (Infinity Cache / HBM side) -- MEASURED on this chip, not a spec: scattered
# 8-byte gathers beyond the L2s 55-59 G/s (tools/gather_bench.hip), 12-byte record appends 54-65 G write requests/s whatever their mix of
# 32- and 64-byte requests (tools/probe/record_store_bench.hip), the training forward 55 G/s.  DESIGN.md 4.11.
FABRIC_REQ_CEILING_GPS = 57.0
MFMA_F32_PEAK_TFLOPS = 157.3  # fp32-input MFMA = fp32 vector peak (MI355X_MICROARCH.md, Matrix cores)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA (MI355X_MICROARCH.md); the h3 arithmetic issues 3 f16 products per term
BYTES_FWD_PER_RAY = 24 + 20 + 128 * 16 * 8 * 2 * 4        # 131 116 B (fp32, S=128, L=16)
BYTES_BWD_PER_RAY = 128 * 16 * (8 + 64 + 16 * 8)          # 409 600 B
SYN_ITERS = 100                                           # config/default.yaml:5


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--path", default=os.environ.get("SCANERF_BENCH_PATH", "auto"), choices=["auto", "fused", "ops"])
    ap.add_argument("--rays", type=int, default=65536)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--log2-T", type=int, default=19, help="hash-table entries per level (configs[1]: 19; the reference's default.yaml: 24)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-legs", action="store_true",
                    help="skip the two extra timings of the default run (ops_path_ms_per_step, render_ms_per_frame)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="launcher / collective rehearsal on gloo + CPU tensors: no kernels, no measurement")
    ap.add_argument("--tiles-per-gpu", type=int, default=1,
                    help="tiles resident on each GPU and stepped round-robin (configs[4]: 32 tiles on 8 GPUs = 4; the reference "
                         "swaps them through host memory, tile.py:574-636 -- 288 GB of HBM keeps them resident)")
    ap.add_argument("--scatter", default="auto", choices=["auto", "fused", "dfeat"],
                    help="table-gradient records emitted by the backward kernel (fused, default) or by the stand-alone "
                         "binned scatter from a level-major dfeat (tuning comparison)")
    ap.add_argument("--arith", default=None, choices=["f32", "h3", "t16", "t16s"],
                    help="decoder arithmetic of the timed step (default: the fastest f32-equivalent one, render.FP32_EQUIV_ARITH; "
                         "\"t16\" is the reduced-precision option and is reported under its own name, never as the metric)")
    ap.add_argument("--infer-arith", default="t16", choices=["t16", "h3", "f32"],
                    help="decoder arithmetic of the render-time inference ops (configs4-render; hashgrid.lib.HASHGRID.INFER_ARITH): 16-sample "
                         "tiles (default), round 4's 32-sample-tile kernel, or the single-pass f32 kernel")
    ap.add_argument("--arith-side-off", action="store_true", help="skip the side timings of the other arithmetics (profiling passes)")
    ap.add_argument("--pose-grads", action="store_true",
                    help="with --workload configs1 / configs1-fgbg: the iteration also returns dL/d(rays_o), dL/d(rays_d) (the "
                         "reference's default: CAMOPT.ENABLE)")
    ap.add_argument("--workload", default="configs1", choices=["configs1", "configs1-fgbg", "configs2", "configs4-render"],
                    help="configs1 (default, the metric's configuration): fp32 table, fully occupied sampler grid; "
                         "configs1-fgbg: the reference's complete iteration (tile.py:639-692, config/default.yaml:15-18): foreground + "
                         "T_left * background branch, --samples fg + --samples bg samples per ray, one merged loss, one Adam step; "
                         "configs2: + sphere-shell occupancy at log2dim 7 and a bf16 table (fp32 master + fused sparse Adam); "
                         "configs4-render: 4 tiles per GPU + background, one 1920x1080 novel view per step (render rays/s)")
    return ap.parse_args()


def launch_ranks(args):
    """--gpus N without a torchrun environment: this process only launches.  It never calls into torch.cuda (a process
    that has initialised the GPU must not start others that share it by exec; children are plain subprocesses)."""
    import socket
    import subprocess
    if any(k.startswith(("ROCPROFILER_", "ROCPROF_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        # under rocprofv3 the preloaded profiler has initialised the GPU in THIS process already: starting ranks from it is the
        # exec-after-GPU-init the pool forbids.  Profile one rank (--gpus 1), or profile under torchrun's own ranks.
        print("bench.py: refusing to launch ranks from a profiled process (profile with --gpus 1)", file=sys.stderr)
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    line = None
    for ln in (out0 or "").splitlines():
        if ln.startswith("{"):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if any(codes) or line is None:
        print(f"bench.py: ranks exited with {codes}" + ("" if line else "; rank 0 printed no JSON line"), file=sys.stderr)
        sys.exit(next((c for c in codes if c), 1))
    print(line)


def dry_run_cpu(args, world, rank):
    """The launcher's and BOTH exchanges' rehearsal on gloo: same process-group bring-up, barrier and MAX-reduce of the elapsed
    time as the real run, CPU tensors, no kernels.  What runs is the real multi-tile driver (admm.AdmmDriver: consensus
    all-reduce + shared-depth MIN exchange once per stretch on every rank) over configs[4]'s tile -> rank map with UNEQUAL tile
    counts (4 * world + 1 tiles round-robin, admm_trainer.py:74-83: rank 0 owns one tile more), with stand-in tile trainers."""
    import types

    from scanerf_amd import admm
    from scanerf_amd import consensus as cons
    from scanerf_amd import occlusion
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    n_tiles = 4 * world + 1
    n_cam_per, overlap = 120, 24
    n_cam = (n_cam_per - overlap) * n_tiles + overlap
    mine = admm.tiles_of_rank(n_tiles, rank, world)

    class StandIn:   # a TileTrainer without kernels: the interface AdmmDriver drives
        def __init__(self, t):
            g = torch.Generator().manual_seed(t)
            idx = torch.arange(n_cam_per) + t * (n_cam_per - overlap)
            self.tile, self.cam_idx = t, idx
            self.cameras = types.SimpleNamespace(se3_refine=torch.nn.Parameter(torch.randn(n_cam_per, 6, generator=g) * 1e-3))
            self.consensus = cons.ConsensusState(n_cam, idx, "cpu")
            self.admm, self.steps = False, 0

        def maybe_prune(self):
            pass

        def train_one_step(self):
            self.steps += 1

    trainers = [StandIn(t) for t in mine]
    depth = torch.full((n_cam, 4, 6), occlusion.NO_DEPTH)   # half-resolution depth maps, one publisher per camera

    def publish(tr):   # every tile publishes the maps of the cameras it owns outright (the first n_cam_per - overlap)
        cams = tr.cam_idx[: n_cam_per - overlap].tolist()
        for c in cams:
            depth[c] = float(tr.tile + 1)
        return cams

    syn = max(1, min(SYN_ITERS, args.steps))
    drv = admm.AdmmDriver(trainers, total_step=args.steps, syn_iters=syn, depth_hooks=(publish, lambda tr: None, depth))
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    hist = drv.run()
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # every rank must have received every tile's maps, and stepped its own tiles only
    owned = torch.zeros(n_cam, dtype=torch.bool)
    for tt in range(n_tiles):
        owned[torch.arange(n_cam_per - overlap) + tt * (n_cam_per - overlap)] = True
    ok = bool((depth[owned] != occlusion.NO_DEPTH).all()) and all(tr.steps == sum(drv.stretches) for tr in trainers)
    if rank == 0:
        print(json.dumps({"metric": "training rays/s per GPU (128 samples, L=16 hash)", "value": 0.0, "unit": "rays/s",
                          "n_gpus": world, "rccl_world_size": dist.get_world_size() if world > 1 else 1, "dry_run": True,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t.item()) / sum(drv.stretches) * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none (dry run)",
                          "data": "synthetic",
                          "config": {"workload": "launcher + gloo rehearsal of the multi-tile driver: consensus all-reduce and shared-depth MIN "
                                                 "exchange per stretch, no kernels",
                                     "tiles": n_tiles, "tiles_of_rank0": len(mine), "exchanges": len(hist), "exchange_ok": ok}}))
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)


def cpu_baseline(samples):
    """The oracle (CPU port of the same training iteration) on this box's host cores, on a bounded
    sample of the workload.  Only this leg of bench.py touches oracle/."""
    import numpy as np

    from oracle import oracle as O

    # the GPU box gives one GPU a 16-core share of the host: more threads only oversubscribe
    cores = min(os.cpu_count() or 1, 16)
    os.environ["OMP_NUM_THREADS"] = str(cores)  # the C oracle's OpenMP loops (set before it is loaded)
    torch.set_num_threads(cores)
    rng = np.random.default_rng(0)
    B = 2048   # 1/32 of the step's 65 536 rays, same samples / table / decoder: ~1 s of host work per iteration
    tile = O.Tile([-4, -4, -4], [8, 8, 8], log2_T=19)
    o = rng.uniform(-4, 4, (B, 3)).astype(np.float32)
    d = rng.normal(size=(B, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    feats = torch.randn(16, tile.T, 2) * 1e-3
    feats.requires_grad_(True)
    sd = {k: v.requires_grad_(True) for k, v in O.init_mlp(0).items()}
    tgt = torch.rand(B, 3)
    m = np.zeros((16 * tile.T * 2 // 8, 8), np.float32)
    v = np.zeros_like(m)

    def forward(i, grad):
        z, dd = O.sample_points_grid(o, d, tile.occ_corner, tile.occ_size, tile.occ, tile.log2dim, samples)
        valid = torch.from_numpy((z != -1).all(1))
        with torch.set_grad_enabled(grad):
            out = O.render_batch_rays(torch.from_numpy(o)[valid], torch.from_numpy(d)[valid], torch.from_numpy(z)[valid],
                                      torch.from_numpy(dd)[valid], feats, tile.res, sd, O.TRAIN if grad else O.INFERENCE,
                                      lambda x: O.contract_fore(x, tile.min_bbox, tile.bbox_size), 1000 + i)
        return out, valid

    def step(i):
        out, valid = forward(i, True)
        loss = torch.nn.functional.mse_loss(out["rgb"], tgt[valid]) + 0.01 * out["l2_reg_specular"]
        feats.grad = None
        loss.backward()
        p = feats.detach().numpy().reshape(-1, 8)
        O.adam_step(p, feats.grad.numpy().reshape(-1, 8), m, v, 1e-2, 0.9, 0.99, 1e-15, i)

    # 3 warm-up + 10 timed iterations, forward and forward+backward(+sparse Adam) separately
    WARM, TIMED = 3, 10
    for i in range(WARM):
        step(i)
    t0 = time.time()
    for i in range(TIMED):
        step(WARM + i)
    dt = (time.time() - t0) / TIMED
    for i in range(WARM):
        forward(i, False)
    t0 = time.time()
    for i in range(TIMED):
        forward(i, False)
    dt_fwd = (time.time() - t0) / TIMED
    return {"value": B / dt, "unit": "rays/s", "cores": cores, "kind": "port", "forward_only_rays_per_s": B / dt_fwd,
            "psnr_vs_oracle_db": psnr_vs_oracle(samples),
            "sample": f"{WARM} warm-up + {TIMED} timed training iterations (forward + backward + sparse Adam; forward-only timed the "
                      f"same way) of {B} rays x {samples} samples = 1/{65536 // B} of the step's rays (same tile config, T=2^19), "
                      f"oracle/ on {cores} host threads; a reported baseline, not a target"}


def psnr_vs_oracle(samples, B=2048, log2_T=15):
    """PSNR (tools/utils.py:53-55: 10 log10(255^2 / (mse + 1e-8)) on 0..255 values) of the HIP render of B rays against the
    oracle's render of the same rays, table and decoder -- the metric's "PSNR vs ref" on the reference-equivalent CPU path
    (the reference ships no dataset and no CUDA build runs here).  Part of the cpu_baseline leg: the oracle is the checker."""
    import numpy as np

    import scanerf_amd  # noqa: F401
    from oracle import oracle as O
    from scanerf_amd import network, render
    from scanerf_amd.cuda import sample_points_grid
    dev = "cuda:0"
    rng = np.random.default_rng(1)
    tile = O.Tile([-4, -4, -4], [8, 8, 8], log2_T=log2_T)
    o = rng.uniform(-4, 4, (B, 3)).astype(np.float32)
    d = rng.normal(size=(B, 3)).astype(np.float32)
    feat = (rng.normal(size=(16, tile.T, 2)) * 0.5).astype(np.float32)
    sd = O.init_mlp(seed=2)
    t = lambda a: torch.as_tensor(a).to(dev).contiguous()
    z = torch.full((B, samples), -1.0, device=dev)
    dist_ = torch.full((B, samples), -1.0, device=dev)
    sample_points_grid(t(o), t(d), z, dist_, t(tile.occ_corner), t(tile.occ_size), t(tile.occ), t(tile.log2dim))
    valid = torch.all(z != -1, dim=-1)
    pk = render.PackedDecoder(dev).pack(O.pack_blob(sd).to(dev), network.weight_feature(20000, dev))
    out, _ = render.render_forward(t(o), t(d), z, dist_, t(feat), t(tile.res), pk, tile.min_bbox.tolist(), tile.bbox_size.tolist(),
                                   render.FORE, False, ray_valid=valid, want_weights=False)
    v = valid.cpu()
    with torch.no_grad():
        ref = O.render_batch_rays(torch.from_numpy(o)[v], torch.from_numpy(d)[v], z.cpu()[v], dist_.cpu()[v], torch.from_numpy(feat),
                                  tile.res, sd, O.INFERENCE, lambda x: O.contract_fore(x, tile.min_bbox, tile.bbox_size), 20000)
    mse = float(((out[valid][:, 0:3].cpu() * 255.0 - ref["rgb"] * 255.0) ** 2).mean())
    return 10.0 * float(np.log10(255.0 ** 2 / (mse + 1e-8)))


def time_render(args, world, rank, dev, steps, warmup):
    """configs[4] render leg: 4 tiles per GPU (admm_trainer.py:74-83 round-robin), background shells, one 1920x1080 view
    per step through the multi-tile renderer (rendering.py:286-544 counterpart).  -> (seconds for `steps` frames, MAX over ranks;
    H, W, ntile, fraction of opaque pixels)."""
    import tempfile

    from scanerf_amd import renderer as R
    from scanerf_amd import tile_model as tm
    from scanerf_amd.hashgrid.lib import HASHGRID as _HG
    _HG.INFER_ARITH = getattr(args, "infer_arith", "t16")
    H, W, ntile = 1080, 1920, (args.tiles_per_gpu if args.tiles_per_gpu > 1 else 4)
    tiles = []
    with tempfile.TemporaryDirectory() as tmp:
        for t in range(ntile):
            m = tm.TileModel([-4.0 * ntile + 8.0 * t, -4, -4], [8, 8, 8], dev, log2_T=args.log2_T, seed=rank * ntile + t, sampler_log2dim=7)
            m.set_occupancy(tm.sphere_shell_occupancy(m, 3.0, 0.5))
            with torch.no_grad():
                m.features.mul_(300.0)  # xavier std of a 2^19-entry table gives sigma ~ softplus(0): make the shell visible
            R.export_tile(os.path.join(tmp, f"tile{t}"), m)
            tiles.append(R.load_tile(os.path.join(tmp, f"tile{t}")))
            del m
    rend = R.TileSetRenderer(dev, tiles)
    K = [1600.0, 0, W / 2, 0, 1600.0, H / 2, 0, 0, 1]
    c2w = torch.tensor([[1.0, 0, 0, 0.0], [0, 1, 0, 0.5], [0, 0, 1, -14.0]])

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        rend.render(H, W, K, c2w, num_sample=args.samples, num_bg_sample=args.samples)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = rend.render(H, W, K, c2w, num_sample=args.samples, num_bg_sample=args.samples)
    sync()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()), H, W, ntile, float((out[3] < 0.5).float().mean())


def bench_render(args, world, rank, dev):
    """`--workload configs4-render`: value = rendered rays (pixels) per second."""
    elapsed, H, W, ntile, opaque = time_render(args, world, rank, dev, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps({
            "metric": "novel-view render rays/s per GPU (128 samples, L=16 hash)", "value": world * H * W * args.steps / elapsed,
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (f16 tables)", "data": "synthetic",
            "config": {"workload": f"configs[4] render leg: {ntile} tiles per GPU (f16 tables T=2^{args.log2_T}, shell occupancy) + "
                                   f"blended backgrounds, one {W}x{H} view per step, {args.samples} fg + {args.samples} bg samples",
                       "opaque_fraction": opaque}}))
    if dist.is_initialized():
        dist.destroy_process_group()


def autograd_route_leg(args, dev, rays_o, rays_d, target, S, step0, n=3):
    """autograd_route_ms_per_step: the same iteration through the reference-shaped classes with the caller's OWN loss code --
    hashgrid.HashGrid.render_fore_rays (-> render.FusedRenderRays: the fused forward / backward kernels behind torch autograd),
    a torch MSE + l2_reg loss, loss.backward(), torch.optim.Adam on the network.ShallowMLP parameters, adam_step_cuda on the
    table's dense gradient -- what tile.py:880-1015 runs when it keeps its loss terms (depth / smooth / ADMM penalty)."""
    from scanerf_amd import network
    from scanerf_amd.cuda import adam_step_cuda
    from scanerf_amd.hashgrid import HashGrid
    hg = HashGrid(dev, torch.tensor([-4.0, -4, -4]), torch.tensor([8.0, 8, 8]), log2_hashmap_size=args.log2_T, grid_resolution=[32, 2048],
                  sampler_log2dim=4)
    dec = network.init_model(network.ShallowMLP(32), "xavier").to(dev)
    opt = torch.optim.Adam(dec.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    m1, m2 = torch.zeros_like(hg.HE.features), torch.zeros_like(hg.HE.features)
    K = hg.HE.features.numel() // 8

    def step(i):
        hg.HE.features.grad = None
        opt.zero_grad(set_to_none=True)
        o, ok = hg.render_fore_rays(rays_o, rays_d, S, dec, 0, global_step=step0 + i)
        loss = torch.nn.functional.mse_loss(o["pred_color"], target) + 0.01 * o["l2_reg_specular"]
        loss.backward()
        with torch.no_grad():
            adam_step_cuda(hg.HE.features.data.view(K, 8), hg.HE.features.grad.view(K, 8), m1.view(K, 8), m2.view(K, 8), 1e-2, 0.9, 0.99,
                           1e-15, i)
        opt.step()
    step(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(1 + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    assert hg.last_render_route == "fused"
    del hg, dec, opt, m1, m2
    torch.cuda.empty_cache()
    return {"autograd_route_ms_per_step": ms,
            "autograd_route_is": (f"hashgrid.HashGrid.render_fore_rays (render.FusedRenderRays) + torch loss + loss.backward() + torch Adam on "
                                  f"network.ShallowMLP + adam_step_cuda on the dense table gradient, {rays_o.shape[0]} rays x {S} samples, 1 warm-up + "
                                  f"{n} timed steps")}


def decoder_op_leg(dev, N, n=5):
    """decoder_op: the stand-alone decoder (csrc/decoder.hip, what network.ShallowMLP runs) on N samples, HIP-event timed: the one
    kernel pair of the path that is MATRIX-bound rather than gather- or stream-bound.  flops = SURVEY.md 8(d): 27 456 per sample
    forward (x3 MFMA products for the hi/lo split), peak = 2.5 PFLOP/s dense f16."""
    from scanerf_amd import decoder_op, network
    torch.manual_seed(3)
    x = torch.cat([0.3 * torch.randn(N, 32, device=dev), torch.randn(N, 3, device=dev)], -1).requires_grad_(True)
    blob = network.xavier_blob(1, dev, bias_scale=0.05).requires_grad_(True)
    wf = network.weight_feature(20000, dev)
    g = [torch.randn(N, c, device=dev) for c in (1, 3, 3, 3)]
    ev = lambda: torch.cuda.Event(enable_timing=True)
    tf = tb = 0.0
    for i in range(n + 1):
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        outs = decoder_op.decoder_apply(x, blob, wf)
        e1.record()
        torch.autograd.backward(outs, g)
        e2.record()
        torch.cuda.synchronize()
        if i:   # (first pass = warm-up)
            tf += e0.elapsed_time(e1)
            tb += e1.elapsed_time(e2)
        x.grad = blob.grad = None
    tf, tb = tf / n, tb / n
    flops = N * 2 * 13728
    return {"decoder_op": {"samples": N, "forward_ms": tf, "backward_ms": tb, "forward_useful_TFLOPs": flops / tf / 1e9,
                           "forward_mfma_TFLOPs": 3 * flops / tf / 1e9, "forward_mfma_frac_of_peak": 3 * flops / tf / 1e9 / 2500.0,
                           "peak_TFLOPs": 2500.0, "is": "scanerf_decoder_forward / _backward (incl. decoder packing and the "
                           "op's torch glue) on [N,35] inputs; split-f16 operands: three f16 MFMA products per useful product"}}


def side_legs(args, dev, rays_o, rays_d, target, S, step0):
    """Two more timings inside the default run, so that they are on the driver's clock too (rank 0, N = 1, configs[1] only;
    `--no-side-legs` skips them):
      ops_path_ms_per_step -- the same training iteration through the binding-surface ops one by one (`--path ops`: sampler,
        `embedding_bg_forward/backward_cuda` via the autograd wrapper, the decoder op of csrc/decoder.hip (what network.ShallowMLP
        runs), torch compositing, `adam_step_cuda`): the route a caller takes that keeps `tile.py` / `hashgrid/__init__.py` unchanged;
      render_ms_per_frame -- configs[4]'s render leg (4 resident tiles + backgrounds, one 1920x1080 view)."""
    from scanerf_amd import tile_model as tm
    out = {}
    m = tm.TileModel([-4.0, -4, -4], [8, 8, 8], dev, log2_T=args.log2_T, seed=17, sampler_log2dim=4)
    opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    tm.train_step_ops(m, opt, rays_o, rays_d, target, S, step0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_ops = 3
    for i in range(n_ops):
        tm.train_step_ops(m, opt, rays_o, rays_d, target, S, step0 + 1 + i)
    torch.cuda.synchronize()
    out["ops_path_ms_per_step"] = (time.perf_counter() - t0) / n_ops * 1e3
    out["ops_path_is"] = (f"`--path ops`: the binding-surface ops called one by one -- sampler, encoder op through its autograd wrapper, the "
                          f"decoder as ONE HIP op each way (csrc/decoder.hip), torch compositing + autograd, adam_step_cuda -- "
                          f"{rays_o.shape[0]} rays x {S} samples, 1 warm-up + {n_ops} timed steps (round 4, torch decoder graph: 155 ms)")
    del m, opt
    torch.cuda.empty_cache()
    out.update(autograd_route_leg(args, dev, rays_o, rays_d, target, S, step0))
    out.update(decoder_op_leg(dev, rays_o.shape[0] * S))
    torch.cuda.empty_cache()
    n_fr = 5
    elapsed, H, W, ntile, opaque = time_render(args, 1, 0, dev, n_fr, 2)
    out["render_ms_per_frame"] = elapsed / n_fr * 1e3
    out["render_is"] = (f"configs[4] render leg: {ntile} resident tiles (f16 tables T=2^{args.log2_T}, shell occupancy) + blended backgrounds, "
                        f"one {W}x{H} view, {args.samples} + {args.samples} samples, 2 warm-up + {n_fr} timed frames; opaque fraction {opaque:.3f}")
    torch.cuda.empty_cache()
    out.update(reference_default_leg(args, dev, S, step0))
    torch.cuda.empty_cache()
    return out


def reference_default_leg(args, dev, S, step0, n=8):
    """The iteration of the reference's SHIPPED configuration (config/default.yaml:2,15-18; tile.py:301,1010): T = 2^24 entries
    per level, 16 384 rays, foreground + T_left * background (128 + 128 samples), ray gradients for the pose refinement, one
    sparse Adam step over both branches' table gradients -- `--workload configs1-fgbg --log2-T 24 --rays 16384 --pose-grads` as a
    side leg of the default line (2 warm-up + n timed iterations; per-section HIP-event times beside the wall clock)."""
    from scanerf_amd import tile_model as tm
    Bd = 16384
    g = torch.Generator(device=dev).manual_seed(24)
    m = tm.TileModel([-4.0, -4, -4], [8, 8, 8], dev, log2_T=24, seed=24, sampler_log2dim=4)
    opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    ro = torch.rand(Bd, 3, device=dev, generator=g) * 8 - 4
    rd = torch.nn.functional.normalize(torch.randn(Bd, 3, device=dev, generator=g), dim=-1) * (0.5 + torch.rand(Bd, 1, device=dev, generator=g))
    tg = torch.rand(Bd, 3, device=dev, generator=g)
    for i in range(2):
        tm.train_step_fgbg(m, opt, ro, rd, tg, S, S, step0 + i, pose_grads=True)
    torch.cuda.synchronize()
    timer = tm.KernelTimer()
    t0 = time.perf_counter()
    for i in range(n):
        tm.train_step_fgbg(m, opt, ro, rd, tg, S, S, step0 + 2 + i, pose_grads=True, timer=timer)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    secs = {k: round(v, 4) for k, v in timer.summary().items()}   # (average per LAUNCH of the section: two forwards / backwards per iteration)
    del m, opt
    torch.cuda.empty_cache()
    # OPT-IN variant, under its own key: the table's Adam moments in half precision (adam_step_cuda_fp16 semantics,
    # cuda/adam_kernel.cu:98-144) -- not what the reference's live code runs (torch.optim.Adam, fp32 state), never the default
    ms16 = None
    try:
        m = tm.TileModel([-4.0, -4, -4], [8, 8, 8], dev, log2_T=24, seed=24, sampler_log2dim=4, fp16_moments=True)
        opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
        for i in range(2):
            tm.train_step_fgbg(m, opt, ro, rd, tg, S, S, step0 + i, pose_grads=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            tm.train_step_fgbg(m, opt, ro, rd, tg, S, S, step0 + 2 + i, pose_grads=True)
        torch.cuda.synchronize()
        ms16 = (time.perf_counter() - t0) / n * 1e3
        del m, opt
    except Exception as e:  # noqa: BLE001
        print(f"bench.py: fp16-moment leg failed: {e}", file=sys.stderr)
    return {"reference_default_ms_per_iteration": ms, "reference_default_rays_per_s": Bd / ms * 1e3,
            "reference_default_sections_ms": secs, "reference_default_fp16_moments_ms_per_iteration": ms16,
            "reference_default_is": (f"the reference's shipped configuration: T=2^24 fp32 table (fp32 Adam moments), {Bd} rays x ({S} + {S}) "
                                     f"samples, fg + T_left*bg, pose (ray) gradients, one sparse Adam; 2 warm-up + {n} timed iterations")}


# rocprof kernel names of the timer's sections, per arithmetic (profiles/r06_kernel_stats.txt lists them with their durations)
KERNEL_OF = {
    "render_forward": lambda ar: "k_render_fwd<0>" if ar == "f32" else ("k_render_fwd_h3<0, true, false>" if ar in ("t16", "t16s") else "k_render_fwd_h3<0, false, false>"),
    "render_backward": lambda ar: {"f32": "k_render_bwd<0>", "h3": "k_render_bwd_h3<0>", "t16": "k_render_bwd_t16<0, 1, false, false>",
                                   "t16s": "k_render_bwd_t16<0, 2, false, true>"}[ar],
    "table_grad_accumulate_adam": lambda ar: {"t16": "k_bin_accumulate<512, 32, true, true>", "t16s": "k_bin_accumulate<512, 16, true, true>"}.get(ar, "k_bin_accumulate<256, 32, true, true>"),
}
PMC_FILES = ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json")   # the newest committed counter file wins


def load_pmc():
    for name in PMC_FILES:
        try:
            return json.load(open(os.path.join(ROOT, "profiles", name))), "profiles/" + name
        except (OSError, ValueError):
            continue
    return {}, None


def design_bytes(section, rays, S, arith, T):
    """What THIS design moves per launch, by construction: the denominator of `amplification`.
    forward: the table gathers (8(d)'s S*L*8*F*4) + rays/z/dists in + x-stash and tile_T out; backward: x-stash, z/dists, per-ray
    rows in + the scatter records out (4 per (sample, level), 12 / 8 / 16 bytes by arithmetic; bucket straddles add < 1 %);
    accumulate: the records in + the touched entries of table and both Adam moments read and written (bounded by the table)."""
    rec = {"t16s": 12, "t16": 8}.get(arith, 16)
    if section == "render_forward":
        return {"table_gathers": rays * S * 16 * 8 * 2 * 4, "rays_z_dists_in": rays * (24 + 8 * S),
                "xstash_out": rays * S * 32 * 4, "tile_T_out_ray_out": rays * (4 * ((S + 15) // 16) + 64)}
    if section == "render_backward":
        return {"xstash_in": rays * S * 32 * 4, "rays_z_dists_tile_T_rows_in": rays * (24 + 8 * S + 4 * ((S + 15) // 16) + 128),
                "records_out": rays * S * 16 * 4 * rec}
    if section == "table_grad_accumulate_adam":
        return {"records_in": rays * S * 16 * 4 * rec, "table_and_moments_rmw_upper_bound": 16 * T * 2 * 4 * 3 * 2}
    return None


def fabric_ceiling_live(dev):
    """The chip's L2 <-> fabric request rate for scattered accesses, measured NOW: 2^26 8-byte entries (512 MB, 16x the L2s),
    4 x 256 workgroups x 512 threads x 256 scattered loads each; loads / second ~ requests / second (one L2 miss per lane-load up
    to the 6 % that hit).  scanerf_gather_rate_probe is measurement infrastructure of the library (include/scanerf_hip.h)."""
    import ctypes

    from scanerf_amd import _capi
    entries, blocks, per = 1 << 26, 1024, 256
    table = torch.zeros(entries, 2, dtype=torch.float32, device=dev)
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    call = lambda: _capi.check(_capi.lib().scanerf_gather_rate_probe(
        ctypes.c_void_p(table.data_ptr()), ctypes.c_longlong(entries), ctypes.c_int(blocks), ctypes.c_int(per),
        ctypes.c_void_p(sink.data_ptr()), _capi.stream()), "gather_rate_probe")
    call()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    del table
    return blocks * 512 * per / (best * 1e-3) / 1e9


def roofline(timer, arith, args, B, S, valid_frac, fgbg, ms_per_step):
    """`roofline` of the JSON line.  Live fields (HIP events of this very run): `avg_launch_ms`, `achieved`, `frac`.  Per kernel of
    the step: SURVEY.md 8(d)'s algorithmic bytes (strictly: bytes_fwd = 24 + 20 + S*L*8*F*4 per ray for the forward; for the
    backward the reference ALGORITHM's S*L*(8 + 64 + 16*8) = 409 600 B per ray -- grad-in read, feature re-gather, 16 RMW atomics
    -- which this design does not perform: reported because 8(d) defines it), `design_bytes` (what this design moves by
    construction) and, from the COMMITTED rocprofv3 passes (`counters_source`; fields prefixed `rocprof_` / named `traffic*`,
    `frac_counter`, `mfma_busy`, `wait_any`, `valu_active`, `lds_bank_conflict` are read from that file, NOT measured in this
    run): `traffic` = FETCH_SIZE + WRITE_SIZE per launch with the guide's x2 FETCH correction applied where the kernel's reads
    are 16-B-per-lane streams (`fetch_x2` in the file), `amplification` = traffic / design bytes, `frac_at_rocprof_avg` = the 8(d)
    fraction at the profile's average duration.  The top-level fields describe the dominant kernel."""
    rays = B * valid_frac
    secs = timer.summary()
    pmc, pmc_src = load_pmc()
    same_cfg = args.workload == "configs1" and B == 65536 and S == 128 and args.log2_T == 19 and not args.pose_grads
    alg = {"render_forward": rays * BYTES_FWD_PER_RAY * S / 128.0, "embedding_bg_forward": rays * BYTES_FWD_PER_RAY * S / 128.0,
           "render_backward": rays * BYTES_BWD_PER_RAY * S / 128.0, "embedding_bg_backward": rays * BYTES_BWD_PER_RAY * S / 128.0}
    kernels = {}
    for name, ms in secs.items():
        k = {"avg_launch_ms": ms}
        if name in KERNEL_OF:
            k["kernel"] = KERNEL_OF[name](arith)
        if name in alg:
            k["algorithmic_bytes"] = alg[name]
            k["achieved_GBps"] = alg[name] / (ms * 1e-3) / 1e9
            k["frac"] = k["achieved_GBps"] / HBM_PEAK_GBS
        db = design_bytes(name, rays, S, arith, 1 << args.log2_T)
        if db:
            k["design_bytes"] = dict(db, total=float(sum(db.values())))
        c = pmc.get("kernels", {}).get(k.get("kernel", ""), None) if same_cfg else None
        if c and "fetch_bytes" in c:
            k["traffic"] = c.get("traffic_bytes", c["fetch_bytes"] * (2 if c.get("fetch_x2") else 1) + c["write_bytes"])
            k["traffic_fetch_x2_applied"] = bool(c.get("fetch_x2"))
            k["rocprof_avg_us"] = c.get("avg_us")
            k["frac_counter"] = k["traffic"] / (c["avg_us"] * 1e-6) / (HBM_PEAK_GBS * 1e9) if c.get("avg_us") else None
            if db:
                k["amplification"] = k["traffic"] / k["design_bytes"]["total"]
            if name in alg and c.get("avg_us"):
                k["frac_at_rocprof_avg"] = alg[name] / (c["avg_us"] * 1e-6) / (HBM_PEAK_GBS * 1e9)
            for key in ("mfma_busy", "wait_any", "valu_active", "lds_bank_conflict", "fabric_requests", "fabric_request_rate_Gps"):
                if key in c:
                    k[key] = c[key]
            if "fabric_request_rate_Gps" in c:
                k["frac_fabric_request_ceiling"] = c["fabric_request_rate_Gps"] / FABRIC_REQ_CEILING_GPS
        kernels[name] = k
    with_alg = [n for n in kernels if "algorithmic_bytes" in kernels[n]]
    # (a path whose sections carry no 8(d) byte count -- none today -- still gets a line: its slowest section, time only)
    name = max(with_alg or kernels, key=lambda n: kernels[n]["avg_launch_ms"])
    d = kernels[name]
    roof = {"bound": "hbm", "achieved": d.get("achieved_GBps"), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d.get("frac"),
            "traffic": d.get("traffic"), "kernel": d.get("kernel", name), "section": name, "avg_launch_ms": d["avg_launch_ms"],
            "algorithmic_bytes_per_launch": d.get("algorithmic_bytes"),
            "algorithmic_bytes_is": ("SURVEY.md 8(d) bytes_bwd: the REFERENCE algorithm's backward traffic (re-gather + 16 RMW atomics per "
                                     "(sample, level)); this design replaces it by the x-stash + records: see design_bytes / traffic / frac_counter")
            if name in ("render_backward", "embedding_bg_backward") else "SURVEY.md 8(d) bytes_fwd",
            "live_fields": "avg_launch_ms, achieved, frac, kernels.*.avg_launch_ms / achieved_GBps / frac (HIP events of this run)",
            "frac_at_rocprof_avg": d.get("frac_at_rocprof_avg"), "rocprof_avg_us": d.get("rocprof_avg_us"),
            "design_bytes_per_launch": (d.get("design_bytes") or {}).get("total"), "amplification": d.get("amplification"),
            "frac_counter": d.get("frac_counter"), "mfma_busy": d.get("mfma_busy"), "wait_any": d.get("wait_any"),
            "counters_source": (pmc_src + " (committed rocprofv3 --pmc / --stats passes of this command, tools/profile_bench.sh; NOT measured in "
                                "this run)") if "traffic" in d else None,
            "kernels": kernels}
    req = [kernels.get(n, {}).get("fabric_requests") for n in ("render_forward", "render_backward", "table_grad_accumulate_adam")]
    if all(r is not None for r in req):
        # the roofline that binds this design: every kernel of the step lives on requests to the fabric; at the
        # measured ceiling the step's requests alone take floor_ms
        roof["fabric_requests"] = {"per_step": float(sum(req)), "ceiling_Gps": FABRIC_REQ_CEILING_GPS, "floor_ms": sum(req) / (FABRIC_REQ_CEILING_GPS * 1e9) * 1e3,
                                   "step_ms": ms_per_step, "frac": sum(req) / (FABRIC_REQ_CEILING_GPS * 1e9) * 1e3 / ms_per_step,
                                   "source": "request counts: " + (pmc_src or "") + " (committed counters); ceiling: measured micro-benchmarks, not a spec"}
    if S == 128:
        whole = rays * (BYTES_FWD_PER_RAY + BYTES_BWD_PER_RAY) * (2 if fgbg else 1)
        roof["whole_step"] = {"bytes": whole, "ms": ms_per_step, "frac": whole / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * 1e9),
                              "note": "8(d) bytes of forward + backward per ray / step time; the backward part is the reference algorithm's"}
    return roof


def arith_evidence(samples, B=16384, log2_T=19):
    """Why the headline's arithmetic counts as f32-equivalent: on B rays the fused step's table and decoder gradients under every
    arithmetic against autograd through the oracle (torch f32 on the CPU), as relative L2 errors; and the render's outputs against
    the oracle's.  B = 16 384 rays on the step's own table size (T = 2^19): thousands of records meet in one coarse entry, as in
    the timed step (~10 s of host work, ~16 GB of host memory for the oracle's autograd graph).  Part of the cpu_baseline leg
    (rank 0, N = 1): the oracle is the checker here."""
    import numpy as np

    import scanerf_amd  # noqa: F401
    from oracle import oracle as O
    from scanerf_amd import network, render
    dev = "cuda:0"
    rng = np.random.default_rng(3)
    T = 2 ** log2_T
    o = rng.uniform(-3.5, 3.5, (B, 3)).astype(np.float32)
    d = rng.normal(size=(B, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    z = np.sort(rng.uniform(0.2, 3.0, (B, samples)).astype(np.float32), axis=1)
    dist_ = np.concatenate([z[:, 1:] - z[:, :-1], np.full((B, 1), 0.03, np.float32)], 1)
    feat = (rng.normal(size=(16, T, 2)) * 0.4).astype(np.float32)
    sd = {k: v.clone().requires_grad_(True) for k, v in O.init_mlp(seed=7, bias_scale=0.05).items()}
    res = O.level_resolutions(torch.tensor([32, 32, 32]), torch.tensor([2048, 2048, 2048]))
    mn, sz = torch.tensor([-8.0, -8.0, -8.0]), torch.tensor([16.0, 16.0, 16.0])
    F = torch.from_numpy(feat).requires_grad_(True)
    step = 20000
    tgt = torch.from_numpy(rng.random((B, 3)).astype(np.float32))
    # The oracle's f32 autograd in chunks of 1 024 rays, gradients summed over the chunks in float64: torch's CPU reductions over all
    # 2.1e6 samples at once lose ~3e-4 of the decoder gradient to their own f32 running sums (every arithmetic, the exact-f32 kernels
    # included, then shows the same 2.8e-4 against it); per chunk that error is ~1e-5 and the 16 partial sums add in float64.
    CH = 1024
    gb64, gF64, rgb_ref = torch.zeros(O.pack_blob(sd).numel(), dtype=torch.float64), torch.zeros(F.shape, dtype=torch.float64), []
    for c0 in range(0, B, CH):
        sl = slice(c0, min(c0 + CH, B))
        n = sl.stop - sl.start
        ref = O.render_batch_rays(torch.from_numpy(o[sl]), torch.from_numpy(d[sl]), torch.from_numpy(z[sl]), torch.from_numpy(dist_[sl]), F, res, sd,
                                  O.TRAIN, lambda x: O.contract_fore(x, mn, sz), step)
        # mean over all B rays = sum over chunks of (chunk mean) * n / B
        ((torch.nn.functional.mse_loss(ref["rgb"], tgt[sl]) + 0.01 * ref["l2_reg_specular"]) * (n / B)).backward()
        gb64 += O.pack_blob({k: v.grad for k, v in sd.items()}).double()
        gF64 += F.grad.double()
        rgb_ref.append(ref["rgb"].detach())
        F.grad = None
        for v in sd.values():
            v.grad = None
    ref = {"rgb": torch.cat(rgb_ref)}
    gb_ref, gF_ref = gb64.numpy(), gF64.numpy()
    t = lambda a: torch.as_tensor(a).to(dev).contiguous()
    blob = O.pack_blob({k: v.detach() for k, v in sd.items()}).to(dev)
    wf = network.weight_feature(step, dev)
    pk = render.PackedDecoder(dev).pack(blob, wf)
    box = (mn.tolist(), sz.tolist(), render.FORE, False)
    pts = O.contract_fore((torch.from_numpy(o)[:, None, :] + torch.from_numpy(z)[..., None] * torch.from_numpy(d)[:, None, :]).reshape(-1, 3), mn, sz).numpy()
    out = {}
    keep = render.ARITH
    try:
        for ar in render.ARITH_NAMES:
            render.set_arith(ar)
            tile_T = torch.empty(B, render.tile_T_columns(samples), device=dev)
            xs = torch.empty(B * samples, 32, device=dev)
            o_r, _ = render.render_forward(t(o), t(d), t(z), t(dist_), t(feat), t(res.numpy()), pk, *box, tile_T=tile_T, xstash=xs)
            loss, gout = render.photometric_loss_grad(o_r, t(tgt.numpy()), None, 0.01)
            dfeat, gblob = render.render_backward(t(o), t(d), t(z), t(dist_), t(feat), t(res.numpy()), pk, wf, *box, o_r, tile_T, gout, xstash=xs)
            gF = render.scatter_table_grad(t(pts), dfeat, torch.zeros(16, T, 2, device=dev), t(res.numpy())).cpu().numpy()
            l2 = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
            out[ar] = {"table_grad_rel_l2_vs_oracle": l2(gF, gF_ref), "decoder_grad_rel_l2_vs_oracle": l2(gblob.cpu().numpy(), gb_ref),
                       "rgb_max_abs_err_vs_oracle": float((o_r[:, 0:3].cpu() - ref["rgb"].detach()).abs().max())}
    finally:
        render.ARITH = keep
    out["sample"] = (f"{B} rays x {samples} samples, T=2^{log2_T}, oracle = torch f32 autograd on the host in chunks of {CH} rays, chunk gradients "
                     "summed in float64 (its own rounding ~1e-6)")
    return out


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when its first communicator comes up; the contract is ONE JSON line there.  File
    descriptor 1 points at stderr while the process group is initialised and its first collective has run."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def init_rccl(world, local):
    """The `nccl` (= RCCL) process group of this rank: rendezvous from the launcher's environment for N > 1, an in-process
    store for one rank (no child process) -- so that the consensus exchange runs through RCCL's all-reduce on every box."""
    with stdout_to_stderr():
        dev = torch.device("cuda", local)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("nccl", store=dist.HashStore(), rank=0, world_size=1, device_id=dev)
        t = torch.zeros(8, device=dev)
        dist.all_reduce(t)       # brings the communicator up (and its banner out) here
        torch.cuda.synchronize()


def library_state():
    from scanerf_amd import _capi
    st = _capi.audit_state()
    lib = _capi.lib()
    return {"isa_audit": st.get("status"), "compiler": st.get("compiler"), "abi": int(lib.scanerf_abi_version()),
            "experiments_build": bool(lib.scanerf_experiments_enabled()), "so_bytes": os.path.getsize(_capi.LIB_PATH)}


def rccl_mapped():
    """librccl is mapped into this process (what `torch.distributed`'s nccl backend is on ROCm)."""
    try:
        return any("librccl" in ln for ln in open("/proc/self/maps"))
    except OSError:
        return False


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args)  # the parent never touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running {world} rank(s)", file=sys.stderr)
    if args.dry_run_cpu:
        return dry_run_cpu(args, world, rank)
    torch.cuda.set_device(local)
    rccl_error = None
    if world > 1:
        init_rccl(world, local)
    else:
        try:
            init_rccl(1, local)
        except Exception as e:  # noqa: BLE001  (reported in the line as rccl_loaded: false + the reason)
            rccl_error = f"{type(e).__name__}: {e}"
            if dist.is_initialized():   # a group whose first all-reduce failed must not stay up: every later exchange would re-enter it
                dist.destroy_process_group()
    dev = f"cuda:{local}"

    import scanerf_amd  # noqa: F401  (fails loudly if the HIP library is missing)
    from scanerf_amd import consensus as cons
    from scanerf_amd import tile_model as tm

    B, S = args.rays, args.samples
    torch.manual_seed(rank)
    if args.workload == "configs4-render":
        return bench_render(args, world, rank, dev)
    occ = args.workload == "configs2"
    ntile = max(1, args.tiles_per_gpu)
    models, dec_opts = [], []
    for t in range(ntile):  # the rank's tiles share one footprint (the batch lies in it); weights and state are their own
        m = tm.TileModel([-4.0 + 8.0 * rank, -4, -4], [8, 8, 8], dev, log2_T=args.log2_T, seed=rank * ntile + t,
                         sampler_log2dim=7 if occ else 4, table_dtype=torch.bfloat16 if occ else torch.float32)
        if occ:  # shell of radius 3 m, thickness 0.5 m around the tile centre
            m.set_occupancy(tm.sphere_shell_occupancy(m, 3.0, 0.5))
        models.append(m)
        dec_opts.append(torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15))
    model, dec_opt = models[0], dec_opts[0]
    corner = torch.tensor([-4.0 + 8.0 * rank, -4, -4], device=dev)
    rays_o = torch.rand(B, 3, device=dev) * 8 + corner
    rays_d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1) * (0.5 + torch.rand(B, 1, device=dev))
    target = torch.rand(B, 3, device=dev)

    path = args.path
    if path == "auto":
        path = "fused" if hasattr(tm, "train_step_fused") else "ops"
    step_fn = tm.train_step_fused if path == "fused" else tm.train_step_ops
    if path == "fused" and args.pose_grads:
        import functools
        step_fn = functools.partial(tm.train_step_fused, pose_grads=True)
    fgbg = args.workload == "configs1-fgbg"
    if fgbg:
        path = "fused"
        step_fn = lambda m_, o_, ro, rd, tg, S_, st, timer=None: tm.train_step_fgbg(m_, o_, ro, rd, tg, S_, S_, st, timer=timer,
                                                                                    pose_grads=args.pose_grads)
    if path == "fused" and args.scatter != "auto":
        import functools
        if fgbg:
            print("bench.py: --scatter applies to the foreground-only step (configs1 / configs2), not to configs1-fgbg", file=sys.stderr)
            sys.exit(2)
        step_fn = functools.partial(tm.train_step_fused, fused_scatter=args.scatter == "fused", pose_grads=bool(args.pose_grads))
    timer = tm.KernelTimer() if hasattr(tm, "KernelTimer") else None
    # decoder arithmetic of the timed step: the fastest f32-equivalent one unless --arith says otherwise
    from scanerf_amd import render as _render
    arith = args.arith or _render.FP32_EQUIV_ARITH
    if path == "fused":
        _render.set_arith(arith)

    # ADMM consensus state: N_cam cameras, each tile sees M of them, 20 % shared with the next tile
    n_cam_per, overlap = 120, 24
    n_cam = (n_cam_per - overlap) * world + overlap
    cam_idx = torch.arange(n_cam_per, device=dev) + rank * (n_cam_per - overlap)
    admm = cons.ConsensusState(n_cam, cam_idx, dev)
    se3 = torch.randn(n_cam_per, 6, device=dev) * 1e-3

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # steady-state iteration count: past the coarse-to-fine warm-up (hashgrid/__init__.py:228-235) every one
    # of the 16 levels is active -- the first 10 000 iterations mask fine levels and do less useful work
    step0 = 20000
    for i in range(max(args.warmup, ntile if ntile > 1 else 0)):
        step_fn(models[i % ntile], dec_opts[i % ntile], rays_o, rays_d, target, S, step0 + i)
    admm.exchange(se3)
    sync()
    if timer:
        timer.reset()
    t0 = time.perf_counter()
    for i in range(args.steps):
        mi, oi = models[i % ntile], dec_opts[i % ntile]
        step_fn(mi, oi, rays_o, rays_d, target, S, step0 + args.warmup + i, timer=timer) if timer else \
            step_fn(mi, oi, rays_o, rays_d, target, S, step0 + args.warmup + i)
        if (i + 1) % SYN_ITERS == 0:
            admm.exchange(se3)
    sync()
    elapsed = time.perf_counter() - t0
    # consensus exchange timed on its own (it runs once per SYN_ITERS steps)
    sync()
    c0 = time.perf_counter()
    for _ in range(10):
        admm.exchange(se3)
    sync()
    consensus_ms = (time.perf_counter() - c0) / 10 * 1e3

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3

    # the same step under the other arithmetics, timed the same way, printed beside the headline under their own names
    fused = path == "fused"
    dtype_label = _render.ARITH_DTYPE[arith] if fused else "f32 (op-by-op path)"
    if occ:
        dtype_label = "bf16 gather table, fp32 master + accumulate; " + dtype_label
    side_ms = {}
    if fused and not occ and not fgbg and not args.arith and not args.arith_side_off:
        for other in ("t16", "h3", "f32"):
            if other == arith:
                continue
            _render.set_arith(other)
            try:
                for i in range(2):
                    step_fn(models[i % ntile], dec_opts[i % ntile], rays_o, rays_d, target, S, step0 + i)
                sync()
                f0 = time.perf_counter()
                n_side = max(args.steps // 2, 5)
                for i in range(n_side):
                    step_fn(models[i % ntile], dec_opts[i % ntile], rays_o, rays_d, target, S, step0 + i)
                sync()
                side_ms[other] = (time.perf_counter() - f0) / n_side * 1e3
            finally:
                _render.set_arith(arith)

    with torch.no_grad():  # rays that meet no occupied cell are skipped by every kernel: they are not counted as work
        valid_frac = float((model.sample(rays_o, rays_d, S)[0] != -1).all(1).float().mean())
    if rank == 0:
        value = world * B * valid_frac * args.steps / elapsed
        f32_equiv = (not fused) or arith in ("f32", "h3", "t16s")
        line = {
            "metric": "training rays/s per GPU (128 samples, L=16 hash)" if f32_equiv else
                      "training rays/s per GPU (128 samples, L=16 hash) -- REDUCED-PRECISION gradient arithmetic (t16), not the metric",
            "value": value, "value_is": "whole-job aggregate over n_gpus (one tile per GPU)", "value_per_gpu": value / world,
            "unit": "rays/s", "n_gpus": world, "rccl_world_size": dist.get_world_size() if dist.is_initialized() else 1,
            # the consensus exchanges of this run went through a `nccl` process group and librccl is mapped into the process
            "rccl_loaded": bool(dist.is_initialized() and dist.get_backend() == "nccl" and rccl_mapped()),
            "rccl_error": rccl_error,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype_label, "data": "synthetic",
            "config": {"workload": (f"configs[2]: as configs[1] + sphere-shell occupancy (r=3 m, 0.5 m thick, log2dim 7, "
                                    f"{float(model.occupied_grid.float().mean()):.3f} of cells), bf16 gather table (fp32 master, fp32 accumulate), "
                                    f"fused sparse Adam; {B} rays x {S} samples" if occ else
                                    f"configs[1] rays through the reference's COMPLETE iteration (tile.py:639-692): foreground (occupancy-"
                                    f"sampled, contract_fore) + T_left * background (inverse-z, contract_bg, infinity), L=16 T=2^{args.log2_T} fp32 "
                                    f"hash grid, {B} rays x ({S} + {S}) samples, one merged loss, one sparse Adam step" if fgbg else
                                    f"configs[1]: single 8m^3 tile per GPU, L=16 T=2^{args.log2_T} fp32 hash grid, 2-hidden x 64 "
                                    f"decoder, {B} rays x {S} samples, full training iteration "
                                    f"(sample+encode+decode+composite fwd, bwd, sparse Adam); foreground branch"),
                       "path": path, "arith": arith if fused else "ops", "rays_per_step": B, "valid_ray_fraction": valid_frac, "samples": S,
                       "tiles_per_gpu": ntile, "parallelism": f"tile-per-gpu x{world}", "syn_iters": SYN_ITERS},
            "pose_grads": bool(args.pose_grads),
            # the same step under the other arithmetics (ms): t16 = reduced-precision gradient products + 13-bit records;
            # h3 = round 1's 32-sample-tile split kernel; f32 = exact f32-input MFMA
            "t16_ms_per_step": side_ms.get("t16"), "h3_ms_per_step": side_ms.get("h3"), "f32_arith_ms_per_step": side_ms.get("f32"),
            "consensus_ms": consensus_ms,
            "consensus_frac_of_iteration": consensus_ms / (SYN_ITERS * ms_per_step),
            # which library produced these numbers: the ISA audit's verdict on the loaded file (tools/isa_audit.py), its compiler, and
            # whether it is the product build (no environment switches) or an experiments build (make EXP=1)
            "library": library_state(),
        }
        if timer and timer.count:
            line["roofline"] = roofline(timer, arith if fused else "f32", args, B, S, valid_frac, fgbg, ms_per_step)
        if timer and timer.count and world == 1 and "fabric_requests" in line.get("roofline", {}):
            try:   # the ceiling of roofline.fabric_requests, measured live beside the committed micro-benchmarks' 57 G/s
                live = fabric_ceiling_live(dev)
                fr = line["roofline"]["fabric_requests"]
                fr["ceiling_live_Gps"] = live
                fr["frac_at_live_ceiling"] = fr["per_step"] / (live * 1e9) * 1e3 / ms_per_step
            except (RuntimeError, AttributeError) as e:
                print(f"bench.py: fabric ceiling probe skipped: {e}", file=sys.stderr)
        if (world == 1 and fused and args.workload == "configs1" and not args.no_side_legs and not args.arith and not args.pose_grads
                and ntile == 1 and args.scatter == "auto"):
            del models, dec_opts, model, dec_opt
            torch.cuda.empty_cache()
            try:   # (the headline above is already measured: a failing side leg must not discard it)
                line.update(side_legs(args, dev, rays_o, rays_d, target, S, step0))
            except Exception as e:  # noqa: BLE001
                line["side_legs_error"] = f"{type(e).__name__}: {e}"[:400]
                print(f"bench.py: side legs failed: {e}", file=sys.stderr)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(S)
            if fused and not occ and not fgbg:
                line["arith_evidence"] = arith_evidence(S)
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
