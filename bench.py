#!/usr/bin/env python3
"""bench.py -- training rays/s of the per-tile volume-rendering hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--path fused|ops] [--rays B]

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
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="launcher / collective rehearsal on gloo + CPU tensors: no kernels, no measurement")
    ap.add_argument("--tiles-per-gpu", type=int, default=1,
                    help="tiles resident on each GPU and stepped round-robin (configs[4]: 32 tiles on 8 GPUs = 4; the reference "
                         "swaps them through host memory, tile.py:574-636 -- 288 GB of HBM keeps them resident)")
    ap.add_argument("--scatter", default="auto", choices=["auto", "fused", "dfeat"],
                    help="table-gradient records emitted by the backward kernel (fused, default) or by the stand-alone "
                         "binned scatter from a level-major dfeat (tuning comparison)")
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
    """The launcher's and the consensus exchange's rehearsal on gloo: same process-group bring-up, barrier, MAX-reduce of
    the elapsed time and all-reduce consensus as the real run, CPU tensors, no kernels."""
    from scanerf_amd import consensus as cons
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    n_cam_per, overlap = 120, 24
    n_cam = (n_cam_per - overlap) * world + overlap
    cam_idx = torch.arange(n_cam_per) + rank * (n_cam_per - overlap)
    admm = cons.ConsensusState(n_cam, cam_idx, "cpu")
    se3 = torch.randn(n_cam_per, 6) * 1e-3
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        admm.exchange(se3)
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "training rays/s per GPU (128 samples, L=16 hash)", "value": 0.0, "unit": "rays/s",
                          "n_gpus": world, "rccl_world_size": dist.get_world_size() if world > 1 else 1, "dry_run": True,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t.item()) / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none (dry run)",
                          "data": "synthetic", "config": {"workload": "launcher + gloo consensus rehearsal, no kernels"}}))
    if world > 1:
        dist.destroy_process_group()


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
    B = 256
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
                      f"same way) of {B} rays x {samples} samples (same tile config, T=2^19), oracle/ on {cores} host threads"}


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


def bench_render(args, world, rank, dev):
    """configs[4] render leg: 4 tiles per GPU (admm_trainer.py:74-83 round-robin), background shells, one 1920x1080 view
    per step through the multi-tile renderer (rendering.py:286-544 counterpart).  value = rendered rays (pixels) per second."""
    import tempfile

    from scanerf_amd import renderer as R
    from scanerf_amd import tile_model as tm
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

    for _ in range(args.warmup):
        rend.render(H, W, K, c2w, num_sample=args.samples, num_bg_sample=args.samples)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = rend.render(H, W, K, c2w, num_sample=args.samples, num_bg_sample=args.samples)
    sync()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "novel-view render rays/s per GPU (128 samples, L=16 hash)", "value": world * H * W * args.steps / elapsed,
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (f16 tables)", "data": "synthetic",
            "config": {"workload": f"configs[4] render leg: {ntile} tiles per GPU (f16 tables T=2^{args.log2_T}, shell occupancy) + "
                                   f"blended backgrounds, one {W}x{H} view per step, {args.samples} fg + {args.samples} bg samples",
                       "opaque_fraction": float((out[3] < 0.5).float().mean())}}))
    if world > 1:
        dist.destroy_process_group()


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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
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
        step_fn = functools.partial(tm.train_step_fused, fused_scatter=args.scatter == "fused")
    timer = tm.KernelTimer() if hasattr(tm, "KernelTimer") else None

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

    # the same step with the exact-f32 decoder arithmetic (f32-input MFMA), timed the same way, printed beside the headline
    from scanerf_amd import render as _render
    h3_run = path == "fused" and _render.ARITH != 0
    dtype_label = ("f32 tables/compositing; decoder GEMMs split-f16 x3 MFMA, f32 accumulate (22-bit operands); backward gradient "
                   "products f16 MFMA + 13-bit table-gradient records, summed in 64-bit fixed point (gradient error vs oracle 7e-4 "
                   "rel. L2; h3_grad_ms_per_step = the same step with 22-bit gradient products and f32 records, 1e-5)"
                   if h3_run and _render.ARITH == 2 else
                   "f32 tables/accumulate/compositing; decoder GEMMs split-f16 x3 MFMA, f32 accumulate (22-bit operands)"
                   if h3_run else "f32")
    if occ:
        dtype_label = "bf16 gather table, fp32 master + accumulate; " + dtype_label
    f32_ms = h3_ms = None
    if h3_run and not occ and not fgbg:
        for other in ("f32", "h3"):
            _render.set_arith(other)
            try:
                for i in range(2):
                    step_fn(models[i % ntile], dec_opts[i % ntile], rays_o, rays_d, target, S, step0 + i)
                sync()
                f0 = time.perf_counter()
                for i in range(args.steps):
                    step_fn(models[i % ntile], dec_opts[i % ntile], rays_o, rays_d, target, S, step0 + i)
                sync()
                ms = (time.perf_counter() - f0) / args.steps * 1e3
                if other == "f32":
                    f32_ms = ms
                else:
                    h3_ms = ms
            finally:
                _render.set_arith(_render.DEFAULT_ARITH)

    with torch.no_grad():  # rays that meet no occupied cell are skipped by every kernel: they are not counted as work
        valid_frac = float((model.sample(rays_o, rays_d, S)[0] != -1).all(1).float().mean())
    if rank == 0:
        value = world * B * valid_frac * args.steps / elapsed
        line = {
            "metric": "training rays/s per GPU (128 samples, L=16 hash)",
            "value": value, "value_is": "whole-job aggregate over n_gpus (one tile per GPU)", "value_per_gpu": value / world,
            "unit": "rays/s", "n_gpus": world, "rccl_world_size": dist.get_world_size() if world > 1 else 1,
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
                       "path": path, "rays_per_step": B, "valid_ray_fraction": valid_frac, "samples": S, "tiles_per_gpu": ntile,
                       "parallelism": f"tile-per-gpu x{world}", "syn_iters": SYN_ITERS},
            "pose_grads": bool(args.pose_grads), "f32_arith_ms_per_step": f32_ms, "h3_grad_ms_per_step": h3_ms,
            "consensus_ms": consensus_ms,
            "consensus_frac_of_iteration": consensus_ms / (SYN_ITERS * ms_per_step),
        }
        if timer and timer.count:
            name, avg_ms, alg_bytes, alg_flops = timer.dominant()
            from scanerf_amd import render as _render
            h3 = _render.ARITH != 0 and path == "fused"
            # matrix-pipe floor of the launch: f32-input MFMA, or 3 f16 MFMAs per term for the split arithmetic
            mfma_peak = MFMA_F16_PEAK_TFLOPS if h3 else MFMA_F32_PEAK_TFLOPS
            t_hbm, t_mfma = alg_bytes / (HBM_PEAK_GBS * 1e9), (3 if h3 else 1) * alg_flops / (mfma_peak * 1e12)
            line["config"]["decoder_arith"] = ("f32 MFMA" if not h3 else "split f16 x3 MFMA, f32 accumulate (csrc/render_h3.h)" if _render.ARITH == 1
                                               else "forward + backward recompute: split f16 x3 MFMA, f32 accumulate; gradient products: f16 MFMA, f32 "
                                                    "accumulate, power-of-two scaled (csrc/render_t16.h); table-gradient records 8 bytes: 13-bit "
                                                    "significands under a per-record exponent (csrc/scatter_common.h)")
            if t_mfma > t_hbm:  # the kernel's floor is set by the matrix pipe, not by HBM
                ach = alg_flops / (avg_ms * 1e-3) / 1e12
                roof = {"bound": "mfma", "achieved": ach, "peak": mfma_peak, "unit": "TFLOP/s",
                        "frac": ach / mfma_peak, "algorithmic_flops_per_launch": alg_flops}
            else:
                ach = alg_bytes / (avg_ms * 1e-3) / 1e9
                roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
            # HBM-side bytes of that kernel per launch: rocprofv3 PMC passes of this same command, committed under profiles/
            traffic = None
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")))
                if h3 and name in pm and args.workload == "configs1" and B == 65536 and S == 128 and args.log2_T == 19:
                    traffic = pm[name]["fetch_bytes"] + pm[name]["write_bytes"]
                    roof["traffic_source"] = "profiles/r02_pmc_traffic.json (FETCH_SIZE + WRITE_SIZE of " + pm[name]["kernel"] + ")"
            except (OSError, ValueError, KeyError):
                pass
            # SURVEY.md 8(d) bytes of the WHOLE step (forward + backward per ray) against the step time
            whole = B * valid_frac * (BYTES_FWD_PER_RAY + BYTES_BWD_PER_RAY) * (2 if fgbg else 1) if S == 128 else None
            if whole:
                roof["whole_step"] = {"bytes": whole, "ms": ms_per_step, "frac": whole / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * 1e9)}
            roof.update({"kernel": name, "traffic": traffic, "avg_launch_ms": avg_ms,
                         "algorithmic_bytes_per_launch": alg_bytes, "all_kernels_ms": timer.summary()})
            line["roofline"] = roof
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(S)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
