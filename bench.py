#!/usr/bin/env python3
"""bench.py -- training rays/s of the per-tile volume-rendering hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--path fused|ops] [--rays B]

One "step" = one full training iteration of one tile on a synthetic ray batch
(BASELINE.json configs[1]: L=16, T=2^19 fp32 table, 2-hidden x 64 decoder, 65 536 rays x
128 samples): occupancy-grid sampling -> hash encode -> decoder -> compositing -> MSE +
0.01*l2_reg_specular -> backward -> fused sparse Adam on the table + Adam on the decoder.
Inputs (rays, targets, parameters) are resident in HBM before the timed region.

N > 1: one process per GPU (torchrun), one independent tile per rank (tiles shard one per GPU:
admm_trainer.py:74-83) -> weak scaling, no data-path collective; the ADMM camera-consensus
exchange (RCCL all-reduce) runs every SYN_ITERS=100 steps and is also timed on its own.

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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--path", default=os.environ.get("SCANERF_BENCH_PATH", "auto"), choices=["auto", "fused", "ops"])
    ap.add_argument("--rays", type=int, default=65536)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--log2-T", type=int, default=19, help="hash-table entries per level (configs[1]: 19; the reference's default.yaml: 24)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tiles-per-gpu", type=int, default=1,
                    help="tiles resident on each GPU and stepped round-robin (configs[4]: 32 tiles on 8 GPUs = 4; the reference "
                         "swaps them through host memory, tile.py:574-636 -- 288 GB of HBM keeps them resident)")
    ap.add_argument("--scatter", default="auto", choices=["auto", "fused", "dfeat"],
                    help="table-gradient records emitted by the backward kernel (fused, default) or by the stand-alone "
                         "binned scatter from a level-major dfeat (tuning comparison)")
    ap.add_argument("--workload", default="configs1", choices=["configs1", "configs2", "configs4-render"],
                    help="configs1 (default, the metric's configuration): fp32 table, fully occupied sampler grid; "
                         "configs2: + sphere-shell occupancy at log2dim 7 and a bf16 table (fp32 master + fused sparse Adam); "
                         "configs4-render: 4 tiles per GPU + background, one 1920x1080 novel view per step (render rays/s)")
    return ap.parse_args()


def cpu_baseline(samples, seconds_budget=15.0):
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

    def step(i):
        z, dd = O.sample_points_grid(o, d, tile.occ_corner, tile.occ_size, tile.occ, tile.log2dim, samples)
        valid = torch.from_numpy((z != -1).all(1))
        out = O.render_batch_rays(torch.from_numpy(o)[valid], torch.from_numpy(d)[valid], torch.from_numpy(z)[valid],
                                  torch.from_numpy(dd)[valid], feats, tile.res, sd, O.TRAIN,
                                  lambda x: O.contract_fore(x, tile.min_bbox, tile.bbox_size), 1000 + i)
        loss = torch.nn.functional.mse_loss(out["rgb"], tgt[valid]) + 0.01 * out["l2_reg_specular"]
        feats.grad = None
        loss.backward()
        p = feats.detach().numpy().reshape(-1, 8)
        O.adam_step(p, feats.grad.numpy().reshape(-1, 8), m, v, 1e-2, 0.9, 0.99, 1e-15, i)

    step(0)
    t0 = time.time()
    n = 0
    while n < 2 or (time.time() - t0 < seconds_budget and n < 50):
        step(n + 1)
        n += 1
    dt = (time.time() - t0) / n
    return {"value": B / dt, "unit": "rays/s", "cores": cores, "kind": "port", "psnr_vs_oracle_db": psnr_vs_oracle(samples),
            "sample": f"{n} training iterations of {B} rays x {samples} samples (same tile config, T=2^19), oracle/ on {cores} host threads"}


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
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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

    with torch.no_grad():  # rays that meet no occupied cell are skipped by every kernel: they are not counted as work
        valid_frac = float((model.sample(rays_o, rays_d, S)[0] != -1).all(1).float().mean())
    if rank == 0:
        value = world * B * valid_frac * args.steps / elapsed
        line = {
            "metric": "training rays/s per GPU (128 samples, L=16 hash)",
            "value": value, "value_is": "whole-job aggregate over n_gpus (one tile per GPU)", "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"configs[2]: as configs[1] + sphere-shell occupancy (r=3 m, 0.5 m thick, log2dim 7, "
                                    f"{float(model.occupied_grid.float().mean()):.3f} of cells), bf16 gather table (fp32 master, fp32 accumulate), "
                                    f"fused sparse Adam; {B} rays x {S} samples" if occ else
                                    f"configs[1]: single 8m^3 tile per GPU, L=16 T=2^{args.log2_T} fp32 hash grid, 2-hidden x 64 "
                                    f"decoder, {B} rays x {S} samples, full training iteration "
                                    f"(sample+encode+decode+composite fwd, bwd, sparse Adam); foreground branch"),
                       "path": path, "rays_per_step": B, "valid_ray_fraction": valid_frac, "samples": S, "tiles_per_gpu": ntile,
                       "parallelism": f"tile-per-gpu x{world}", "syn_iters": SYN_ITERS},
            "consensus_ms": consensus_ms,
            "consensus_frac_of_iteration": consensus_ms / (SYN_ITERS * ms_per_step),
        }
        if timer and timer.count:
            name, avg_ms, alg_bytes, alg_flops = timer.dominant()
            from scanerf_amd import render as _render
            h3 = _render.ARITH == 1 and path == "fused"
            # matrix-pipe floor of the launch: f32-input MFMA, or 3 f16 MFMAs per term for the split arithmetic
            mfma_peak = MFMA_F16_PEAK_TFLOPS if h3 else MFMA_F32_PEAK_TFLOPS
            t_hbm, t_mfma = alg_bytes / (HBM_PEAK_GBS * 1e9), (3 if h3 else 1) * alg_flops / (mfma_peak * 1e12)
            line["config"]["decoder_arith"] = "split f16 x3 MFMA, f32 accumulate (csrc/render_h3.h)" if h3 else "f32 MFMA"
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
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
                if h3 and name in pm and args.workload == "configs1" and B == 65536 and S == 128 and args.log2_T == 19:
                    traffic = pm[name]["fetch_bytes"] + pm[name]["write_bytes"]
                    roof["traffic_source"] = "profiles/r01_pmc_traffic.json (FETCH_SIZE + WRITE_SIZE of " + pm[name]["kernel"] + ")"
            except (OSError, ValueError, KeyError):
                pass
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
