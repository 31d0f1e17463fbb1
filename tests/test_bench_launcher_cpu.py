"""bench.py --gpus N launches its own ranks (admm_trainer.py:312-337: the reference spawns one worker per GPU itself).
Rehearsed on gloo/CPU with --dry-run-cpu: same launcher, process-group bring-up, barrier, MAX-reduce and consensus
all-reduce as the GPU run, no kernels."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, env=e,
                          timeout=600)


def test_gpus_2_spawns_two_ranks_and_relays_rank0_line():
    r = _run("--gpus", "2", "--steps", "4", "--dry-run-cpu")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_world_size"] == 2 and line["dry_run"] is True
    assert line["steps"] == 4 and line["ms_per_step"] > 0
    # configs[4]'s tile -> rank map with unequal counts: 9 tiles on 2 ranks (rank 0 owns 5), both exchanges of the real driver
    # (consensus all-reduce, shared-depth MIN) once per stretch plus the initial one, every published depth map received
    cfg = line["config"]
    assert cfg["tiles"] == 9 and cfg["tiles_of_rank0"] == 5 and cfg["exchanges"] == 2 and cfg["exchange_ok"] is True


def test_single_rank_line_unchanged_without_launcher():
    r = _run("--steps", "2", "--dry-run-cpu")
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["rccl_world_size"] == 1


def test_under_torchrun_env_the_process_is_a_rank_not_a_launcher():
    # WORLD_SIZE=1 / RANK=0 in the environment (what torchrun exports): no children are spawned even with --gpus 2
    r = _run("--gpus", "2", "--steps", "2", "--dry-run-cpu", env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1
    assert "WORLD_SIZE=1" in r.stderr


def test_failing_rank_makes_the_launcher_fail():
    # --steps 0: rank 0 divides by zero while formatting its line -> the launcher must exit non-zero and print no JSON
    r = _run("--gpus", "2", "--steps", "0", "--dry-run-cpu")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


class _FakeTimer:   # tile_model.KernelTimer.summary() without a GPU
    def __init__(self, secs):
        self.secs = secs

    def summary(self):
        return dict(self.secs)


def _args(**kw):
    import types
    base = dict(workload="configs1", log2_T=19, pose_grads=False)
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_roofline_of_the_ops_path_sections_gives_a_line():
    # `python bench.py --path ops` records these sections (tile_model.train_step_ops + PyHashGridBG.TIMER); round 3's roofline()
    # took max() over an empty sequence there and printed no JSON line
    sys.path.insert(0, ROOT)
    import bench
    t = _FakeTimer({"forward_total": 60.0, "backward_total": 80.0, "sparse_adam": 0.06, "embedding_bg_forward": 5.5, "embedding_bg_backward": 7.0})
    r = bench.roofline(t, "f32", _args(), 65536, 128, 1.0, False, 150.0)
    assert r["section"] == "embedding_bg_backward" and r["frac"] > 0 and r["traffic"] is None and r["counters_source"] is None
    # no section with an 8(d) byte count at all: the slowest section, time only
    r = bench.roofline(_FakeTimer({"forward_total": 60.0, "backward_total": 80.0}), "f32", _args(), 65536, 128, 1.0, False, 150.0)
    assert r["section"] == "backward_total" and r["frac"] is None and r["achieved"] is None
    json.dumps(r)


def test_roofline_fields_recompute_from_the_committed_counter_file():
    sys.path.insert(0, ROOT)
    import bench
    pmc, src = bench.load_pmc()
    assert src and src.startswith("profiles/")
    secs = {"sample_points_grid": 0.13, "render_forward": 3.0, "render_backward": 5.0, "table_grad_accumulate_adam": 1.5}
    r = bench.roofline(_FakeTimer(secs), "t16s", _args(), 65536, 128, 1.0, False, 9.8)
    assert r["section"] == "render_backward" and r["kernel"] == "k_render_bwd_t16<0, 2, false, true>"
    k = r["kernels"]["render_backward"]
    assert abs(k["frac"] - 65536 * 409600 / 5.0e-3 / 8e12) < 1e-9            # live: 8(d) bytes / live duration / 8 TB/s
    c = pmc["kernels"][k["kernel"]]
    want = c.get("traffic_bytes", c["fetch_bytes"] * (2 if c.get("fetch_x2") else 1) + c["write_bytes"])
    assert k["traffic"] == want and r["traffic"] == want
    assert abs(k["frac_counter"] - want / (c["avg_us"] * 1e-6) / 8e12) < 1e-9   # counters: traffic / the PROFILE's duration
    assert abs(k["amplification"] - want / k["design_bytes"]["total"]) < 1e-9
    assert abs(k["design_bytes"]["records_out"] - 65536 * 128 * 16 * 4 * 12) < 1
    # another configuration than the profiled one: no counter fields
    r2 = bench.roofline(_FakeTimer(secs), "t16s", _args(pose_grads=True), 65536, 128, 1.0, False, 9.8)
    assert r2["traffic"] is None and "frac_counter" not in r2["kernels"]["render_backward"]
