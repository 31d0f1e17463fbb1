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
