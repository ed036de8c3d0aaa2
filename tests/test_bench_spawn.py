"""bench.py --gpus N without a launcher starts N rank processes itself (one per GPU, like the
reference's tools/dist_test.sh:13-24) and forwards rank 0's line.  Runs without a GPU: the
test hook makes every rank print its rendezvous environment instead of benchmarking."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv),
                          env=env, capture_output=True, text=True, timeout=300)


def test_parent_spawns_one_rank_per_gpu_and_forwards_rank0():
    r = _run({"GANET_BENCH_SPAWN_ONLY": "1"}, "--gpus", "3", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                     # only rank 0's stdout is forwarded
    seen = json.loads(lines[0])
    assert seen["RANK"] == "0" and seen["LOCAL_RANK"] == "0" and seen["WORLD_SIZE"] == "3"
    assert seen["MASTER_ADDR"] == "127.0.0.1" and int(seen["MASTER_PORT"]) > 0


def test_worst_rank_exit_code_is_returned():
    # no GPU here: every rank exits with the "needs a GPU" message, and so does the parent
    import torch
    if torch.cuda.is_available():
        return
    r = _run({}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0


def test_launcher_world_size_must_match_gpus():
    r = _run({"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "4")
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_a_dead_rank_does_not_leave_the_parent_waiting():
    """Rank 1 exits with an error while the others would wait in a barrier for ever: the
    parent gives them a grace period, ends them and returns the failed rank's code."""
    import time
    t0 = time.monotonic()
    r = _run({"GANET_BENCH_SPAWN_ONLY": "rank1_dies", "GANET_BENCH_SPAWN_GRACE": "1"},
             "--gpus", "3", "--steps", "2", "--warmup", "1")
    assert r.returncode == 3
    assert time.monotonic() - t0 < 120
