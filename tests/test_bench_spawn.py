"""bench.py honours --gpus N by itself (the driver's invocation form): without a launcher it re-launches under
torch.distributed.run with N ranks, rendezvous on 127.0.0.1, and rank 0 prints ONE JSON line with n_gpus = N.
--dry-run keeps the launcher / barrier / max-over-ranks / JSON plumbing and skips the GPU work, so this runs on CPU (gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    r = run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--batch", "4"])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1
    assert r["config"]["global_batch"] == 8 and r["scaling"] == "weak"
    assert r["value"] > 0 and r["ms_per_step"] > 0


def test_single_rank_default():
    r = run(["--dry-run", "--steps", "2", "--warmup", "0"])
    assert r["n_gpus"] == 1
