"""bench.py's launch contract on CPU (no GPU touched: `--dry-run` stops after the rendezvous): `--gpus N` on its own becomes N ranks,
a launcher-provided world must equal `--gpus`, and N = 1 stays a single process.  The reference's launch line is torchrun --nproc_per_node 8
(scripts/train_clipself_coco_image_patches_eva_vitb16.sh:1)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


def test_gpus_2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines == [{"dry_run": True, "n_gpus": 2, "ranks_seen": 2}]          # one line, from rank 0, after an all-reduce over both ranks


def test_single_process_default_and_world_mismatch():
    r = _run(["--dry-run"])
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1]) == {"dry_run": True, "n_gpus": 1, "ranks_seen": 1}
    bad = _run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "4", "RANK": "0"})
    assert bad.returncode != 0 and "WORLD_SIZE=4" in bad.stderr
    assert _run(["--gpus", "0"]).returncode != 0
