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


def test_expected_data_parallel_fields_follow_the_design_table():
    """DESIGN.md section 5's prediction table as a function of the world size (bench.py attaches it to `data_parallel` so that the first multi-GPU
    line is self-judging): 8 ranks -> 12 buckets of 28.3 MB, 49.5 MB per bucket on a rank's links, 0.2-0.8 ms per bucket, no queueing behind
    the 0.83 ms issue spacing, >= 7.8 x one GPU; one rank -> nothing on the links."""
    sys.path.insert(0, str(ROOT))
    import bench
    e8 = bench.expected_data_parallel(8)
    assert e8["buckets_per_step"] == 12 and abs(e8["allreduce_mb_per_step"] - 339.6) < 0.5 and abs(e8["mb_on_a_ranks_links_per_bucket"] - 49.5) < 0.1
    lo, hi = e8["bucket_issue_to_done_ms"]
    assert 0.15 < lo < 0.3 and 0.6 < hi < 0.9 and not e8["buckets_queue_behind_each_other"]
    assert 7.5 < e8["speedup_over_one_gpu"][0] <= e8["speedup_over_one_gpu"][1] < 8.0
    assert e8["images_per_s"][0] == round(8 * 64 / e8["ms_per_step"][1] * 1e3) or abs(e8["images_per_s"][0] - 8 * 64 / e8["ms_per_step"][1] * 1e3) < 10
    e1 = bench.expected_data_parallel(1)
    assert e1["mb_on_a_ranks_links_per_bucket"] == 0.0 and e1["reserve_cost_ms"] == 0.0 and e1["speedup_over_one_gpu"] == [1.0, 1.0]
    e2, e4 = bench.expected_data_parallel(2), bench.expected_data_parallel(4)
    assert e2["speedup_over_one_gpu"][0] > 1.9 and e4["speedup_over_one_gpu"][0] > 3.8


def test_floor_table_reproduces_from_the_tracked_launch_sequence():
    """profiles/r06_floor.md is generated, not typed: tools/floor_table.py on the tracked launch sequence of the profiled step gives the same table -- every
    launch is assigned to a class (the classes' measured times add up to the step's kernel time) and the design floor is below the measurement."""
    import re
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "floor_table.py"), str(ROOT / "profiles" / "r06_m_sequence_inline.txt")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r"\| \*\*step\*\* \| (\d+) \| \*\*([\d.]+)\*\* \| \*\*([\d.]+)\*\* \|", r.stdout)
    assert m, r.stdout[-500:]
    launches, measured, floor = int(m.group(1)), float(m.group(2)), float(m.group(3))
    head = re.search(r"(\d+) launches, ([\d.]+) ms of kernel time", r.stdout)
    assert launches == int(head.group(1)) == 599 and abs(measured - float(head.group(2))) < 0.05       # nothing dropped, nothing counted twice
    assert 70.0 < floor < measured < 95.0
    tracked = (ROOT / "profiles" / "r06_floor.md").read_text()
    assert f"**{floor:.2f}**" in tracked and f"**{measured:.2f}**" in tracked
