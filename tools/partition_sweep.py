#!/usr/bin/env python
"""Static CU partition between the prefetched teacher pass and the student's step: one process, one pair of towers,
the bench's step under a list of (teacher share R, student cap, hardware CU masks) settings, interleaved and repeated so that box drift
shows.  R = 0 is the shared pool (the default schedule of rounds 1-4); `inline` runs the teacher on the main stream.
usage (GPU box): python tools/partition_sweep.py [steps [passes [configs...]]]     config = R[:cap][m]   e.g. 0 16 32 32:48 32m inline"""
import json
import sys
import time
from contextlib import nullcontext
from pathlib import Path
from types import SimpleNamespace

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.init import synthetic_batch  # noqa: E402
from clipself_amd.open_clip import create_model  # noqa: E402
from clipself_amd.training.clipself import CLIPSelf, mark_all_valid  # noqa: E402
from clipself_amd.training.optim import FlatAdamW  # noqa: E402
from clipself_amd.training.scheduler import cosine_lr  # noqa: E402
from clipself_amd.training.train import train_step  # noqa: E402

argv = sys.argv[1:]
STEPS = int(argv[0]) if argv else 12
PASSES = int(argv[1]) if len(argv) > 1 else 2
CONFIGS = argv[2:] or ["0", "16", "24", "32", "48", "64", "32m", "48m", "inline"]
MODEL, BATCH, CROPS, SIZE = "EVA02-CLIP-B-16", 64, 32, 224
dev = "cuda:0"
student = create_model(MODEL, "eva", precision="amp_bf16", device=dev, cache_dir=None)
teacher = create_model(MODEL, "eva", precision="amp_bf16", device=dev, cache_dir=None, trainable=False)
teacher.visual.teacher_chunk = 2048
cfg = student.visual.cfg
student.lock_image_tower(unlocked_groups=cfg.layers)
student.train(); teacher.eval()
opt = FlatAdamW(student, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
sched = cosine_lr(opt, 1e-5, 1000, 100000)
args = SimpleNamespace(device=dev, precision="amp_bf16", distributed=False, skip_scheduler=False, grad_clip_norm=None, multiscale=False,
                       extract_type="v2", cosine_weight=1.0, teacher_prefetch=True)
batches = [tuple(t.to(dev) for t in synthetic_batch(BATCH, CROPS, SIZE, SIZE, seed=1234 + 977 * j)) for j in range(2)]
for b in batches:
    mark_all_valid(b[1], True)
torch.cuda.synchronize()
step = 0


def run(conf):
    global step
    inline = conf == "inline"
    mask = conf.endswith("m")
    body = conf.rstrip("m")
    share, cap = (0, None) if inline else ((int(body.split(":")[0]), int(body.split(":")[1])) if ":" in body else (int(body), None))
    args.teacher_prefetch = not inline
    method = CLIPSelf(partition_cus=share, partition_mask=mask, partition_cap=cap)
    sstream = method.student_stream(student.visual.engine.ops)
    ctx = torch.cuda.stream(sstream) if sstream is not None else nullcontext()
    last = None
    with ctx:
        for _ in range(3):
            train_step(student, method, batches[step % 2], opt, sched, step, teacher, args, next_batch=batches[(step + 1) % 2]); step += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            last, _, _ = train_step(student, method, batches[step % 2], opt, sched, step, teacher, args, next_batch=batches[(step + 1) % 2]); step += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    student.visual.engine.ops.cap_compute_units(0)
    loss = float(last["loss"])
    return {"config": conf, "ms_per_step": 1e3 * dt / STEPS, "images_per_s": BATCH * STEPS / dt, "loss": loss}


run("0")                                      # warm the box (clocks, allocator) before anything is recorded
for p in range(PASSES):
    for conf in CONFIGS:
        r = run(conf)
        r["pass"] = p
        print(json.dumps(r), flush=True)
