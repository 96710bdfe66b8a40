#!/usr/bin/env python
"""Where a step's GPU time goes, phase by phase, without CU sharing between the towers: the frozen teacher's pass, the student's forward
(dense map + RoI pooling + loss), its backward, and the optimiser, each bracketed by HIP events on one stream (inline schedule).
usage (GPU box): python tools/step_phases.py [EVA02-CLIP-B-16 [images [crops [repeats]]]]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.init import synthetic_batch  # noqa: E402
from clipself_amd.open_clip import create_model  # noqa: E402
from clipself_amd.training.optim import FlatAdamW  # noqa: E402

argv = sys.argv[1:]
MODEL = argv[0] if argv else "EVA02-CLIP-B-16"
B, K = (int(argv[1]) if len(argv) > 1 else 64), (int(argv[2]) if len(argv) > 2 else 32)
REP = int(argv[3]) if len(argv) > 3 else 5
dev = "cuda:0"
pre = "eva" if MODEL.startswith("EVA") else ""
student = create_model(MODEL, pre, device=dev, cache_dir=None)
teacher = create_model(MODEL, pre, device=dev, cache_dir=None, trainable=False)
cfg = student.visual.cfg
student.lock_image_tower(unlocked_groups=cfg.layers)
student.train(); teacher.eval()
opt = FlatAdamW(student, lr=1e-5, weight_decay=0.1)
images, boxes, crops = (t.to(dev) for t in synthetic_batch(B, K, cfg.image_size, cfg.image_size, seed=3))
rois = [b[:, :4] for b in boxes]
flat = crops.flatten(0, 1)


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    return out, (e0, e1)


acc = {"teacher": 0.0, "student fwd": 0.0, "student bwd": 0.0, "adamw": 0.0}
for it in range(REP + 2):
    opt.zero_grad()
    with torch.no_grad():
        t, ev_t = timed(lambda: teacher.encode_image(flat, normalize=True))
    s, ev_f = timed(lambda: student.encode_pseudo_boxes(images, rois, normalize=True, extract_type="v2"))
    loss = 1.0 - (s * t).sum(-1).mean()
    _, ev_b = timed(loss.backward)
    _, ev_o = timed(opt.step)
    torch.cuda.synchronize()
    if it >= 2:                                             # two warm-up iterations
        for key, (a, b) in zip(acc, (ev_t, ev_f, ev_b, ev_o)):
            acc[key] += a.elapsed_time(b) / REP
total = sum(acc.values())
print(f"{MODEL}, {B} images x {K} crops, inline: " + " | ".join(f"{k} {v:.2f} ms ({100 * v / total:.0f} %)" for k, v in acc.items())
      + f" | sum {total:.2f} ms = {B / total * 1e3:.0f} images/s", flush=True)
