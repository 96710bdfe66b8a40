#!/usr/bin/env python
"""The reference's own training recipe on one GPU (scripts/train_clipself_coco_image_patches_eva_vit{b16,l14}.sh:1-8): --batch-size 2 per GPU,
grid_distill boxes (max_boxes 20, on average 13 valid per image), crops at the tower's native size, the student's image at --det-image-size
1024 (B/16) / 896 (L/14-336) = a 64 x 64 token grid, 4097 tokens, where attention is ~47 % of a block's FLOPs.  Prints ms/step (overlapped and
inline schedule), the inline phases, step FLOPs (SURVEY.md §8 M4 formulas at these shapes) and the share of the MFMA peak.
usage (GPU box): python tools/recipe_bench.py [EVA02-CLIP-B-16|EVA02-CLIP-L-14-336 [det_size [images [steps]]]]"""
import json
import sys
import time
from pathlib import Path
from types import SimpleNamespace

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.init import synthetic_batch  # noqa: E402
from clipself_amd.open_clip import create_model  # noqa: E402
from clipself_amd.training.clipself import CLIPSelf  # noqa: E402
from clipself_amd.training.optim import FlatAdamW  # noqa: E402
from clipself_amd.training.train import train_step  # noqa: E402

argv = sys.argv[1:]
MODEL = argv[0] if argv else "EVA02-CLIP-B-16"
DET = int(argv[1]) if len(argv) > 1 else (1024 if "B-16" in MODEL else 896)
B = int(argv[2]) if len(argv) > 2 else 2
STEPS = int(argv[3]) if len(argv) > 3 else 10
MAXB, PEAK = 20, 2500.0
dev = "cuda:0"
student = create_model(MODEL, "eva", precision="amp_bf16", device=dev, cache_dir=None)
teacher = create_model(MODEL, "eva", precision="amp_bf16", device=dev, cache_dir=None, trainable=False)
cfg = student.visual.cfg
student.lock_image_tower(unlocked_groups=cfg.layers)
student.train(); teacher.eval()
opt = FlatAdamW(student, lr=1e-5, weight_decay=0.1)
args = SimpleNamespace(device=dev, precision="amp_bf16", distributed=False, skip_scheduler=True, grad_clip_norm=None, multiscale=False,
                       extract_type="v2", cosine_weight=1.0, teacher_prefetch=True)
batches = [tuple(t.to(dev) for t in synthetic_batch(B, MAXB, DET, cfg.image_size, seed=11 + j, valid_prob=0.65)) for j in range(2)]
nvalid = [int((b[1][..., -1] > 0.5).sum()) for b in batches]


def run(prefetch, steps):
    args.teacher_prefetch = prefetch
    method = CLIPSelf()
    for i in range(3):
        train_step(student, method, batches[i % 2], opt, None, i, teacher, args, next_batch=batches[(i + 1) % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out, _, _ = train_step(student, method, batches[(i + 1) % 2], opt, None, i, teacher, args, next_batch=batches[i % 2])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, float(out["loss"].detach())


def phases(rep=5):
    images, boxes, crops = batches[0]
    valid = boxes[..., -1] > 0.5
    flat = crops[valid]
    rois = [b[v][:, :4] for b, v in zip(boxes, valid)]
    acc = [0.0] * 4
    for it in range(rep + 2):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        opt.zero_grad()
        ev[0].record()
        with torch.no_grad():
            t = teacher.encode_image(flat, normalize=True)
        ev[1].record()
        s = student.encode_pseudo_boxes(images, rois, normalize=True, extract_type="v2")
        loss = 1.0 - (s * t).sum(-1).mean()
        ev[2].record()
        loss.backward()
        ev[3].record()
        opt.step()
        ev[4].record()
        torch.cuda.synchronize()
        if it >= 2:
            for k in range(4):
                acc[k] += ev[k].elapsed_time(ev[k + 1]) / rep
    return dict(zip(("teacher", "student_fwd", "student_bwd", "adamw"), acc))


g = DET // cfg.patch_size
N, Nt, C, Hd, E, L, p = g * g + 1, cfg.tokens, cfg.width, cfg.hidden, cfg.embed_dim, cfg.layers, cfg.patch_size
blk = lambda n: 8 * n * C * C + 4 * n * n * C + 6 * n * C * Hd
blk_na = lambda n: 4 * n * C * C + 6 * n * C * Hd
pe = lambda n: 2 * (n - 1) * 3 * p * p * C
T = pe(Nt) + L * blk(Nt) + 2 * C * E
Sf = pe(N) + (L - 1) * blk(N) + blk_na(N) + 2 * (N - 1) * C * E
Sb = 2 * ((L - 1) * blk(N) + blk_na(N)) + 2 * (N - 1) * C * E
kmean = sum(nvalid) / len(nvalid)
F = kmean * T + B * (Sf + Sb)
attn_f, attn_b = B * (L - 1) * 4 * N * N * C, B * (L - 1) * 10 * N * N * C
ms_o, loss = run(True, STEPS)
ms_i, _ = run(False, STEPS)
ph = phases()
out = {"model": MODEL, "det_image_size": DET, "student_tokens": N, "images": B, "valid_crops_per_step": kmean, "crop_size": cfg.image_size,
       "ms_per_step_overlapped": ms_o, "ms_per_step_inline": ms_i, "images_per_s": B / min(ms_o, ms_i) * 1e3, "loss": loss,
       "step_tflop": F / 1e12, "step_mfma_frac": F / 1e12 / (min(ms_o, ms_i) * 1e-3) / PEAK, "phases_inline_ms": ph,
       "teacher_tflop": kmean * T / 1e12, "student_fwd_tflop": B * Sf / 1e12, "student_bwd_tflop": B * Sb / 1e12,
       "attention_fwd_tflop": attn_f / 1e12, "attention_bwd_tflop": attn_b / 1e12,
       "student_fwd_frac": B * Sf / 1e12 / (ph["student_fwd"] * 1e-3) / PEAK, "student_bwd_frac": B * Sb / 1e12 / (ph["student_bwd"] * 1e-3) / PEAK}
print(json.dumps(out), flush=True)
