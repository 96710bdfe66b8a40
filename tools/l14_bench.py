#!/usr/bin/env python
"""One-GPU timing of BASELINE configs[3]'s shapes: EVA02-CLIP-L-14-336 CLIPSelf step, 16 images x 32 crops at 336^2
(577 tokens, width 1024, 24 blocks, SwiGLU hidden 2730 -> padded 2752, patch 14 -> K 640)."""
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

MODEL, B, K, S = "EVA02-CLIP-L-14-336", 16, 32, 336
dev = "cuda:0"
student = create_model(MODEL, "eva", precision="amp_bf16", device=dev, cache_dir=None)
teacher = create_model(MODEL, "eva", precision="amp_bf16", device=dev, cache_dir=None, trainable=False)
cfg = student.visual.cfg
student.lock_image_tower(unlocked_groups=cfg.layers)
student.train(); teacher.eval()
opt = FlatAdamW(student, lr=1e-5, weight_decay=0.1)
args = SimpleNamespace(device=dev, precision="amp_bf16", distributed=False, skip_scheduler=True, grad_clip_norm=None, multiscale=False,
                       extract_type="v2", cosine_weight=1.0)
batches = [tuple(t.to(dev) for t in synthetic_batch(B, K, S, S, seed=5 + j)) for j in range(2)]
method = CLIPSelf()
for i in range(2):
    out, _, _ = train_step(student, method, batches[i % 2], opt, None, i, teacher, args, next_batch=batches[(i + 1) % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 4
for i in range(n):
    out, _, _ = train_step(student, method, batches[i % 2], opt, None, 2 + i, teacher, args, next_batch=batches[(i + 1) % 2])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
N, C, Hd, E, L, p = cfg.tokens, cfg.width, cfg.hidden, cfg.embed_dim, cfg.layers, cfg.patch_size
pe = 2 * (N - 1) * 3 * p * p * C
blk = 8 * N * C * C + 4 * N * N * C + 6 * N * C * Hd
blk_na = 4 * N * C * C + 6 * N * C * Hd
F = K * (pe + L * blk + 2 * C * E) + (pe + (L - 1) * blk + blk_na + 2 * (N - 1) * C * E) + 2 * ((L - 1) * blk + blk_na) + 2 * (N - 1) * C * E
print(f"{MODEL}: {1e3 * dt:.1f} ms/step, {B / dt:.1f} images/s, {F * B / dt / 1e12:.0f} TFLOP/s (SURVEY M4 FLOPs, full last block), loss {float(out['loss'].detach()):.4f}")
