#!/usr/bin/env python
"""Throughput of the GPU region-crop pipeline (cs_crop_resize_u8): 64 decoded 480x640 images x 32 grid crops -> 224^2, plus the
det image of each (BASELINE configs[1] worth of teacher inputs)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.hip import HipOps  # noqa: E402
from clipself_amd.training.data import GpuGridDistillLoader  # noqa: E402

ops = HipOps()
g = torch.Generator().manual_seed(0)
imgs = [torch.randint(0, 256, (480, 640, 3), generator=g, dtype=torch.uint8).cuda() for _ in range(64)]
loader = GpuGridDistillLoader(imgs, ops, batch_size=64, max_boxes=32, det_size=224, crop_size=224, max_split=6, steps=3, seed=0)
it = iter(loader)
next(it)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
batch = next(it)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
k = int(batch[1][..., 4].sum())
print(f"one batch (64 images, {k} crops + 64 det images): {ms:.1f} ms GPU time -> {64 / ms * 1e3:.0f} images/s, {k / ms * 1e3:.0f} crops/s")
