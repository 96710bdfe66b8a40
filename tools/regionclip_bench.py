#!/usr/bin/env python
"""One-GPU timing of BASELINE configs[4]'s shapes: EVA02-CLIP-L-14-336 RegionCLIP (region-text) step, 32 images x <= 20 boxes at 336^2
against a 4764 x 768 noun bank -- student forward (24 blocks, 577 tokens), RoIAlign, federated BCE, backward, AdamW; with the bf16
forward and with "fp8 MFMA weights" (precision amp_fp8: forward linears on e4m3 operands; amp_fp8_dgrad: the dgrad GEMMs as well).  One JSON line per precision.
usage (GPU box): python tools/regionclip_bench.py [steps]"""
import json
import sys
import time
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.init import synthetic_batch  # noqa: E402
from clipself_amd.open_clip import create_model  # noqa: E402
from clipself_amd.training.optim import FlatAdamW  # noqa: E402
from clipself_amd.training.region_clip import RegionCLIP  # noqa: E402
from clipself_amd.training.train import train_step  # noqa: E402

MODEL, B, KBOX, S, NOUNS = "EVA02-CLIP-L-14-336", 32, 20, 336, 4764
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda:0"
g = np.random.Generator(np.random.PCG64(7))
images, nb, _ = synthetic_batch(B, KBOX, S, 32, seed=7)
labels = torch.from_numpy(g.integers(0, NOUNS, size=(B, KBOX, 1)).astype(np.float32))
valid = torch.from_numpy((g.random((B, KBOX, 1)) < 0.7).astype(np.float32))
valid[:, 0] = 1.0
batch = (images.to(dev), torch.cat([nb[..., :4], labels, valid], dim=-1).to(dev))
nouns = torch.from_numpy(g.standard_normal((NOUNS, 768)).astype(np.float32))
for precision in ("amp_bf16", "amp_fp8", "amp_fp8_dgrad"):
    model = create_model(MODEL, "eva", precision=precision, device=dev, cache_dir=None)
    cfg = model.visual.cfg
    model.lock_image_tower(unlocked_groups=cfg.layers)
    model.train()
    method = RegionCLIP(SimpleNamespace(), noun_embeddings=nouns).to(dev)
    opt = FlatAdamW(model, lr=1e-5, weight_decay=0.1)
    args = SimpleNamespace(device=dev, precision=precision, distributed=False, skip_scheduler=True, grad_clip_norm=None, extract_type="v2",
                           contrast_weight=1.0)
    for i in range(2):
        out, _, _ = train_step(model, method, batch, opt, None, i, None, args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out, _, _ = train_step(model, method, batch, opt, None, 2 + i, None, args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    N, C, Hd, E, L, p = cfg.tokens, cfg.width, cfg.hidden, cfg.embed_dim, cfg.layers, cfg.patch_size
    pe = 2 * (N - 1) * 3 * p * p * C
    blk = 8 * N * C * C + 4 * N * N * C + 6 * N * C * Hd
    blk_na = 4 * N * C * C + 6 * N * C * Hd
    F = (pe + (L - 1) * blk + blk_na + 2 * (N - 1) * C * E) + 2 * ((L - 1) * blk + blk_na) + 2 * (N - 1) * C * E     # S_f + S_b
    print(json.dumps({"metric": "images/sec (RegionCLIP region-text step), ViT-L/14-336", "value": B / dt, "unit": "images/sec", "n_gpus": 1,
                      "steps": steps, "ms_per_step": 1e3 * dt, "dtype": {"amp_fp8": "fp8 (e4m3 forward operands) + bf16", "amp_fp8_dgrad": "fp8 (e4m3 forward and dgrad operands) + bf16"}.get(precision, "bf16"),
                      "data": "synthetic", "step_tflops": F * B / dt / 1e12,
                      "config": {"workload": f"{MODEL} RegionCLIP, {B} images x <= {KBOX} boxes, {S}^2, {NOUNS} nouns (BASELINE configs[4])",
                                 "precision": precision, "loss_last_step": float(out["loss"].detach())}}), flush=True)
    del model, opt, method
    torch.cuda.empty_cache()
