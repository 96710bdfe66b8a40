#!/usr/bin/env python
"""The configurations beside the headline one, short runs, one JSON line each -- `bench.py` starts this file once per workload (its own
process: a fresh allocator, and a fault here cannot take the headline line with it) after its timed region and attaches the lines as the
`also` key (the other BASELINE configurations and the reference's own recipe shape should be driver-observed).

    python tools/also_bench.py recipe_b16 | recipe_l14 | cfg2 | cfg3 | cfg4_bf16 | cfg4_fp8 | openai_b16   [steps]

  recipe_b16  the reference's shipped recipe on one GPU (scripts/train_clipself_coco_image_patches_eva_vitb16.sh:1-8): EVA02-CLIP-B-16, 2 images
              per GPU, student at 1024^2 = 4097 tokens, <= 20 grid crops at 224^2 (13 valid on average)
  recipe_l14  the same recipe for EVA02-CLIP-L-14-336 (scripts/train_clipself_coco_image_patches_eva_vitl14.sh): student at 896^2, crops at 336^2
  cfg2        BASELINE configs[2] shapes on one GPU: EVA02-CLIP-B-16 CLIPSelf region proposals, 64 images x 20 box slots (70 % valid: ragged batch), 224^2
  cfg3        BASELINE configs[3] shapes on one GPU: EVA02-CLIP-L-14-336 CLIPSelf, 16 images x 32 crops at 336^2
  cfg4_bf16   BASELINE configs[4] shapes on one GPU: EVA02-CLIP-L-14-336 RegionCLIP, 32 images x <= 20 boxes, 4764 nouns, bf16
  cfg4_fp8    ... with e4m3 forward and dgrad operands (precision amp_fp8_dgrad: "fp8 MFMA weights")
  openai_b16  the headline step with the OpenAI-CLIP ViT-B/16 towers

Fields: workload, ms_per_step, images_per_s, step_tflop (SURVEY.md section 8 M4 formulas at the workload's shapes, full last teacher block), step_mfma_frac
(of 2.5 PFLOP/s), steps, loss_last_step.  Synthetic data, random-init weights (seeded), inputs resident in HBM."""
import json
import sys
import time
from pathlib import Path
from types import SimpleNamespace

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
PEAK = 2500.0
WORKLOADS = ("recipe_b16", "recipe_l14", "cfg2", "cfg3", "cfg4_bf16", "cfg4_fp8", "openai_b16")


def _flops(cfg, n_student, n_teacher, teacher_crops, images):
    C, Hd, E, L, p = cfg.width, cfg.hidden, cfg.embed_dim, cfg.layers, cfg.patch_size
    mlp = 4 if getattr(cfg, "arch", "eva02") == "openai" else 6          # c_fc + c_proj (GELU MLP) vs w1 | w2 | w3 (SwiGLU): 2 MACs per weight
    blk = lambda n: 8 * n * C * C + 4 * n * n * C + mlp * n * C * Hd
    blk_na = lambda n: 4 * n * C * C + mlp * n * C * Hd
    pe = lambda n: 2 * (n - 1) * 3 * p * p * C
    T = pe(n_teacher) + L * blk(n_teacher) + 2 * C * E
    Sf = pe(n_student) + (L - 1) * blk(n_student) + blk_na(n_student) + 2 * (n_student - 1) * C * E
    Sb = 2 * ((L - 1) * blk(n_student) + blk_na(n_student)) + 2 * (n_student - 1) * C * E
    return teacher_crops * T + images * (Sf + Sb)


def _time_steps(step_fn, warm, steps):
    for i in range(warm):
        step_fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for i in range(steps):
        out = step_fn(warm + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, float(out["loss"].detach())


def clipself_workload(model_name, pretrained, images, max_boxes, det, valid_prob, steps, warm=3):
    from clipself_amd.init import synthetic_batch
    from clipself_amd.open_clip import create_model
    from clipself_amd.training.clipself import CLIPSelf, mark_all_valid
    from clipself_amd.training.optim import FlatAdamW
    from clipself_amd.training.train import train_step
    dev = "cuda:0"
    student = create_model(model_name, pretrained, precision="amp_bf16", device=dev, cache_dir=None)
    teacher = create_model(model_name, pretrained, precision="amp_bf16", device=dev, cache_dir=None, trainable=False)
    cfg = student.visual.cfg
    student.lock_image_tower(unlocked_groups=cfg.layers)
    student.train()
    teacher.eval()
    opt = FlatAdamW(student, lr=1e-5, weight_decay=0.1)
    args = SimpleNamespace(device=dev, precision="amp_bf16", distributed=False, skip_scheduler=True, grad_clip_norm=None, multiscale=False,
                           extract_type="v2", cosine_weight=1.0, teacher_prefetch=True)
    kw = dict(valid_prob=valid_prob) if valid_prob < 1.0 else {}
    batches = [tuple(t.to(dev) for t in synthetic_batch(images, max_boxes, det, cfg.image_size, seed=11 + j, **kw)) for j in range(2)]
    if valid_prob >= 1.0:
        for b in batches:
            mark_all_valid(b[1], True)
    crops = sum(int((b[1][..., -1] > 0.5).sum()) for b in batches) / len(batches)
    method = CLIPSelf()
    dt, loss = _time_steps(lambda i: train_step(student, method, batches[i % 2], opt, None, i, teacher, args, next_batch=batches[(i + 1) % 2])[0], warm, steps)
    g = det // cfg.patch_size
    return cfg, dt, loss, _flops(cfg, g * g + 1, cfg.tokens, crops, images), crops


def regionclip_workload(precision, steps, warm=2):
    import numpy as np
    from clipself_amd.init import synthetic_batch
    from clipself_amd.open_clip import create_model
    from clipself_amd.training.optim import FlatAdamW
    from clipself_amd.training.region_clip import RegionCLIP
    from clipself_amd.training.train import train_step
    MODEL, B, KBOX, S, NOUNS = "EVA02-CLIP-L-14-336", 32, 20, 336, 4764
    dev = "cuda:0"
    g = np.random.Generator(np.random.PCG64(7))
    images, nb, _ = synthetic_batch(B, KBOX, S, 32, seed=7)
    labels = torch.from_numpy(g.integers(0, NOUNS, size=(B, KBOX, 1)).astype(np.float32))
    valid = torch.from_numpy((g.random((B, KBOX, 1)) < 0.7).astype(np.float32))
    valid[:, 0] = 1.0
    batch = (images.to(dev), torch.cat([nb[..., :4], labels, valid], dim=-1).to(dev))
    nouns = torch.from_numpy(g.standard_normal((NOUNS, 768)).astype(np.float32))
    model = create_model(MODEL, "eva", precision=precision, device=dev, cache_dir=None)
    cfg = model.visual.cfg
    model.lock_image_tower(unlocked_groups=cfg.layers)
    model.train()
    method = RegionCLIP(SimpleNamespace(), noun_embeddings=nouns).to(dev)
    opt = FlatAdamW(model, lr=1e-5, weight_decay=0.1)
    args = SimpleNamespace(device=dev, precision=precision, distributed=False, skip_scheduler=True, grad_clip_norm=None, extract_type="v2", contrast_weight=1.0)
    dt, loss = _time_steps(lambda i: train_step(model, method, batch, opt, None, i, None, args)[0], warm, steps)
    return cfg, dt, loss, _flops(cfg, cfg.tokens, cfg.tokens, 0, B), B


def run(which, steps):
    if which == "recipe_b16":
        cfg, dt, loss, F, crops = clipself_workload("EVA02-CLIP-B-16", "eva", 2, 20, 1024, 0.65, steps)
        n, name = 2, f"EVA02-CLIP-B-16 CLIPSelf, the reference's recipe shape: 2 images at 1024^2 (4097 student tokens) x <= 20 grid crops at 224^2 ({crops:.1f} valid per step)"
    elif which == "recipe_l14":
        cfg, dt, loss, F, crops = clipself_workload("EVA02-CLIP-L-14-336", "eva", 2, 20, 896, 0.65, steps)
        n, name = 2, f"EVA02-CLIP-L-14-336 CLIPSelf, the reference's recipe shape: 2 images at 896^2 (4097 student tokens) x <= 20 grid crops at 336^2 ({crops:.1f} valid per step)"
    elif which == "cfg2":
        cfg, dt, loss, F, crops = clipself_workload("EVA02-CLIP-B-16", "eva", 64, 20, 224, 0.7, steps)
        n, name = 64, f"EVA02-CLIP-B-16 CLIPSelf region-proposals step, 64 images x 20 box slots ({crops:.0f} valid crops per step: ragged), 224^2 (BASELINE configs[2] shapes, one GPU)"
    elif which == "cfg3":
        cfg, dt, loss, F, crops = clipself_workload("EVA02-CLIP-L-14-336", "eva", 16, 32, 336, 1.0, steps, warm=2)
        n, name = 16, "EVA02-CLIP-L-14-336 CLIPSelf image-patches step, 16 images x 32 crops, 336^2 (BASELINE configs[3] shapes, one GPU)"
    elif which in ("cfg4_bf16", "cfg4_fp8"):
        precision = "amp_bf16" if which == "cfg4_bf16" else "amp_fp8_dgrad"
        cfg, dt, loss, F, n = regionclip_workload(precision, steps)
        name = (f"EVA02-CLIP-L-14-336 RegionCLIP (region-text) step, 32 images x <= 20 boxes, 336^2, 4764 nouns (BASELINE configs[4] shapes, one GPU), "
                + ("bf16" if which == "cfg4_bf16" else "e4m3 forward + dgrad operands (fp8 MFMA weights), bf16 wgrad"))
    elif which == "openai_b16":
        cfg, dt, loss, F, crops = clipself_workload("ViT-B-16", "", 64, 32, 224, 1.0, steps)
        n, name = 64, "OpenAI-CLIP ViT-B/16 CLIPSelf image-patches step, 64 images x 32 crops, 224^2 (the headline workload on the other tower family)"
    else:
        raise SystemExit(f"unknown workload {which!r}; one of {WORKLOADS}")
    return {"id": which, "workload": name, "ms_per_step": 1e3 * dt, "images_per_s": n / dt, "step_tflop": F / 1e12,
            "step_mfma_frac": F / 1e12 / dt / PEAK, "steps": steps, "dtype": "fp8+bf16" if which == "cfg4_fp8" else "bf16", "loss_last_step": loss}


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "recipe_b16"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    print(json.dumps(run(which, steps)), flush=True)
