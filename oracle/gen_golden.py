"""TEST INFRASTRUCTURE -- golden-vector generator.  Runs only where /root/reference
exists (the build container):   python -m oracle.gen_golden

Drives the *real* reference through its public surface
  open_clip.create_model(name, 'eva', cache_dir=None)    src/open_clip/factory.py:111-149
  model.lock_image_tower(unlocked_groups=L)              src/open_clip/eva_clip/model.py:297-299
  training.clipself.CLIPSelf()(batch, model, dist_model, None, 'cpu', None, False, args)
                                                          src/training/clipself.py:7-49
  AdamW param groups as src/training/main.py:198-213, cosine_lr as src/training/scheduler.py:43-53,
  step order as src/training/train.py:80-122
with the build's seeded weights (clipself_amd/init.py) and synthetic batches, and
writes small fixtures (inputs are regenerated from seeds; only outputs are stored):

  tests/golden/tiny_step.npz   full tensors for the tiny EVA02 tower: teacher feats,
                               dense map, roi feats, loss, every gradient, parameters
                               after 3 AdamW steps, a 64-px (rescaled pos-embed/rope) case
  tests/golden/b16_cfg1.npz    EVA02-CLIP-B-16, BASELINE cfg 1 (2 img x 8 boxes, 224^2):
                               loss, feature slices, per-parameter grad norms,
                               grad-None list, 4-step loss trajectory + lr values
  tests/golden/param_groups.json  names -> decay / no-decay / frozen for B/16
"""
from __future__ import annotations

import json
import math
import os
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from clipself_amd.config import tiny_cfg, tiny14_cfg, tiny_openai_cfg, tiny_openai14_cfg, get_tower_cfg          # noqa: E402
from clipself_amd.init import seeded_visual_state, synthetic_batch  # noqa: E402
from oracle.ref_import import import_reference                   # noqa: E402

GOLD = ROOT / "tests" / "golden"

# recipe constants shared with tests/ (kept here so the fixture records them)
TINY = dict(seed_w=1, seed_b=5, batch=2, boxes=3, steps=3, lr=1e-3, wd=0.1, warmup=2, total=10)
B16 = dict(seed_w=0, seed_b=1234, batch=2, boxes=8, steps=4, lr=1e-5, wd=0.1, warmup=1000, total=10000)


def _register_tiny(oc, c=None):
    from open_clip.eva_clip import factory as eva_factory
    c = c or tiny_cfg()
    eva_factory._MODEL_CONFIGS[c.name] = {
        "embed_dim": c.embed_dim,
        "vision_cfg": {"image_size": c.image_size, "layers": c.layers, "width": c.width,
                       "head_width": c.head_width, "patch_size": c.patch_size, "mlp_ratio": c.mlp_ratio,
                       "eva_model_name": "tiny", "drop_path_rate": 0.0, "xattn": False, "fusedLN": False,
                       "rope": True, "pt_hw_seq_len": c.pt_hw_seq_len, "intp_freq": True,
                       "naiveswiglu": True, "subln": True},
        "text_cfg": {"context_length": 8, "vocab_size": 64, "width": c.text_width, "heads": c.text_heads,
                     "layers": c.text_layers, "xattn": False, "fusedLN": False},
    }
    return c


def _build(oc, cfg, seed):
    model = oc.create_model(cfg.name, "eva", cache_dir=None, device="cpu", precision="fp32")
    sd = seeded_visual_state(cfg, seed)
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(("text." in k) or ("rope" in k) or k == "logit_scale" for k in res.missing_keys), res.missing_keys
    return model


def _optimizer(model, lr, wd):
    # src/training/main.py:198-213 (EVA names contain no "vit" -> betas (0.9,0.999), eps 1e-8: params.py:5-11)
    exclude = lambda n, p: p.ndim < 2 or "bn" in n or "ln" in n or "bias" in n or "logit_scale" in n
    named = list(model.named_parameters())
    g0 = [p for n, p in named if exclude(n, p) and p.requires_grad]
    g1 = [p for n, p in named if not exclude(n, p) and p.requires_grad]
    opt = torch.optim.AdamW([{"params": g0, "weight_decay": 0.0}, {"params": g1, "weight_decay": wd}],
                            lr=lr, betas=(0.9, 0.999), eps=1e-8)
    groups = {}
    for n, p in named:
        groups[n] = "frozen" if not p.requires_grad else ("no_decay" if exclude(n, p) else "decay")
    return opt, groups


def _build_openai(oc, cfg, seed):
    """open_clip.create_model(name, '') -> model.CLIP with the OpenAI-style VisionTransformer (src/open_clip/factory.py:163-205)."""
    from open_clip import factory
    factory._MODEL_CONFIGS[cfg.name] = {
        "embed_dim": cfg.embed_dim, "quick_gelu": cfg.quick_gelu,
        "vision_cfg": {"image_size": cfg.image_size, "layers": cfg.layers, "width": cfg.width, "patch_size": cfg.patch_size},
        "text_cfg": {"context_length": cfg.text_context, "vocab_size": cfg.text_vocab, "width": cfg.text_width, "heads": cfg.text_heads,
                     "layers": cfg.text_layers}}
    model = oc.create_model(cfg.name, "", device="cpu", precision="fp32")
    res = model.load_state_dict(seeded_visual_state(cfg, seed), strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert not any(k.startswith("visual.") for k in res.missing_keys), res.missing_keys
    return model


def _run_steps(oc, cfg, rec, image_size, crop_size, build=None):
    from training.clipself import CLIPSelf
    from training.scheduler import cosine_lr
    build = build or _build
    student = build(oc, cfg, rec["seed_w"])
    teacher = build(oc, cfg, rec["seed_w"])          # main.py:150-157 loads the same checkpoint
    if rec.get("lock", True):                        # main.py:161-166: without --lock-image the whole visual tower trains
        student.lock_image_tower(unlocked_groups=rec.get("unlocked", cfg.layers))
    student.train()
    teacher.eval()
    opt, groups = _optimizer(student, rec["lr"], rec["wd"])
    sched = cosine_lr(opt, rec["lr"], rec["warmup"], rec["total"])
    method = CLIPSelf()
    args = SimpleNamespace(multiscale=False, extract_type="v2", cosine_weight=1.0)
    out = {"losses": [], "lrs": []}
    first = {}
    for step in range(rec["steps"]):
        batch = synthetic_batch(rec["batch"], rec["boxes"], image_size, crop_size, seed=rec["seed_b"] + step)
        out["lrs"].append(sched(step))
        opt.zero_grad()
        losses, bs, logit_scale = method(batch, student, teacher, None, "cpu", None, False, args)
        total = sum(losses.values())
        total.backward()
        if step == 0:
            first["grads"] = {n: (p.grad.detach().clone() if p.grad is not None else None)
                              for n, p in student.named_parameters() if p.requires_grad}
            first["bs"] = bs
            first["logit_scale_exp"] = float(logit_scale)
            with torch.no_grad():
                rois = [b[b[:, -1] > 0.5, :4] for b in batch[1]]
                crops = torch.cat([c[b[:, -1] > 0.5] for b, c in zip(batch[1], batch[2])])
                first["teacher"] = teacher.encode_image(crops, normalize=False)
                first["student_roi"] = student.encode_pseudo_boxes(batch[0], rois, normalize=False, extract_type="v2")
                first["dense"] = student.encode_dense(batch[0], normalize=False, keep_shape=False)
        opt.step()
        with torch.no_grad():
            student.logit_scale.clamp_(0, math.log(100))
        out["losses"].append(float(total.detach()))
    return student, teacher, out, first, groups


def gen_tiny(oc):
    cfg = _register_tiny(oc)
    rec = TINY
    student, teacher, out, first, groups = _run_steps(oc, cfg, rec, cfg.image_size, cfg.image_size)
    blob = {"losses": np.array(out["losses"], np.float64), "lrs": np.array(out["lrs"], np.float64),
            "teacher": first["teacher"].numpy(), "student_roi": first["student_roi"].numpy(),
            "dense": first["dense"].numpy()}
    none = []
    for n, g in first["grads"].items():
        if g is None:
            none.append(n)
        else:
            blob["grad/" + n] = g.numpy()
    for n, p in student.named_parameters():
        if n.startswith("visual.") and p.requires_grad:
            blob["final/" + n] = p.detach().numpy()
    blob["grad_none"] = np.array(none)
    # non-native grid: 64-px student image -> 8x8 tokens: pos-embed bicubic rescale + rope.recalculate
    fresh = _build(oc, cfg, rec["seed_w"])
    fresh.eval()
    im, bx, _ = synthetic_batch(2, 3, 64, cfg.image_size, seed=77)
    with torch.no_grad():
        blob["roi64"] = fresh.encode_pseudo_boxes(im, [b[:, :4] for b in bx], normalize=False, extract_type="v2").numpy()
    blob["recipe"] = np.array(json.dumps(rec))
    np.savez_compressed(GOLD / "tiny_step.npz", **blob)
    print("tiny losses", out["losses"], "grad_none", none)


def gen_tiny_unlocked(oc):
    """training.main WITHOUT --lock-image (src/training/main.py:161-166): nothing but the text tower is frozen, so the dense path also
    differentiates the stem (patch_embed.proj, cls_token, pos_embed), the final norm and the head (eva_vit_model.py:537-544,615-623).
    Every gradient of step 0, the parameters after 3 AdamW steps, the optimizer grouping, and the pos_embed / cls_token gradients on a
    64-px image (8x8 tokens: the gradient runs back through rescale_positional_embedding's bicubic resize, :631-643)."""
    cfg = _register_tiny(oc)
    rec = dict(TINY, lock=False, seed_w=4, seed_b=60)
    student, teacher, out, first, groups = _run_steps(oc, cfg, rec, cfg.image_size, cfg.image_size)
    blob = {"losses": np.array(out["losses"], np.float64), "lrs": np.array(out["lrs"], np.float64)}
    none = []
    for n, g in first["grads"].items():
        if g is None:
            none.append(n)
        else:
            blob["grad/" + n] = g.numpy()
    for n, p in student.named_parameters():
        if n.startswith("visual.") and p.requires_grad:
            blob["final/" + n] = p.detach().numpy()
    blob["grad_none"] = np.array(none)
    blob["groups"] = np.array(json.dumps({n: v for n, v in groups.items() if not n.startswith("text.")}))
    from training.clipself import CLIPSelf
    fresh, frozen = _build(oc, cfg, rec["seed_w"]), _build(oc, cfg, rec["seed_w"])
    fresh.train()
    frozen.eval()
    batch = synthetic_batch(2, 3, 64, cfg.image_size, seed=78)
    losses, _, _ = CLIPSelf()(batch, fresh, frozen, None, "cpu", None, False, SimpleNamespace(multiscale=False, extract_type="v2", cosine_weight=1.0))
    sum(losses.values()).backward()
    for n in ("visual.pos_embed", "visual.cls_token", "visual.patch_embed.proj.weight", "visual.head.weight"):
        blob["grad64/" + n] = dict(fresh.named_parameters())[n].grad.numpy()
    blob["loss64"] = np.array(float(sum(losses.values())))
    blob["recipe"] = np.array(json.dumps(rec))
    np.savez_compressed(GOLD / "tiny_unlocked_step.npz", **blob)
    print("tiny unlocked losses", out["losses"], "grad_none", none, "trainable", sum(v != "frozen" for v in groups.values()))


CURVE = dict(TINY, seed_w=7, seed_b=300, steps=24, n_batches=4, warmup=4, total=24, lr=2e-3)


def gen_curve(oc):
    """Loss curve of the real reference over 24 optimiser steps on the tiny tower (4 distinct batches cycled, cosine schedule with warm-up,
    lr high enough that the loss moves by ~40 %): the `loss curve matching reference` check of the north star at a size every test can run."""
    from training.clipself import CLIPSelf
    from training.scheduler import cosine_lr
    cfg = _register_tiny(oc)
    rec = CURVE
    student, teacher = _build(oc, cfg, rec["seed_w"]), _build(oc, cfg, rec["seed_w"])
    student.lock_image_tower(unlocked_groups=cfg.layers)
    student.train()
    teacher.eval()
    opt, _ = _optimizer(student, rec["lr"], rec["wd"])
    sched = cosine_lr(opt, rec["lr"], rec["warmup"], rec["total"])
    method, args = CLIPSelf(), SimpleNamespace(multiscale=False, extract_type="v2", cosine_weight=1.0)
    losses, lrs = [], []
    for step in range(rec["steps"]):
        batch = synthetic_batch(rec["batch"], rec["boxes"], cfg.image_size, cfg.image_size, seed=rec["seed_b"] + step % rec["n_batches"])
        lrs.append(sched(step))
        opt.zero_grad()
        out, _, _ = method(batch, student, teacher, None, "cpu", None, False, args)
        total = sum(out.values())
        total.backward()
        opt.step()
        losses.append(float(total.detach()))
    np.savez_compressed(GOLD / "tiny_curve.npz", losses=np.array(losses, np.float64), lrs=np.array(lrs, np.float64),
                        recipe=np.array(json.dumps(rec)))
    print("curve", [round(x, 4) for x in losses])


def gen_tiny14(oc):
    """L/14-shaped miniature (patch 14, hidden 341): pins the zero-padded storage paths."""
    cfg = _register_tiny(oc, tiny14_cfg())
    rec = dict(TINY, seed_w=2, seed_b=9, steps=2)
    student, teacher, out, first, groups = _run_steps(oc, cfg, rec, cfg.image_size, cfg.image_size)
    blob = {"losses": np.array(out["losses"], np.float64), "lrs": np.array(out["lrs"], np.float64),
            "teacher": first["teacher"].numpy(), "student_roi": first["student_roi"].numpy(), "dense": first["dense"].numpy()}
    none = []
    for n, g in first["grads"].items():
        if g is None:
            none.append(n)
        else:
            blob["grad/" + n] = g.numpy()
    for n, p in student.named_parameters():
        if n.startswith("visual.") and p.requires_grad:
            blob["final/" + n] = p.detach().numpy()
    blob["grad_none"] = np.array(none)
    blob["recipe"] = np.array(json.dumps(rec))
    np.savez_compressed(GOLD / "tiny14_step.npz", **blob)
    print("tiny14 losses", out["losses"])


def gen_tiny_openai(oc):
    """OpenAI-CLIP ViT family: the CLIPSelf step through model.CLIP / transformer.VisionTransformer, GELU and
    QuickGELU variants, plus a non-native grid (positional-embedding rescale)."""
    blob = {}
    for tag, quick, steps in (("", False, 3), ("q/", True, 1)):
        cfg = tiny_openai_cfg(quick)
        rec = dict(TINY, seed_w=3, seed_b=21, steps=steps, unlocked=cfg.layers)
        student, teacher, out, first, groups = _run_steps(oc, cfg, rec, cfg.image_size, cfg.image_size, build=_build_openai)
        blob.update({tag + "losses": np.array(out["losses"], np.float64), tag + "lrs": np.array(out["lrs"], np.float64),
                     tag + "teacher": first["teacher"].numpy(), tag + "student_roi": first["student_roi"].numpy(),
                     tag + "dense": first["dense"].numpy()})
        none = []
        for n, g in first["grads"].items():
            if g is None:
                none.append(n)
            elif not quick or n.endswith(("mlp.c_fc.weight", "ln_1.weight")):
                blob[tag + "grad/" + n] = g.numpy()
        blob[tag + "grad_none"] = np.array(none)
        if not quick:
            for n, p in student.named_parameters():
                if n.startswith("visual.") and p.requires_grad:
                    blob["final/" + n] = p.detach().numpy()
            blob["frozen"] = np.array([n for n, p in student.named_parameters() if n.startswith("visual.") and not p.requires_grad])
            fresh = _build_openai(oc, cfg, rec["seed_w"])
            fresh.eval()
            im, bx, _ = synthetic_batch(2, 3, 64, cfg.image_size, seed=78)
            with torch.no_grad():
                blob["roi64"] = fresh.encode_pseudo_boxes(im, [b[:, :4] for b in bx], normalize=False, extract_type="v2").numpy()
            blob["recipe"] = np.array(json.dumps(rec))
        print("tiny openai", "quick" if quick else "gelu", "losses", out["losses"], "grad_none", none)
    # ViT-L/14-shaped miniature (patch 14: zero-padded conv1 storage), one step: features, loss, three gradients
    cfg = tiny_openai14_cfg()
    rec = dict(TINY, seed_w=4, seed_b=31, steps=1, unlocked=cfg.layers)
    student, teacher, out, first, groups = _run_steps(oc, cfg, rec, cfg.image_size, cfg.image_size, build=_build_openai)
    blob.update({"p14/losses": np.array(out["losses"], np.float64), "p14/teacher": first["teacher"].numpy(), "p14/student_roi": first["student_roi"].numpy(),
                 "p14/recipe": np.array(json.dumps(rec))})
    for n in ("visual.transformer.resblocks.0.attn.in_proj_weight", "visual.transformer.resblocks.1.mlp.c_proj.weight", "visual.transformer.resblocks.0.ln_1.bias"):
        blob["p14/grad/" + n] = first["grads"][n].numpy()
    np.savez_compressed(GOLD / "tiny_openai_step.npz", **blob)


def gen_tiny_openai_stem(oc):
    """VisionTransformer.lock with more groups than blocks (transformer.py:391-422: groups = [[conv1, class_embedding, ln_pre],
    positional_embedding, block 0 .. L-1]): L + 1 unlocks the positional embedding, L + 2 the stem as well; ln_post / proj stay frozen.
    One step at L + 1, three at L + 2 (native grid), one at L + 2 on a 64-px image (gradient through the bicubic rescale of the
    positional embedding, transformer.py:724-734), three with no lock at all (the whole visual tower trains)."""
    cfg = tiny_openai_cfg()
    L = cfg.layers
    blob = {}
    for tag, unlocked, steps, size in (("pos/", L + 1, 1, cfg.image_size), ("stem/", L + 2, 3, cfg.image_size), ("stem64/", L + 2, 1, 64),
                                       ("all/", None, 3, cfg.image_size)):
        # "all/": no lock_image_tower() call at all (training.main without --lock-image, main.py:161-166): ln_post and proj train as well
        rec = dict(TINY, seed_w=3, seed_b=41, steps=steps, unlocked=unlocked, lock=unlocked is not None)
        student, teacher, out, first, groups = _run_steps(oc, cfg, rec, size, cfg.image_size, build=_build_openai)
        blob[tag + "losses"] = np.array(out["losses"], np.float64)
        blob[tag + "lrs"] = np.array(out["lrs"], np.float64)
        blob[tag + "groups"] = np.array(json.dumps({n: k for n, k in groups.items() if n.startswith("visual.")}))
        blob[tag + "recipe"] = np.array(json.dumps(dict(rec, image_size=size)))
        none = []
        for n, g in first["grads"].items():
            if g is None:
                none.append(n)
            elif n.startswith("visual.") and (".resblocks." not in n or n.endswith(("resblocks.0.ln_1.weight", "resblocks.0.attn.in_proj_weight",
                                                                                   "resblocks.1.mlp.c_fc.weight"))):
                blob[tag + "grad/" + n] = g.numpy()
        blob[tag + "grad_none"] = np.array(none)
        if steps > 1:
            for n, p in student.named_parameters():
                if n.startswith("visual.") and p.requires_grad and ".resblocks." not in n:
                    blob[tag + "final/" + n] = p.detach().numpy()
        print("tiny openai", tag, "losses", out["losses"], "trainable", sorted(n for n, k in groups.items() if k != "frozen" and ".resblocks." not in n))
    np.savez_compressed(GOLD / "tiny_openai_stem.npz", **blob)


def gen_tiny_openai_maskattn(oc):
    """OpenAI-CLIP family only: extract_type='v1' and encode_masks(mask_attn=True) (transformer.py:515-521,660-671,736-834) -- every mask /
    box becomes an extra query token (a copy of the CLS embedding after ln_pre) that runs through all blocks attending the CLS token and
    the image tokens inside its mask; nobody attends the extra tokens.  Inference vectors: 2 images, 3 + 2 masks (one of them empty), boxes
    on the native 4x4 grid and on a 64-px image (8x8 grid, rescaled positional embedding)."""
    blob = {}
    for tag, quick in (("", False), ("q/", True)):
        cfg = tiny_openai_cfg(quick)
        model = _build_openai(oc, cfg, 3)
        model.eval()
        gen = torch.Generator().manual_seed(97)
        images = torch.randn(2, 3, cfg.image_size, cfg.image_size, generator=gen)
        g = cfg.image_size // cfg.patch_size
        masks = [torch.rand(3, g, g, generator=gen) > 0.5, torch.rand(2, g, g, generator=gen) > 0.4]
        masks[0][1] = False                                  # an empty mask: its token attends the CLS token only
        boxes = [torch.tensor([[0.05, 0.10, 0.60, 0.70], [0.30, 0.26, 0.95, 0.99], [0.55, 0.55, 0.70, 0.60]]),
                 torch.tensor([[0.0, 0.0, 1.0, 1.0], [0.26, 0.51, 0.74, 0.76]])]
        with torch.no_grad():
            blob[tag + "mask_attn"] = model.encode_masks(images, masks, normalize=False, mask_attn=True).numpy()
            blob[tag + "mask_attn_normalized"] = model.encode_masks(images, masks, normalize=True, mask_attn=True).numpy()
            blob[tag + "v1"] = model.encode_pseudo_boxes(images, boxes, normalize=False, extract_type="v1").numpy()
            if not quick:
                images64 = torch.randn(2, 3, 64, 64, generator=gen)
                blob["v1_64"] = model.encode_pseudo_boxes(images64, boxes, normalize=False, extract_type="v1").numpy()
                blob["images64"] = images64.numpy()
        if not quick:
            blob["images"] = images.numpy()
            for i, (m, b) in enumerate(zip(masks, boxes)):
                blob[f"masks{i}"] = m.numpy()
                blob[f"boxes{i}"] = b.numpy()
        print("tiny openai mask_attn", "quick" if quick else "gelu", blob[tag + "mask_attn"].shape, float(np.abs(blob[tag + "mask_attn"]).mean()))
    np.savez_compressed(GOLD / "tiny_openai_maskattn.npz", **blob)


def gen_vitb16(oc):
    """OpenAI-CLIP ViT-B/16 at BASELINE cfg-1 size (2 images x 8 boxes, 224^2), nn.GELU variant (`--pretrained ''` path of the factory):
    loss trajectory, feature slices, every gradient norm."""
    cfg = get_tower_cfg("ViT-B-16")
    rec = dict(B16, seed_w=5, seed_b=4321, steps=2, unlocked=cfg.layers)
    student, teacher, out, first, groups = _run_steps(oc, cfg, rec, 224, 224, build=_build_openai)
    names, norms, none = [], [], []
    for n, g in first["grads"].items():
        if g is None:
            none.append(n)
        else:
            names.append(n)
            norms.append(float(g.double().norm()))
    blob = {"losses": np.array(out["losses"], np.float64), "lrs": np.array(out["lrs"], np.float64),
            "teacher_slice": first["teacher"][:4, :16].numpy(), "student_roi_slice": first["student_roi"][:4, :16].numpy(),
            "teacher_rownorm": first["teacher"].norm(dim=-1).numpy(), "student_rownorm": first["student_roi"].norm(dim=-1).numpy(),
            "cos": torch.nn.functional.cosine_similarity(first["teacher"], first["student_roi"], dim=-1).numpy(),
            "grad_names": np.array(names), "grad_norms": np.array(norms, np.float64), "grad_none": np.array(none),
            "recipe": np.array(json.dumps(rec))}
    for n in ("visual.transformer.resblocks.11.mlp.c_proj.bias", "visual.transformer.resblocks.0.ln_1.weight",
              "visual.transformer.resblocks.5.attn.in_proj_bias", "visual.transformer.resblocks.11.attn.in_proj_bias"):
        blob["grad/" + n] = first["grads"][n].numpy()
    np.savez_compressed(GOLD / "vitb16_cfg1.npz", **blob)
    print("vit-b/16 losses", out["losses"], "grad_none", none)


def regionclip_inputs(cfg, n_nouns=150, batch=5, boxes=24, seed=31):
    """Seeded RegionCLIP batch: (images, boxes [B,k,6] = xyxy, label, valid) with >= 100 distinct labels so that the
    federated column set is deterministic (no multinomial draw), and a seeded noun-embedding bank."""
    g = np.random.Generator(np.random.PCG64(seed))
    images, nb, _ = synthetic_batch(batch, boxes, cfg.image_size, cfg.image_size, seed=seed)
    labels = torch.from_numpy(g.permutation(n_nouns)[: batch * boxes].astype(np.float32)).reshape(batch, boxes, 1)
    bx = torch.cat([nb[..., :4], labels, nb[..., 4:5]], dim=-1)
    bx[0, 3, -1] = 0.0                                   # one invalid box
    nouns = torch.from_numpy(g.standard_normal((n_nouns, cfg.embed_dim)).astype(np.float32))
    return images, bx, nouns


def gen_regionclip(oc):
    """RegionCLIP.__call__ (src/training/region_clip.py:28-67) on the tiny tower."""
    import tempfile
    from training.region_clip import RegionCLIP
    cfg = _register_tiny(oc)
    student = _build(oc, cfg, 4)
    student.lock_image_tower(unlocked_groups=cfg.layers)
    student.train()
    images, bx, nouns = regionclip_inputs(cfg)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "nouns.npy")
        np.save(path, nouns.numpy())
        method = RegionCLIP(SimpleNamespace(train_embed_path=path))
    args = SimpleNamespace(extract_type="v2", contrast_weight=1.0)
    losses, bs, temp = method((images, bx), student, None, None, "cpu", None, False, args)
    total = sum(losses.values())
    total.backward()
    blob = {"loss": np.float64(total.detach()), "temp": np.float64(temp), "bs": np.int64(bs)}
    for n in ("visual.blocks.0.norm1.weight", "visual.blocks.1.mlp.w3.weight", "visual.blocks.0.attn.q_proj.weight",
              "visual.blocks.1.attn.v_bias"):
        blob["grad/" + n] = dict(student.named_parameters())[n].grad.numpy()
    np.savez_compressed(GOLD / "tiny_regionclip.npz", **blob)
    print("regionclip loss", float(total))


ZEROSHOT = dict(seed_w=6, steps=2, batch=3, boxes=5, num_classes=12, seed=4321)


def gen_zeroshot(oc):
    """zero_shot.run + macc_with_is_thing (src/training/zero_shot.py:11-173) of the reference on the tiny tower, fed with the
    synthetic panoptic-style validation batches of clipself_amd/training/data.py (inputs are seeded, so tests rebuild them)."""
    from training import zero_shot as ref_zs
    from clipself_amd.training.data import SyntheticPanopticVal
    cfg = _register_tiny(oc)
    rec = ZEROSHOT
    model = _build(oc, cfg, rec["seed_w"])
    model.eval()
    val = SyntheticPanopticVal(rec["steps"], rec["batch"], rec["boxes"], cfg.image_size, cfg.image_size, cfg.image_size // cfg.patch_size,
                               cfg.embed_dim, num_classes=rec["num_classes"], seed=rec["seed"])
    class _Loader:
        dataset = val

        def __iter__(self):
            return iter(val.batches)

        def __len__(self):
            return len(val.batches)

    args = SimpleNamespace(device="cpu", precision="fp32", distributed=False, horovod=False, extract_type="v2", image_ave_pool=False,
                           rank=0, local_rank=0)
    (hit_rois, hit_crops, hit_mask, sim_rois, sim_crops, sim_mask, sizes, thing, labels) = ref_zs.run(model, _Loader(), args)
    metrics = {}
    for key, h in (("rois", hit_rois), ("crops", hit_crops), ("maskpool", hit_mask)):
        metrics.update(ref_zs.macc_with_is_thing(h, thing, labels, key))
    blob = {"hit_rois": hit_rois.numpy(), "hit_crops": hit_crops.numpy(), "hit_maskpool": hit_mask.numpy(),
            "sim_rois": sim_rois.numpy(), "sim_crops": sim_crops.numpy(), "sim_maskpool": sim_mask.numpy(),
            "size": sizes.numpy(), "thing": thing.numpy(), "label": labels.numpy(),
            "metric_names": np.array(list(metrics)), "metric_values": np.array([metrics[k] for k in metrics], np.float64),
            "recipe": np.array(json.dumps(rec))}
    np.savez_compressed(GOLD / "tiny_zeroshot.npz", **blob)
    print("zero-shot metrics", metrics)


PARAM_CASES = [["--model", "EVA02-CLIP-B-16"],
               ["--model", "EVA02-CLIP-L-14-336", "--lr", "1e-5", "--wd", "0.1", "--lock-image", "--lock-image-unlocked-groups", "24",
                "--dataset-type", "proposals_distill", "--batch-size", "16", "--alpha", "0.7", "--precision", "amp_bf16", "--epochs", "6"],
               ["--model", "ViT-B-16", "--dataset-type", "region_clip", "--max-boxes", "32", "--crop-scale", "1.5", "--multiscale",
                "--aug-cfg", "scale=(0.4, 1.0)", "use_timm=True"]]


def gen_params(oc):
    """Namespaces produced by the reference's training.params.parse_args (src/training/params.py:25-476) for a few command lines:
    every flag the reference knows, its default (incl. the model-dependent Adam defaults) and its parsed type."""
    from training.params import parse_args as ref_parse
    out = []
    for argv in PARAM_CASES:
        ns = vars(ref_parse(list(argv)))
        out.append({"argv": argv, "namespace": {k: (v if isinstance(v, (int, float, str, bool, type(None), list, dict)) else repr(v))
                                                 for k, v in sorted(ns.items())}})
    (GOLD / "params_namespaces.json").write_text(json.dumps(out, indent=0))
    print("params:", [len(c["namespace"]) for c in out])


def gen_schedules(oc):
    """LR schedules of src/training/scheduler.py (cosine_lr :43-53, const_lr :13-21, const_lr_cooldown :24-40) sampled at fixed steps."""
    from training import scheduler as ref_s
    opt = SimpleNamespace(param_groups=[{"lr": 0.0}, {"lr": 0.0}])
    steps = [0, 1, 5, 99, 100, 101, 500, 2500, 4999, 5000, 7777, 9999]
    out = {"steps": steps,
           "cosine": [float(ref_s.cosine_lr(opt, 1e-5, 100, 10000)(t)) for t in steps],
           "const": [float(ref_s.const_lr(opt, 3e-4, 100, 10000)(t)) for t in steps],
           "cooldown": [float(ref_s.const_lr_cooldown(opt, 3e-4, 100, 10000, 2000, 2.0, 1e-6)(t)) for t in steps]}
    (GOLD / "lr_schedules.json").write_text(json.dumps(out))
    print("schedules", out["cosine"][:4])


def gen_b16(oc):
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    rec = B16
    student, teacher, out, first, groups = _run_steps(oc, cfg, rec, 224, 224)
    names, norms, none = [], [], []
    for n, g in first["grads"].items():
        if g is None:
            none.append(n)
        else:
            names.append(n)
            norms.append(float(g.double().norm()))
    blob = {"losses": np.array(out["losses"], np.float64), "lrs": np.array(out["lrs"], np.float64),
            "teacher_slice": first["teacher"][:4, :16].numpy(), "student_roi_slice": first["student_roi"][:4, :16].numpy(),
            "teacher_rownorm": first["teacher"].norm(dim=-1).numpy(),
            "student_rownorm": first["student_roi"].norm(dim=-1).numpy(),
            "cos": torch.nn.functional.cosine_similarity(first["teacher"], first["student_roi"], dim=-1).numpy(),
            "grad_names": np.array(names), "grad_norms": np.array(norms, np.float64),
            "grad_none": np.array(none), "logit_scale_exp": np.float64(first["logit_scale_exp"]),
            "recipe": np.array(json.dumps(rec))}
    # a few full small gradients for element-wise checks
    for n in ("visual.blocks.11.mlp.w3.bias", "visual.blocks.0.norm1.weight", "visual.blocks.5.attn.q_bias",
              "visual.blocks.11.attn.v_bias"):
        blob["grad/" + n] = first["grads"][n].numpy()
    np.savez_compressed(GOLD / "b16_cfg1.npz", **blob)
    census = {"decay": sum(1 for v in groups.values() if v == "decay"),
              "no_decay": sum(1 for v in groups.values() if v == "no_decay"),
              "frozen": sum(1 for v in groups.values() if v == "frozen")}
    (GOLD / "param_groups.json").write_text(json.dumps(
        {"census": census, "groups": {n: v for n, v in groups.items() if not n.startswith("text.")}}, indent=0))
    print("b16 losses", out["losses"], "grad_none", none, census)


B16_CURVE = dict(B16, seed_w=0, seed_b=5000, steps=24, n_batches=4, lr=5e-5, warmup=4, total=24)


def gen_b16_curve(oc):
    """Loss curve of the real reference at a real tower size: EVA02-CLIP-B-16 at BASELINE cfg-1 shape (2 images x 8 boxes, 224^2), 24 optimiser
    steps of train.py:80-122 with the AdamW groups of main.py:198-213, 4 distinct batches cycled, warm-up + cosine decay, an lr at which the
    loss moves by tens of percent (the shipped recipe's 1e-5 with 1000 warm-up steps moves nothing in 24 steps)."""
    from training.clipself import CLIPSelf
    from training.scheduler import cosine_lr
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    rec = B16_CURVE
    student, teacher = _build(oc, cfg, rec["seed_w"]), _build(oc, cfg, rec["seed_w"])
    student.lock_image_tower(unlocked_groups=cfg.layers)
    student.train()
    teacher.eval()
    opt, _ = _optimizer(student, rec["lr"], rec["wd"])
    sched = cosine_lr(opt, rec["lr"], rec["warmup"], rec["total"])
    method, args = CLIPSelf(), SimpleNamespace(multiscale=False, extract_type="v2", cosine_weight=1.0)
    losses, lrs = [], []
    for step in range(rec["steps"]):
        batch = synthetic_batch(rec["batch"], rec["boxes"], 224, 224, seed=rec["seed_b"] + step % rec["n_batches"])
        lrs.append(sched(step))
        opt.zero_grad()
        out, _, _ = method(batch, student, teacher, None, "cpu", None, False, args)
        total = sum(out.values())
        total.backward()
        opt.step()
        with torch.no_grad():
            student.logit_scale.clamp_(0, math.log(100))
        losses.append(float(total.detach()))
        print("b16 curve step", step, losses[-1], flush=True)
    np.savez_compressed(GOLD / "b16_curve.npz", losses=np.array(losses, np.float64), lrs=np.array(lrs, np.float64),
                        recipe=np.array(json.dumps(rec)))


STRESS = dict(seed_w=0, seed_b=1234, batch=2, boxes=8, crops=4, row_offset_sigmas=(5.0, 1.5))


def gen_stress(oc):
    """The real reference on weights with trained-like activation statistics (oracle/stress_weights.py: outlier channels x50..x200 and rows
    with |mean| / sigma = 5, resp. 1.5): teacher features of a few crops, student RoI features, the loss -- what pins the fp32 oracle in the
    regime the folded LayerNorms of the frozen schedule are sensitive to."""
    from oracle.stress_weights import trained_statistics_state
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    rec = STRESS
    blob = {"recipe": np.array(json.dumps(rec))}
    for ros in rec["row_offset_sigmas"]:
        model = oc.create_model(cfg.name, "eva", cache_dir=None, device="cpu", precision="fp32")
        res = model.load_state_dict(trained_statistics_state(cfg, rec["seed_w"], row_offset_sigmas=ros), strict=False)
        assert not res.unexpected_keys
        model.eval()
        images, boxes, crops = synthetic_batch(rec["batch"], rec["boxes"], 224, 224, seed=rec["seed_b"])
        flat = crops.flatten(0, 1)[:rec["crops"]]
        with torch.no_grad():
            t = model.encode_image(flat, normalize=False)
            s = model.encode_pseudo_boxes(images, [b[:, :4] for b in boxes], normalize=False, extract_type="v2")
        tag = f"ros{ros:g}/"
        blob[tag + "teacher"] = t.numpy()
        blob[tag + "student_roi"] = s.numpy()
        print("stress", ros, "teacher norm", float(t.norm()), "student norm", float(s.norm()), flush=True)
    np.savez_compressed(GOLD / "b16_stress.npz", **blob)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    oc = import_reference()
    GOLD.mkdir(parents=True, exist_ok=True)
    if "--tiny14-only" in sys.argv:
        gen_tiny14(oc)
        return
    if "--regionclip-only" in sys.argv:
        gen_regionclip(oc)
        return
    if "--zeroshot-only" in sys.argv:
        gen_zeroshot(oc)
        return
    if "--unlocked-only" in sys.argv:
        gen_tiny_unlocked(oc)
        return
    if "--curve-only" in sys.argv:
        gen_curve(oc)
        return
    if "--openai-stem-only" in sys.argv:
        gen_tiny_openai_stem(oc)
        return
    if "--openai-maskattn-only" in sys.argv:
        gen_tiny_openai_maskattn(oc)
        return
    if "--openai-only" in sys.argv:
        gen_tiny_openai(oc)
        gen_tiny_openai_stem(oc)
        gen_tiny_openai_maskattn(oc)
        if "--tiny-only" not in sys.argv:
            gen_vitb16(oc)
        return
    if "--b16-curve-only" in sys.argv:
        gen_b16_curve(oc)
        return
    if "--stress-only" in sys.argv:
        gen_stress(oc)
        return
    if "--params-only" in sys.argv:
        gen_params(oc)
        gen_schedules(oc)
        return
    gen_tiny(oc)
    gen_tiny_unlocked(oc)
    gen_tiny14(oc)
    gen_regionclip(oc)
    gen_curve(oc)
    gen_tiny_openai(oc)
    gen_tiny_openai_stem(oc)
    gen_tiny_openai_maskattn(oc)
    gen_zeroshot(oc)
    gen_params(oc)
    gen_schedules(oc)
    if "--tiny-only" not in sys.argv:
        gen_b16(oc)
        gen_vitb16(oc)
        gen_b16_curve(oc)
        gen_stress(oc)


if __name__ == "__main__":
    main()
