"""CPU: the L/14-shaped miniature (patch 14 -> K 588 padded to 640; hidden 341 padded to 384) through the engine
with the per-kernel references, against goldens captured from the real reference (tests/golden/tiny14_step.npz)."""
import json
from types import SimpleNamespace

import numpy as np
import torch

from clipself_amd.config import tiny14_cfg
from clipself_amd.init import seeded_visual_state, synthetic_batch
from clipself_amd.open_clip.model import CustomCLIP
from clipself_amd.training.clipself import CLIPSelf
from clipself_amd.training.optim import FlatAdamW
from clipself_amd.training.scheduler import cosine_lr
from clipself_amd.training.train import train_step
from oracle.ops_ref import RefOps


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def run_tiny14(ops_factory, device, golden_dir, log=None):
    g = np.load(golden_dir / "tiny14_step.npz")
    rec = json.loads(str(g["recipe"]))
    cfg = tiny14_cfg()
    student, teacher = CustomCLIP(cfg, ops=ops_factory(), trainable=True), CustomCLIP(cfg, ops=ops_factory(), trainable=False)
    eng = student.visual.engine
    assert eng.Hp == 384 and eng.Kpe == 640 and cfg.hidden == 341
    for m in (student, teacher):
        m.visual.engine.load_state(seeded_visual_state(cfg, rec["seed_w"]))
    student.lock_image_tower(unlocked_groups=cfg.layers)
    student.train()
    teacher.eval()
    images, boxes, crops = synthetic_batch(rec["batch"], rec["boxes"], cfg.image_size, cfg.image_size, seed=rec["seed_b"])
    with torch.no_grad():
        t = teacher.encode_image(crops.flatten(0, 1).to(device))
        s = student.encode_pseudo_boxes(images.to(device), [b[:, :4].to(device) for b in boxes])
    assert rel(t, g["teacher"]) < 2e-2 and rel(s, g["student_roi"]) < 2e-2
    opt = FlatAdamW(student, lr=rec["lr"], weight_decay=rec["wd"])
    sched = cosine_lr(opt, rec["lr"], rec["warmup"], rec["total"])
    args = SimpleNamespace(device=device, precision="amp", distributed=False, skip_scheduler=False, grad_clip_norm=None,
                           multiscale=False, extract_type="v2", cosine_weight=1.0)
    losses, worst = [], 0.0
    for step in range(rec["steps"]):
        batch = synthetic_batch(rec["batch"], rec["boxes"], cfg.image_size, cfg.image_size, seed=rec["seed_b"] + step)
        out, _, _ = train_step(student, CLIPSelf(), batch, opt, sched, step, teacher, args)
        losses.append(out["loss"].detach().item())
        if step == 0:
            none = {str(n) for n in g["grad_none"]}
            for n, p in student.named_parameters():
                if not p.requires_grad or n in none:
                    continue
                r = rel(p.grad, g["grad/" + n])
                worst = max(worst, r)
                assert r < 6e-2, (n, r)
    assert np.allclose(losses, g["losses"], atol=1e-2), (losses, g["losses"].tolist())
    for n in ("visual.blocks.0.mlp.w3.weight", "visual.blocks.1.mlp.w1.bias", "visual.blocks.0.mlp.ffn_ln.weight"):
        assert rel(dict(student.named_parameters())[n], g["final/" + n]) < 2e-2, n
    # padding stays exactly zero through forward, backward and AdamW
    for name in ("visual.blocks.0.mlp.w3.weight", "visual.blocks.0.mlp.w1.weight", "visual.blocks.0.mlp.ffn_ln.bias", "visual.patch_embed.proj.weight"):
        for buf in (eng.master, eng.grad, eng.exp_avg):
            full, logical = eng.storage_of(buf, name), eng.view_of(buf, name)
            assert abs(float(full.double().abs().sum()) - float(logical.double().abs().sum())) < 1e-12 * max(1.0, float(full.double().abs().sum())), name
    if log:
        log(f"tiny14 worst grad rel={worst:.3e} losses={losses}")
    return worst


def test_l14_shaped_tower_with_padded_storage(golden_dir):
    run_tiny14(RefOps, "cpu", golden_dir)
