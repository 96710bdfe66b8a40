"""Training WITHOUT --lock-image on CPU (the engine's schedule through the per-kernel references): the reference then trains the whole
visual tower (src/training/main.py:161-166) -- cls_token, pos_embed, patch_embed.proj, the final norm and the head besides the blocks.
Golden = tests/golden/tiny_unlocked_step.npz, captured from the real reference by oracle/gen_golden.py::gen_tiny_unlocked."""
import json
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from clipself_amd.config import tiny_cfg
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


def _args():
    return SimpleNamespace(device="cpu", precision="amp", distributed=False, skip_scheduler=False, grad_clip_norm=None, multiscale=False,
                           extract_type="v2", cosine_weight=1.0)


def build_pair(cfg, seed, ops_cls=RefOps):
    student, teacher = CustomCLIP(cfg, ops=ops_cls(), trainable=True), CustomCLIP(cfg, ops=ops_cls(), trainable=False)
    for m in (student, teacher):
        m.visual.engine.load_state(seeded_visual_state(cfg, seed))
    student.train()
    teacher.eval()
    return student, teacher


STEM_HEAD = ("visual.cls_token", "visual.pos_embed", "visual.patch_embed.proj.weight", "visual.patch_embed.proj.bias", "visual.norm.weight",
             "visual.norm.bias", "visual.head.weight", "visual.head.bias")


def check_unlocked_step(golden_dir, ops_cls, device, tol_grad, tol_final):
    g = np.load(golden_dir / "tiny_unlocked_step.npz")
    rec = json.loads(str(g["recipe"]))
    cfg = tiny_cfg()
    student, teacher = build_pair(cfg, rec["seed_w"], ops_cls)
    eng = student.visual.engine
    assert eng.train_all and eng.first_trainable == 0          # a fresh model is unlocked, like the reference's
    groups = json.loads(str(g["groups"]))
    named = dict(student.named_parameters())
    for n, kind in groups.items():
        assert named[n].requires_grad == (kind != "frozen"), n
    opt = FlatAdamW(student, lr=rec["lr"], betas=(0.9, 0.999), eps=1e-8, weight_decay=rec["wd"])
    decay = {id(p) for p in opt.param_groups[1]["params"]}
    for n, kind in groups.items():
        if kind != "frozen":
            assert (id(named[n]) in decay) == (kind == "decay"), n
    sched = cosine_lr(opt, rec["lr"], rec["warmup"], rec["total"])
    args = _args()
    args.device = device
    losses, worst = [], 0.0
    for step in range(rec["steps"]):
        batch = synthetic_batch(rec["batch"], rec["boxes"], cfg.image_size, cfg.image_size, seed=rec["seed_b"] + step)
        out, _, _ = train_step(student, CLIPSelf(), tuple(t.to(device) for t in batch), opt, sched, step, teacher, args)
        losses.append(float(out["loss"].detach()))
        if step == 0:
            none = {str(n) for n in g["grad_none"]}
            for n, p in student.named_parameters():
                if not p.requires_grad:
                    continue
                if n in none:
                    assert p.grad is None, n
                    continue
                assert p.grad is not None, n
                r = rel(p.grad, g["grad/" + n])
                worst = max(worst, r)
                assert r < tol_grad, (n, r)
    assert np.allclose(losses, g["losses"], atol=1e-2), (losses, g["losses"])
    for n in STEM_HEAD + ("visual.blocks.0.mlp.w1.weight",):
        assert rel(named[n], g["final/" + n]) < tol_final, (n, rel(named[n], g["final/" + n]))
    return worst


def check_rescaled_grid_gradients(golden_dir, ops_cls, device, tol):
    """64-px images on the 32-px tiny tower: the pos_embed gradient runs back through the bicubic rescale (eva_vit_model.py:631-643)."""
    g = np.load(golden_dir / "tiny_unlocked_step.npz")
    rec = json.loads(str(g["recipe"]))
    cfg = tiny_cfg()
    student, teacher = build_pair(cfg, rec["seed_w"], ops_cls)
    batch = tuple(t.to(device) for t in synthetic_batch(2, 3, 64, cfg.image_size, seed=78))
    args = _args()
    args.device = device
    losses, _, _ = CLIPSelf()(batch, student, teacher, None, device, None, False, args)
    sum(losses.values()).backward()
    assert float(sum(losses.values())) == pytest.approx(float(g["loss64"]), abs=2e-3)
    named = dict(student.named_parameters())
    for n in ("visual.pos_embed", "visual.cls_token", "visual.patch_embed.proj.weight", "visual.head.weight"):
        assert rel(named[n].grad, g["grad64/" + n]) < tol, (n, rel(named[n].grad, g["grad64/" + n]))


def test_unlocked_tower_matches_the_reference_goldens(golden_dir):
    worst = check_unlocked_step(golden_dir, RefOps, "cpu", tol_grad=6e-2, tol_final=2e-2)
    assert worst > 0.0


def test_unlocked_pos_embed_gradient_through_the_bicubic_rescale(golden_dir):
    check_rescaled_grid_gradients(golden_dir, RefOps, "cpu", 6e-2)


def test_lock_after_unlock_and_back():
    cfg = tiny_cfg()
    student, _ = build_pair(cfg, 1)
    eng = student.visual.engine
    student.lock_image_tower(unlocked_groups=1)
    assert not eng.train_all and eng.first_trainable == cfg.layers - 1
    named = dict(student.named_parameters())
    assert not named["visual.pos_embed"].requires_grad and named[f"visual.blocks.{cfg.layers - 1}.mlp.w3.weight"].requires_grad
    active = (eng.flags & 1).bool()
    o, st = eng.offsets["visual.head.weight"]
    assert not bool(active[o // 64])
    student.visual.unlock()
    assert eng.train_all and named["visual.pos_embed"].requires_grad and bool((eng.flags & 1).bool()[o // 64])
    assert eng.bucket_range("stem") == (0, eng.block_ranges[0][0]) and eng.bucket_range("head")[0] == eng.block_ranges[-1][1]
    assert set(eng.trainable_names()) == set(eng.public_names())


def test_lock_all_blocks_then_unlock_reattaches_stem_and_head_gradients():
    """lock(unlocked_groups=L) and unlock() both start training at block 0 but differ in the stem / head flags, which live in a buffer that is
    rewritten in place: the cached "which parameters does the dense path reach" mask must follow the flag version, or the
    stem / head parameters never get a .grad attached while the engine's AdamW still updates them.  The transposed / e4m3 shadows that only
    trainable blocks need follow the lock state as well."""
    cfg = tiny_cfg()
    student, teacher = build_pair(cfg, 5)
    eng = student.visual.engine
    L = cfg.layers
    named = dict(student.named_parameters())
    batch = synthetic_batch(2, 3, cfg.image_size, cfg.image_size, seed=9)
    opt = FlatAdamW(student, lr=1e-3, weight_decay=0.1)

    student.lock_image_tower(unlocked_groups=L)
    train_step(student, CLIPSelf(), batch, opt, None, 0, teacher, _args())
    assert all(named[n].grad is None for n in STEM_HEAD)
    assert named["visual.blocks.0.mlp.w3.weight"].grad is not None

    student.visual.unlock()
    train_step(student, CLIPSelf(), batch, opt, None, 1, teacher, _args())
    for n in STEM_HEAD:
        assert named[n].grad is not None and float(named[n].grad.abs().sum()) > 0.0, n

    # shadows follow the lock state: frozen blocks drop their W^T copies, re-trained ones get them back
    student.lock_image_tower(unlocked_groups=1)
    assert sorted({k[0] for k in eng.wt if isinstance(k, tuple)}) == [L - 1]
    eng.enable_fp8_forward(True, dgrad=True)
    assert sorted({k[0] for k in eng.wt8}) == [L - 1]
    student.lock_image_tower(unlocked_groups=L)
    assert sorted({k[0] for k in eng.wt if isinstance(k, tuple)}) == list(range(L)) == sorted({k[0] for k in eng.wt8})
    train_step(student, CLIPSelf(), batch, opt, None, 2, teacher, _args())       # fp8 dgrad of the newly trainable blocks finds its shadows
