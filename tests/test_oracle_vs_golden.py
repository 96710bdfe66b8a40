"""Pins the CPU oracle (oracle/eva_ref.py) against golden vectors captured from the real
reference by oracle/gen_golden.py (fixtures: tests/golden/tiny_step.npz, b16_cfg1.npz)."""
import json

import numpy as np
import pytest
import torch

from clipself_amd.config import get_tower_cfg, tiny_cfg
from clipself_amd.init import seeded_visual_state, synthetic_batch
from oracle import eva_ref


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def tiny(golden_dir):
    g = np.load(golden_dir / "tiny_step.npz")
    return g, json.loads(str(g["recipe"]))


def test_tiny_forward_matches_reference(tiny):
    g, rec = tiny
    cfg = tiny_cfg()
    sd = seeded_visual_state(cfg, rec["seed_w"])
    batch = synthetic_batch(rec["batch"], rec["boxes"], cfg.image_size, cfg.image_size, seed=rec["seed_b"])
    with torch.no_grad():
        loss, student, teacher = eva_ref.clipself_loss(sd, sd, cfg, batch)
        dense, _ = eva_ref.encode_dense(sd, cfg, batch[0])
    assert _rel(teacher, g["teacher"]) < 2e-6
    assert _rel(student, g["student_roi"]) < 2e-6
    assert _rel(dense, g["dense"]) < 2e-6
    assert abs(float(loss) - g["losses"][0]) < 2e-6


def test_tiny_rescaled_grid_matches_reference(tiny):
    g, rec = tiny
    cfg = tiny_cfg()
    sd = seeded_visual_state(cfg, rec["seed_w"])
    im, bx, _ = synthetic_batch(2, 3, 64, cfg.image_size, seed=77)
    with torch.no_grad():
        roi = eva_ref.encode_pseudo_boxes(sd, cfg, im, [b[:, :4] for b in bx])
    assert _rel(roi, g["roi64"]) < 2e-6


def test_tiny_three_steps_grads_and_adamw(tiny):
    g, rec = tiny
    cfg = tiny_cfg()
    student = seeded_visual_state(cfg, rec["seed_w"])
    teacher = seeded_visual_state(cfg, rec["seed_w"])
    batches = [synthetic_batch(rec["batch"], rec["boxes"], cfg.image_size, cfg.image_size, seed=rec["seed_b"] + s)
               for s in range(rec["steps"])]
    # first-step grads
    s0 = {k: v.clone() for k, v in student.items()}
    log, grads = eva_ref.train_steps(s0, teacher, cfg, batches[:1], lr=rec["lr"], wd=rec["wd"],
                                     warmup=rec["warmup"], total_steps=rec["total"])
    none = sorted(n for n, v in grads.items() if v is None)
    assert none == sorted(str(x) for x in g["grad_none"])
    for n, v in grads.items():
        if v is not None:
            assert _rel(v, g["grad/" + n]) < 1e-4, n
    # full trajectory + final params
    log, _ = eva_ref.train_steps(student, teacher, cfg, batches, lr=rec["lr"], wd=rec["wd"],
                                 warmup=rec["warmup"], total_steps=rec["total"])
    assert np.allclose([l["loss"] for l in log], g["losses"], atol=5e-6)
    assert np.allclose([l["lr"] for l in log], g["lrs"], rtol=1e-12)
    for k in g.files:
        if k.startswith("final/"):
            # Adam normalises tiny gradients, so fp32 rounding noise in g is amplified in the update
            assert _rel(student[k[6:]].detach(), g[k]) < 1e-3, k


def test_param_group_rule(golden_dir):
    blob = json.loads((golden_dir / "param_groups.json").read_text())
    assert blob["census"]["decay"] == 84 and blob["census"]["no_decay"] == 169
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    from clipself_amd.init import visual_param_shapes
    shapes = visual_param_shapes(cfg)
    shapes["logit_scale"] = ()
    keep = set(eva_ref.trainable_names(shapes, cfg, cfg.layers))
    for n, grp in blob["groups"].items():
        if grp == "frozen":
            assert n not in keep, n
        else:
            assert n in keep, n
            assert eva_ref.is_no_decay(n, len(shapes[n])) == (grp == "no_decay"), n


def test_cosine_lr_schedule():
    # scheduler.py:9-10,43-53
    assert eva_ref.cosine_lr_value(0, 1e-5, 1000, 10000) == pytest.approx(1e-8)
    assert eva_ref.cosine_lr_value(999, 1e-5, 1000, 10000) == pytest.approx(1e-5)
    assert eva_ref.cosine_lr_value(1000, 1e-5, 1000, 10000) == pytest.approx(1e-5)
    assert eva_ref.cosine_lr_value(5500, 1e-5, 1000, 10000) == pytest.approx(0.5e-5)


@pytest.mark.slow
def test_b16_cfg1_matches_reference(golden_dir):
    g = np.load(golden_dir / "b16_cfg1.npz")
    rec = json.loads(str(g["recipe"]))
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    student = seeded_visual_state(cfg, rec["seed_w"])
    teacher = seeded_visual_state(cfg, rec["seed_w"])
    batches = [synthetic_batch(rec["batch"], rec["boxes"], 224, 224, seed=rec["seed_b"])]
    log, grads = eva_ref.train_steps(student, teacher, cfg, batches, lr=rec["lr"], wd=rec["wd"],
                                     warmup=rec["warmup"], total_steps=rec["total"])
    assert abs(log[0]["loss"] - g["losses"][0]) < 5e-6
    assert log[0]["lr"] == pytest.approx(g["lrs"][0])
    none = sorted(n for n, v in grads.items() if v is None)
    assert none == sorted(str(x) for x in g["grad_none"])
    norms = dict(zip((str(x) for x in g["grad_names"]), g["grad_norms"]))
    for n, v in grads.items():
        if v is not None:
            assert abs(float(v.double().norm()) - norms[n]) <= 2e-4 * norms[n] + 1e-12, n
