"""BASELINE configs[4] "fp8 MFMA weights" on CPU: the engine's fp8-forward schedule (e4m3 weight shadows + row-quantised activations,
clipself_amd/engine.py:_linear) driven through the per-kernel references (oracle/ops_ref.py: torch.float8_e4m3fn arithmetic) --
the quantiser's contract, closeness to the bf16 schedule, a full RegionCLIP training step, and the shadow refresh after AdamW."""
from types import SimpleNamespace

import numpy as np
import torch

from clipself_amd.config import tiny_cfg
from clipself_amd.init import seeded_visual_state, synthetic_batch
from clipself_amd.open_clip.model import CustomCLIP
from oracle.ops_ref import RefOps


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_row_quantiser_contract():
    ops = RefOps()
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(37, 200, generator=g) * torch.logspace(-3, 2, 37)[:, None]).to(torch.bfloat16)
    x[5] = 0
    q, s = torch.full((37, 256), 0x55, dtype=torch.uint8), torch.empty(37)
    ops.quant_rows_fp8(x, q, s)
    assert torch.equal(q[:, 200:], torch.zeros(37, 56, dtype=torch.uint8)), "padding columns must be zero bytes"
    deq = q[:, :200].view(torch.float8_e4m3fn).float() * s[:, None]
    assert float(s[5]) == 1.0 and torch.equal(deq[5], torch.zeros(200))
    amax = x.float().abs().amax(1)
    assert torch.allclose(deq.abs().amax(1)[amax > 0], amax[amax > 0], rtol=1e-6), "the row maximum maps to +-448 exactly"
    err = (deq - x.float()).abs() / amax.clamp_min(1e-30)[:, None]
    assert float(err.max()) < 2 ** -4 + 1e-6                   # e4m3: 3 mantissa bits -> relative step 2^-3, half of it after rounding


def _model(cfg, fp8, seed=2, dgrad=False):
    m = CustomCLIP(cfg, ops=RefOps(), trainable=True)
    m.visual.engine.load_state(seeded_visual_state(cfg, seed))
    m.lock_image_tower(unlocked_groups=cfg.layers)
    m.train()
    if fp8:
        m.visual.engine.enable_fp8_forward(dgrad=dgrad)
    return m


def test_fp8_forward_tracks_the_bf16_schedule_and_trains():
    from clipself_amd.training.optim import FlatAdamW
    from clipself_amd.training.region_clip import RegionCLIP
    from clipself_amd.training.train import train_step
    from test_regionclip_cpu import regionclip_inputs
    cfg = tiny_cfg()
    images, bx, nouns = regionclip_inputs(cfg)
    rois = [b[b[:, -1] > 0.5][:, :4] for b in bx]
    base, low = _model(cfg, False), _model(cfg, True)
    with torch.no_grad():
        f_bf = base.encode_pseudo_boxes(images, rois, normalize=True)
        f_f8 = low.encode_pseudo_boxes(images, rois, normalize=True)
    r = rel(f_f8, f_bf)
    assert 1e-4 < r < 8e-2, r                                 # really quantised, and close
    args = SimpleNamespace(device="cpu", precision="amp_fp8", distributed=False, skip_scheduler=True, grad_clip_norm=None,
                           extract_type="v2", contrast_weight=1.0)
    out = {}
    for tag, m in (("bf16", base), ("fp8", low)):
        method = RegionCLIP(SimpleNamespace(), noun_embeddings=nouns)
        opt = FlatAdamW(m, lr=1e-3, weight_decay=0.1)
        w8_before = {k: v[0].clone() for k, v in m.visual.engine.w8.items()}
        losses, _, _ = train_step(m, method, (images, bx), opt, None, 0, None, args)
        out[tag] = (float(losses["loss"]), m.visual.engine.grad.clone())
        if tag == "fp8":
            changed = sum(int(not torch.equal(v[0], w8_before[k])) for k, v in m.visual.engine.w8.items())
            assert changed == len(w8_before) > 0, "every e4m3 weight shadow is refreshed after the AdamW step"
    assert abs(out["fp8"][0] - out["bf16"][0]) / out["bf16"][0] < 3e-2, out
    assert rel(out["fp8"][1], out["bf16"][1]) < 0.25
    assert torch.isfinite(out["fp8"][1]).all()


def test_fp8_dgrad_tracks_the_fp8_forward_gradients():
    """precision "amp_fp8_dgrad": the four dgrad GEMMs of every block contract e4m3 x e4m3 (dY per token row, W^T per input-feature row;
    wgrad stays bf16).  Same forward as amp_fp8, so the loss is identical; the gradients differ by the quantisation of dY / W^T only."""
    from clipself_amd.training.optim import FlatAdamW
    from clipself_amd.training.region_clip import RegionCLIP
    from clipself_amd.training.train import train_step
    from test_regionclip_cpu import regionclip_inputs
    cfg = tiny_cfg()
    images, bx, nouns = regionclip_inputs(cfg)
    args = SimpleNamespace(device="cpu", precision="amp_fp8_dgrad", distributed=False, skip_scheduler=True, grad_clip_norm=None,
                           extract_type="v2", contrast_weight=1.0)
    out = {}
    for tag, dgrad in (("fwd", False), ("dgrad", True)):
        m = _model(cfg, True, dgrad=dgrad)
        eng = m.visual.engine
        assert eng.fp8_dgrad == dgrad and (len(eng.wt8) == 4 * (cfg.layers - eng.first_trainable)) == dgrad
        method = RegionCLIP(SimpleNamespace(), noun_embeddings=nouns)
        opt = FlatAdamW(m, lr=1e-3, weight_decay=0.1)
        wt8_before = {k: v[0].clone() for k, v in eng.wt8.items()}
        losses, _, _ = train_step(m, method, (images, bx), opt, None, 0, None, args)
        out[tag] = (float(losses["loss"]), eng.grad.clone())
        if dgrad:
            assert sum(int(not torch.equal(v[0], wt8_before[k])) for k, v in eng.wt8.items()) == len(wt8_before) > 0, \
                "every e4m3 W^T shadow is refreshed after the AdamW step"
    assert out["fwd"][0] == out["dgrad"][0], "same forward"
    r = rel(out["dgrad"][1], out["fwd"][1])
    assert 1e-4 < r < 0.2, r                                  # really quantised, and close
    assert torch.isfinite(out["dgrad"][1]).all()


def test_factory_precision_amp_fp8_switches_the_engine():
    from clipself_amd.open_clip import create_model
    m = create_model("EVA02-CLIP-B-16", "eva", precision="amp_fp8", cache_dir=None, ops=RefOps())
    eng = m.visual.engine
    assert eng.fp8_forward and len(eng.w8) == 4 * eng.cfg.layers
    q, s = eng.w8[(0, "w3")]
    assert q.dtype == torch.uint8 and q.shape == (768, 2048) and s.shape == (768,)
    assert not eng.fp8_dgrad and not eng.wt8
    assert not create_model("EVA02-CLIP-B-16", "eva", precision="amp_bf16", cache_dir=None, ops=RefOps()).visual.engine.fp8_forward


def test_factory_enables_fp8_for_the_student_only():
    """`--precision amp_fp8` builds the CLIPSelf teacher and the evaluation copy with the same precision string (training/main.py); a frozen
    tower must keep its bf16 operands -- distillation targets and evaluation features may not depend on the run's precision flag -- and
    allocates no e4m3 shadows.  Only the trainable student switches its forward linears to fp8."""
    from clipself_amd.open_clip import factory
    from clipself_amd.open_clip.model import CustomCLIP
    cfg = tiny_cfg()
    made = {}

    def fake_cfg(name):
        return cfg
    orig = factory.get_tower_cfg
    factory.get_tower_cfg = fake_cfg
    try:
        for trainable in (False, True):
            made[trainable] = factory.create_model("tiny", "eva", precision="amp_fp8", ops=RefOps(), trainable=trainable)
        plain = factory.create_model("tiny", "eva", precision="amp", ops=RefOps(), trainable=False)
    finally:
        factory.get_tower_cfg = orig
    assert isinstance(made[False], CustomCLIP)
    assert not made[False].visual.engine.fp8_forward and not made[False].visual.engine.w8
    assert made[True].visual.engine.fp8_forward and made[True].visual.engine.w8
    _, _, crops = synthetic_batch(2, 3, cfg.image_size, cfg.image_size, seed=9)
    with torch.no_grad():
        a = made[False].encode_image(crops.flatten(0, 1), normalize=False)
        b = plain.encode_image(crops.flatten(0, 1), normalize=False)
    assert torch.equal(a, b), "teacher features must be bit-identical with and without amp_fp8"
    import pytest
    with pytest.raises(RuntimeError):
        made[False].visual.engine.enable_fp8_forward()


def test_fp8_rejects_rows_wider_than_the_quantiser_covers():
    import dataclasses
    import pytest
    from clipself_amd.engine import EvaEngine
    cfg = tiny_cfg()
    eng = EvaEngine(cfg, RefOps(), trainable=True)
    eng.FP8_MAX_ROW = 16
    with pytest.raises(NotImplementedError):
        eng.enable_fp8_forward()
