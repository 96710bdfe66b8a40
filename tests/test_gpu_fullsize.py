"""`-m gpu`: every BASELINE.json configuration at its FULL size on the MI355X -- the launches `bench.py` times (M = 2048 crops x 197
tokens = 403 456 rows per GEMM, outputs past 2^31 bytes) and the L/14-336 RegionCLIP configuration -- against the CPU oracle on
sampled rows / a scaled-down batch and through size-independent properties.  Measured errors go to gpurun_out/parity_metrics.txt
(copied to profiles/r02_parity.md)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clipself_amd.config import get_tower_cfg              # noqa: E402
from clipself_amd.init import seeded_visual_state, synthetic_batch  # noqa: E402
from test_gpu_step import _args, _pair, one_minus_cos, rel  # noqa: E402


def _log(msg):
    from pathlib import Path
    p = Path(__file__).resolve().parent.parent / "gpurun_out"
    p.mkdir(exist_ok=True)
    with open(p / "parity_metrics.txt", "a") as f:
        f.write(msg + "\n")


def test_cfg1_teacher_all_2048_crops_in_one_pass_against_the_oracle():
    """BASELINE configs[1], teacher side: all 64 x 32 = 2048 crops in ONE pass (the launch shape of bench.py: every GEMM has
    M = 403 456 rows) must equal the 256-crop chunked schedule bit for bit, and 64 crops sampled over the whole row range (first /
    last tiles included) must match oracle/eva_ref.encode_image (fp32 CPU restatement of the reference, pinned on its goldens)."""
    from oracle import eva_ref
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    _, teacher = _pair(cfg, 0)
    _, _, crops = synthetic_batch(64, 32, 224, 224, seed=1234)
    crops = crops.flatten(0, 1)
    dev = crops.cuda()
    with torch.no_grad():
        teacher.visual.teacher_chunk = 2048
        one_pass = teacher.encode_image(dev)
        teacher.visual.teacher_chunk = 256
        chunked = teacher.encode_image(dev)
    assert one_pass.shape == (2048, cfg.embed_dim) and torch.isfinite(one_pass).all()
    assert torch.equal(one_pass, chunked), f"{int((one_pass != chunked).any(-1).sum())} of 2048 rows differ between chunk 2048 and chunk 256"
    idx = torch.unique(torch.cat([torch.arange(0, 2048, 37), torch.tensor([1, 255, 256, 1023, 1024, 2046, 2047])]))[:64]
    with torch.no_grad():
        want = eva_ref.encode_image(seeded_visual_state(cfg, 0), cfg, crops[idx])
    got = one_pass[idx.cuda()]
    r, c = rel(got, want), one_minus_cos(got, want)
    _log(f"cfg1 full size teacher (2048 crops, one pass) vs oracle on {len(idx)} sampled crops: rel-L2 {r:.3e}, max 1-cos {c:.2e}")
    assert r < 2e-2 and c < 2e-4


def test_cfg1_full_size_step_inline_and_prefetch_schedules_agree():
    """BASELINE configs[1], whole step at full size (64 images x 32 crops): finite loss and gradients, the loss equals an fp64 recomputation
    from the step's own features, student RoI features of two images match the oracle, and the one-batch-ahead teacher schedule
    (bench.py default) produces the same loss and gradient as the inline schedule (`--no-overlap`)."""
    from oracle import eva_ref
    from clipself_amd.training.clipself import CLIPSelf
    from clipself_amd.training.optim import FlatAdamW
    from clipself_amd.training.train import train_step
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    batch = tuple(t.cuda() for t in synthetic_batch(64, 32, 224, 224, seed=1234))
    nxt = tuple(t.cuda() for t in synthetic_batch(64, 32, 224, 224, seed=2211))

    def run(prefetch):
        student, teacher = _pair(cfg, 0)
        opt = FlatAdamW(student, lr=1e-5, weight_decay=0.1)
        method = CLIPSelf()
        a = _args(skip_scheduler=True)
        a.teacher_prefetch = prefetch
        if prefetch:                                   # step 0 on `nxt` launches the teacher pass over `batch` on the side stream
            train_step(student, method, nxt, FlatAdamW(student, lr=0.0, weight_decay=0.0), None, 0, teacher, a, next_batch=batch)
            assert method._pending is not None
        out, _, _ = train_step(student, method, batch, opt, None, 0, teacher, a)
        torch.cuda.synchronize()
        return float(out["loss"].detach()), student.visual.engine.grad.clone(), student, teacher

    loss_i, grad_i, student, teacher = run(False)
    assert np.isfinite(loss_i) and 0.0 < loss_i < 2.0
    assert torch.isfinite(grad_i).all() and float(grad_i.abs().sum()) > 0
    loss_p, grad_p, _, _ = run(True)
    loss_r, grad_r, _, _ = run(False)                  # the inline schedule again: run-to-run floor of the backward (RoIAlign's fp32
    gn_i, gn_p = float(grad_i.double().norm()), float(grad_p.double().norm())   # atomics reorder; bf16 operand roundings amplify that)
    _log(f"cfg1 full size step: loss inline {loss_i:.6f} prefetch {loss_p:.6f}; |grad| inline {gn_i:.6e} prefetch {gn_p:.6e}; "
         f"grad rel prefetch-vs-inline {rel(grad_p, grad_i):.2e}, inline-vs-inline repeat {rel(grad_r, grad_i):.2e}")
    assert abs(loss_i - loss_p) < 1e-6 and abs(loss_i - loss_r) < 1e-6
    assert abs(gn_i - gn_p) / gn_i < 1e-4 and rel(grad_p, grad_i) < 2e-3
    # the loss kernel at K = 2048 boxes against fp64 on the step's own features
    images, boxes, crops = batch
    with torch.no_grad():
        t = teacher.encode_image(crops.flatten(0, 1)).double()
        idx = torch.arange(64, device="cuda", dtype=torch.float32).repeat_interleave(32)[:, None]
        fresh_student, _ = _pair(cfg, 0)
        s = fresh_student.encode_pseudo_boxes(images, torch.cat([idx, boxes[..., :4].reshape(-1, 4)], 1)).double()
        want = float(1 - torch.nn.functional.cosine_similarity(s, t, dim=-1).mean())
        _log(f"cfg1 full size loss: kernel {loss_i:.7f} vs fp64 on the same features {want:.7f}")
        assert abs(loss_i - want) < 2e-5
        # two images of the batch against the oracle
        sd = seeded_visual_state(cfg, 0)
        ref = eva_ref.encode_pseudo_boxes(sd, cfg, images[:2].cpu(), [b[:, :4].cpu() for b in boxes[:2]])
        r, c = rel(s[:64], ref), one_minus_cos(s[:64], ref)
        _log(f"cfg1 full size student RoI features (2 of 64 images) vs oracle: rel-L2 {r:.3e}, max 1-cos {c:.2e}")
        assert r < 1.5e-2 and c < 2e-4


@pytest.mark.parametrize("share,mask", [(32, False), (48, True)])
def test_cfg1_full_size_step_with_cu_partition_equals_the_shared_pool(share, mask):
    """The single-GPU tower partition (CLIPSelf(partition_cus=R): the prefetched teacher pass launches its persistent GEMMs on 256 - R
    workgroups and the student's are capped at R, optionally on queues with hardware CU masks) changes WHERE tiles run, not what they hold:
    loss and every gradient bit equal the shared-pool schedule, and the grids really were the partition's."""
    from clipself_amd.training.clipself import CLIPSelf
    from clipself_amd.training.optim import FlatAdamW
    from clipself_amd.training.train import train_step
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    batch = tuple(t.cuda() for t in synthetic_batch(64, 32, 224, 224, seed=1234))
    nxt = tuple(t.cuda() for t in synthetic_batch(64, 32, 224, 224, seed=2211))

    def run(share, mask):
        student, teacher = _pair(cfg, 0)
        sops, tops = student.visual.engine.ops, teacher.visual.engine.ops
        seen = {"teacher": set(), "student": set()}
        for who, ops in (("teacher", tops), ("student", sops)):
            inner = ops.gemm_nt_ln_split if who == "teacher" else ops.gemm_nt
            def spy(*a, _inner=inner, _ops=ops, _who=who, **k):
                seen[_who].add(_ops.persistent_grid())
                return _inner(*a, **k)
            setattr(ops, "gemm_nt_ln_split" if who == "teacher" else "gemm_nt", spy)
        method = CLIPSelf(partition_cus=share, partition_mask=mask)
        a = _args(skip_scheduler=True)
        a.teacher_prefetch = True
        sstream = method.student_stream(sops)
        torch.cuda.synchronize()
        with (torch.cuda.stream(sstream) if sstream is not None else torch.cuda.stream(torch.cuda.current_stream())):
            train_step(student, method, nxt, FlatAdamW(student, lr=0.0, weight_decay=0.0), None, 0, teacher, a, next_batch=batch)
            assert method._pending is not None
            seen["student"].clear()                   # from here on the student runs beside the prefetched pass over `batch`
            out, _, _ = train_step(student, method, batch, FlatAdamW(student, lr=1e-5, weight_decay=0.1), None, 0, teacher, a, next_batch=nxt)
        torch.cuda.synchronize()
        return float(out["loss"].detach()), student.visual.engine.grad.clone(), seen, sops.num_compute_units()

    loss0, grad0, seen0, n = run(0, False)
    loss1, grad1, seen1, _ = run(share, mask)
    assert seen0 == {"teacher": {n}, "student": {n}}
    # (the very first teacher pass, over `nxt`, has nothing to run beside and is inline on the whole chip)
    assert seen1["teacher"] == {n, n - share} and seen1["student"] == {share}, seen1
    diff = int((grad0 != grad1).sum())
    _log(f"cfg1 full size step, CU partition {share} (masks {mask}) vs shared pool: loss {loss1:.7f} vs {loss0:.7f}, {diff} gradient elements differ")
    assert loss0 == loss1 and diff == 0


def _regionclip_batch(cfg, B, n_nouns=4764, max_boxes=20, seed=77):
    g = np.random.Generator(np.random.PCG64(seed))
    images, nb, _ = synthetic_batch(B, max_boxes, cfg.image_size, 32, seed=seed)
    labels = torch.from_numpy(g.integers(0, n_nouns, size=(B, max_boxes, 1)).astype(np.float32))
    valid = torch.from_numpy((g.random((B, max_boxes, 1)) < 0.7).astype(np.float32))
    valid[:, 0] = 1.0
    bx = torch.cat([nb[..., :4], labels, valid], dim=-1)
    nouns = torch.from_numpy(g.standard_normal((n_nouns, cfg.embed_dim)).astype(np.float32))
    return images, bx, nouns


def test_cfg4_l14_336_regionclip_real_config(monkeypatch):
    """BASELINE configs[4] in bf16: EVA02-CLIP-L-14-336 RegionCLIP (region-text) with the real noun-bank size (4764 x 768), <= 20 boxes
    per image: (a) 2 images against oracle/eva_ref.regionclip_loss (pinned on the reference's golden) -- loss and three gradients,
    with the federated column subset fixed on both sides; (b) the full per-GPU batch of 32 images: finite loss / gradients and
    the optimizer step."""
    from oracle import eva_ref
    from clipself_amd.open_clip.model import CustomCLIP
    from clipself_amd.training import region_clip as rc
    from clipself_amd.training.optim import FlatAdamW
    from clipself_amd.training.train import train_step
    cfg = get_tower_cfg("EVA02-CLIP-L-14-336")
    student = CustomCLIP(cfg, trainable=True)
    sd0 = seeded_visual_state(cfg, 3)
    student.visual.engine.load_state(sd0)
    student.lock_image_tower(unlocked_groups=cfg.layers)
    student.train()
    images, bx, nouns = _regionclip_batch(cfg, 2)
    labels = torch.cat([b[b[:, -1] > 0.5][:, 4].long() for b in bx])
    appeared = torch.unique(labels)
    extra = torch.from_numpy(np.setdiff1d(np.arange(4764), appeared.numpy())[: 100 - len(appeared)])
    appeared = torch.cat([appeared, extra])
    monkeypatch.setattr(rc, "get_fed_loss_inds", lambda gt, n, C: appeared.to(gt.device))
    method = rc.RegionCLIP(SimpleNamespace(), noun_embeddings=nouns)
    a = SimpleNamespace(extract_type="v2", contrast_weight=1.0)
    losses, bs, temp = method((images, bx), student, None, None, "cuda", None, False, a)
    total = sum(losses.values())
    total.backward()
    names = ["visual.blocks.0.attn.q_bias", "visual.blocks.12.mlp.w1.weight", "visual.blocks.23.mlp.w3.bias"]
    sd = {k: v.clone().requires_grad_(k in names) for k, v in sd0.items()}
    want = eva_ref.regionclip_loss(sd, cfg, images, bx, nouns, appeared=appeared)
    want.backward()
    lr_ = abs(float(total.detach()) - float(want.detach())) / float(want.detach())
    grs = {n: rel(dict(student.named_parameters())[n].grad, sd[n].grad) for n in names}
    _log(f"cfg4 L/14-336 RegionCLIP (2 images, 4764 nouns) vs oracle: loss {float(total.detach()):.5f} vs {float(want.detach()):.5f} (rel {lr_:.2e}); "
         + ", ".join(f"grad {n} rel {r:.2e}" for n, r in grs.items()))
    assert lr_ < 2e-3
    assert all(r < 6e-2 for r in grs.values()), grs
    # full per-GPU batch of the configuration
    monkeypatch.undo()
    images, bx, nouns = _regionclip_batch(cfg, 32, seed=78)
    method = rc.RegionCLIP(SimpleNamespace(), noun_embeddings=nouns)
    opt = FlatAdamW(student, lr=1e-5, weight_decay=0.1)
    args = _args(skip_scheduler=True, contrast_weight=1.0)
    before = student.visual.engine.master.clone()
    out, bs, _ = train_step(student, method, (images, bx), opt, None, 0, None, args)
    torch.cuda.synchronize()
    g = student.visual.engine.grad
    _log(f"cfg4 L/14-336 RegionCLIP full batch (32 images x <=20 boxes): loss {float(out['loss']):.5f}, |grad| {float(g.double().norm()):.4e}")
    assert bs == 32 and torch.isfinite(out["loss"]).item() and torch.isfinite(g).all().item() and float(g.abs().sum()) > 0
    after = student.visual.engine.master
    assert torch.isfinite(after).all().item() and not torch.equal(after, before)


def test_recipe_shape_1024px_student_4097_tokens():
    """SURVEY section 8(f) N1 at the shape the shipped recipe trains (scripts/train_clipself_coco_image_patches_eva_vitb16.sh:
    --det-image-size 1024, --batch-size 2): the B/16 student on 1024^2 images = 64 x 64 + 1 = 4097 tokens (bicubic pos-embed rescale,
    regenerated RoPE tables, 19 key chunks in the attention kernels).  One image against the CPU oracle (forward), then one full
    recipe-shaped step (2 images, 20 grid boxes each) with finite loss / gradients, timed."""
    import time
    from oracle import eva_ref
    from clipself_amd.training.clipself import CLIPSelf
    from clipself_amd.training.optim import FlatAdamW
    from clipself_amd.training.train import train_step
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    student, teacher = _pair(cfg, 0)
    sd = seeded_visual_state(cfg, 0)
    images, boxes, _ = synthetic_batch(1, 6, 1024, 224, seed=17)
    rois = [b[:, :4] for b in boxes]
    with torch.no_grad():
        want = eva_ref.encode_pseudo_boxes(sd, cfg, images, rois)
        got = student.encode_pseudo_boxes(images.cuda(), [r.cuda() for r in rois])
    r, c = rel(got, want), one_minus_cos(got, want)
    _log(f"N1 recipe shape: B/16 student at 1024^2 (4097 tokens) RoI features vs oracle: rel-L2 {r:.3e}, max 1-cos {c:.2e}")
    assert r < 1e-2 and c < 5e-5                           # measured 4.0e-3 / 7.9e-6
    # round 6: the backward at 19 key chunks against autograd of the fp32 oracle -- gradient of sum(roi feats * w) w.r.t.
    # one early and one late parameter, as test_non_native_grid_multichunk_attention_matches_oracle does at 448^2
    names = ("visual.blocks.0.attn.q_bias", "visual.blocks.1.attn.v_proj.weight", "visual.blocks.10.mlp.w3.bias")
    w = torch.randn(want.shape, generator=torch.Generator().manual_seed(0))
    (student.encode_pseudo_boxes(images.cuda(), [r.cuda() for r in rois]) * w.cuda()).sum().backward()
    ref = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
    (eva_ref.encode_pseudo_boxes(ref, cfg, images, rois) * w).sum().backward()
    for n in names:
        rg = rel(dict(student.named_parameters())[n].grad, ref[n].grad)
        _log(f"N1 recipe shape: gradient at 4097 tokens vs oracle autograd, {n}: rel-L2 {rg:.3e}")
        assert rg < 1.2e-2, (n, rg)                       # measured 6.0e-3 / 5.1e-3 / 1.8e-3 (profiles/r06_parity.md)
    del ref                                                # train_step below starts with optimizer.zero_grad()
    batch = tuple(t.cuda() for t in synthetic_batch(2, 20, 1024, 224, seed=18))
    opt = FlatAdamW(student, lr=1e-5, weight_decay=0.1)
    method, args = CLIPSelf(), _args(skip_scheduler=True)
    out, _, _ = train_step(student, method, batch, opt, None, 0, teacher, args)          # warm-up (allocations, table builds)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(3):
        out, _, _ = train_step(student, method, batch, opt, None, step + 1, teacher, args)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    g = student.visual.engine.grad
    _log(f"N1 recipe shape: step of 2 images x 1024^2 + 40 teacher crops: {ms:.1f} ms/step ({2e3 / ms:.1f} images/s), loss {float(out['loss']):.5f}")
    assert torch.isfinite(out["loss"]).item() and torch.isfinite(g).all().item() and float(g.abs().sum()) > 0


def test_cfg4_l14_336_regionclip_fp8_forward(monkeypatch):
    """BASELINE configs[4] with "fp8 MFMA weights": the same L/14-336 RegionCLIP step with precision 'amp_fp8' (forward linears on e4m3
    operands through the block-scaled fp8 MFMA, EvaEngine.enable_fp8_forward).  (a) the HIP fp8 schedule equals the CPU statement of the
    same schedule (oracle/ops_ref.py, torch.float8_e4m3fn) on a tiny tower; (b) at the real configuration the fp8 loss stays within
    3e-2 of the bf16 loss and of the fp32 oracle, gradients are finite, and both step times are logged."""
    import time
    from oracle import eva_ref
    from oracle.ops_ref import RefOps
    from clipself_amd.config import tiny_cfg
    from clipself_amd.open_clip.model import CustomCLIP
    from clipself_amd.training import region_clip as rc
    from clipself_amd.training.optim import FlatAdamW
    from clipself_amd.training.train import train_step
    from test_regionclip_cpu import regionclip_inputs
    # (a) kernel path == reference path of the same fp8 schedule
    tcfg = tiny_cfg()
    images, bx, nouns = regionclip_inputs(tcfg)
    rois = [b[b[:, -1] > 0.5][:, :4] for b in bx]
    feats = {}
    for tag, ops, dev in (("ref", RefOps(), "cpu"), ("hip", None, "cuda")):
        m = CustomCLIP(tcfg, ops=ops, trainable=True)
        m.visual.engine.load_state(seeded_visual_state(tcfg, 2))
        m.visual.engine.enable_fp8_forward()
        with torch.no_grad():
            feats[tag] = m.encode_pseudo_boxes(images.to(dev), [r.to(dev) for r in rois], normalize=True)
    r = rel(feats["hip"], feats["ref"])
    _log(f"fp8 forward schedule, tiny tower: HIP kernels vs CPU reference ops: rel-L2 {r:.3e}")
    assert r < 6e-2          # two implementations of the same e4m3 schedule: one bf16 ulp upstream moves an e4m3 value by a 2^-3 step
    # (b) the real configuration
    cfg = get_tower_cfg("EVA02-CLIP-L-14-336")
    sd0 = seeded_visual_state(cfg, 3)
    images, bx, nouns = _regionclip_batch(cfg, 2)
    labels = torch.cat([b[b[:, -1] > 0.5][:, 4].long() for b in bx])
    appeared = torch.unique(labels)
    appeared = torch.cat([appeared, torch.from_numpy(np.setdiff1d(np.arange(4764), appeared.numpy())[: 100 - len(appeared)])])
    monkeypatch.setattr(rc, "get_fed_loss_inds", lambda gt, n, C: appeared.to(gt.device))
    a = SimpleNamespace(extract_type="v2", contrast_weight=1.0)
    with torch.no_grad():
        want = float(eva_ref.regionclip_loss(dict(sd0), cfg, images, bx, nouns, appeared=appeared))
    losses, models = {}, {}
    for tag in ("bf16", "fp8", "fp8+dgrad"):
        m = CustomCLIP(cfg, trainable=True)
        m.visual.engine.load_state(sd0)
        m.lock_image_tower(unlocked_groups=cfg.layers)
        m.train()
        if tag != "bf16":
            m.visual.engine.enable_fp8_forward(dgrad=tag == "fp8+dgrad")      # "amp_fp8_dgrad": e4m3 operands in the dgrad GEMMs as well
        out, _, _ = rc.RegionCLIP(SimpleNamespace(), noun_embeddings=nouns)((images, bx), m, None, None, "cuda", None, False, a)
        total = sum(out.values())
        total.backward()
        assert torch.isfinite(m.visual.engine.grad).all().item()
        losses[tag], models[tag] = float(total.detach()), m
    _log(f"cfg4 L/14-336 RegionCLIP fp8 forward (2 images): loss fp8 {losses['fp8']:.4f} bf16 {losses['bf16']:.4f} oracle {want:.4f}; "
         f"grad rel fp8-vs-bf16 {rel(models['fp8'].visual.engine.grad, models['bf16'].visual.engine.grad):.3e}")
    assert abs(losses["fp8"] - losses["bf16"]) / losses["bf16"] < 3e-2 and abs(losses["fp8"] - want) / want < 3e-2
    rd = rel(models["fp8+dgrad"].visual.engine.grad, models["fp8"].visual.engine.grad)
    _log(f"cfg4 L/14-336 RegionCLIP fp8 forward + dgrad (2 images): loss {losses['fp8+dgrad']:.4f} (same forward); "
         f"grad rel vs fp8-forward {rd:.3e}, vs bf16 {rel(models['fp8+dgrad'].visual.engine.grad, models['bf16'].visual.engine.grad):.3e}")
    assert losses["fp8+dgrad"] == losses["fp8"] and 1e-4 < rd < 0.3
    monkeypatch.undo()
    # step time at the full per-GPU batch (32 images x <= 20 boxes), bf16 vs fp8 forward
    images, bx, nouns = _regionclip_batch(cfg, 32, seed=78)
    batch = (images.cuda(), bx.cuda())
    for tag in ("bf16", "fp8", "fp8+dgrad"):
        m = models[tag]
        method = rc.RegionCLIP(SimpleNamespace(), noun_embeddings=nouns)
        opt = FlatAdamW(m, lr=1e-5, weight_decay=0.1)
        args = _args(skip_scheduler=True, contrast_weight=1.0)
        train_step(m, method, batch, opt, None, 0, None, args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for step in range(3):
            out, _, _ = train_step(m, method, batch, opt, None, step + 1, None, args)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        _log(f"cfg4 L/14-336 RegionCLIP step, 32 images x <=20 boxes, {tag} forward: {ms:.1f} ms/step ({32e3 / ms:.1f} images/s), loss {float(out['loss']):.3f}")
        assert torch.isfinite(out["loss"]).item()


def test_multiscale_step_matches_the_oracle_at_the_drawn_size(monkeypatch):
    """--multiscale (src/training/clipself.py:17-27): an 896^2 student batch is resized to one of [336, 448, 672, 896] per step
    (cs_resize_bilinear_f32 = F.interpolate bilinear) before the dense forward.  With the draw pinned to 672 (42 x 42 + 1 = 1765 tokens) the
    step's loss must equal the oracle's loss on the identically resized images."""
    import random
    import torch.nn.functional as F
    from oracle import eva_ref
    from clipself_amd.training.clipself import CLIPSelf
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    student, teacher = _pair(cfg, 0)
    batch = synthetic_batch(1, 5, 896, 224, seed=31)
    monkeypatch.setattr(random, "choice", lambda seq: 672)
    out, bs, _ = CLIPSelf()(tuple(t.cuda() for t in batch), student, teacher, None, "cuda", None, False, _args(multiscale=True))
    images, boxes, crops = batch
    sd = seeded_visual_state(cfg, 0)
    with torch.no_grad():
        small = F.interpolate(images, size=(672, 672), mode="bilinear")
        want = float(eva_ref.clipself_loss(dict(sd), dict(sd), cfg, (small, boxes, crops))[0])
    got = float(out["loss_cosine"].detach())
    _log(f"N1 --multiscale: 896^2 -> 672^2 (1765 tokens) step loss {got:.6f} vs oracle {want:.6f} (rel {abs(got - want) / want:.2e})")
    assert abs(got - want) / want < 1e-3


def _loss_from_features(student, teacher, batch):
    """fp64 cosine loss over the valid boxes, from the towers' own features (clipself.py:29-47)."""
    images, boxes, crops = batch
    valid = boxes[..., 4] > 0.5
    rois = [b[v][:, :4] for b, v in zip(boxes, valid)]
    with torch.no_grad():
        t = teacher.encode_image(crops[valid]).double()
        s = student.encode_pseudo_boxes(images, rois).double()
    return float(1 - torch.nn.functional.cosine_similarity(s, t, dim=-1).mean()), int(valid.sum()), s, t


def test_cfg2_region_proposals_full_batch_with_ragged_validity():
    """BASELINE configs[2] per-GPU shape (B/16, 64 images x 20 proposal slots, ~70 % valid, every image keeps >= 1 box;
    data.py:41,84-132 + clipself.py:29-36): the step's loss equals the fp64 loss over exactly the valid boxes, gradients are finite and
    non-zero, and the features of the first two images' valid boxes match the oracle."""
    from oracle import eva_ref
    from clipself_amd.training.clipself import CLIPSelf
    from clipself_amd.training.optim import FlatAdamW
    from clipself_amd.training.train import train_step
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    batch = synthetic_batch(64, 20, 224, 224, seed=4321, valid_prob=0.7)
    dev = tuple(t.cuda() for t in batch)
    student, teacher = _pair(cfg, 0)
    want, n_valid, s, _ = _loss_from_features(student, teacher, dev)
    assert 64 <= n_valid < 64 * 20
    opt = FlatAdamW(student, lr=1e-5, weight_decay=0.1)
    out, bs, _ = train_step(student, CLIPSelf(), dev, opt, None, 0, teacher, _args(skip_scheduler=True))
    loss = float(out["loss"].detach())
    g = student.visual.engine.grad
    _log(f"cfg2 proposals full batch (64 images x 20 slots, {n_valid} valid): loss {loss:.6f} vs fp64 on the valid boxes {want:.6f}, "
         f"|grad| {float(g.double().norm()):.4e}")
    assert bs == 64 and abs(loss - want) < 2e-5
    assert torch.isfinite(g).all() and float(g.abs().sum()) > 0
    images, boxes, _ = batch
    keep = [b[b[:, 4] > 0.5][:, :4] for b in boxes[:2]]
    ref = eva_ref.encode_pseudo_boxes(seeded_visual_state(cfg, 0), cfg, images[:2], keep)
    got = s[: ref.shape[0]]
    r, c = rel(got, ref), one_minus_cos(got, ref)
    _log(f"cfg2 proposals: RoI features of the first 2 images ({ref.shape[0]} valid boxes) vs oracle: rel-L2 {r:.3e}, max 1-cos {c:.2e}")
    assert r < 1.5e-2 and c < 2e-4


def test_cfg3_l14_336_clipself_full_batch():
    """BASELINE configs[3] per-GPU shape (EVA02-CLIP-L-14-336, 16 images x 32 crops at 336^2: 512 crops x 577 tokens = 295 424 GEMM rows,
    24 layers, hidden 2730 padded to 2752): the step's loss equals the fp64 loss from the towers' own features, six teacher crops sampled
    over the batch match the oracle, the gradient is finite, and a second step lowers nothing to NaN."""
    from oracle import eva_ref
    from clipself_amd.training.clipself import CLIPSelf
    from clipself_amd.training.optim import FlatAdamW
    from clipself_amd.training.train import train_step
    cfg = get_tower_cfg("EVA02-CLIP-L-14-336")
    batch = synthetic_batch(16, 32, 336, 336, seed=99)
    dev = tuple(t.cuda() for t in batch)
    student, teacher = _pair(cfg, 0)
    want, n_valid, _, t = _loss_from_features(student, teacher, dev)
    assert n_valid == 512
    opt = FlatAdamW(student, lr=1e-5, weight_decay=0.1)
    method = CLIPSelf()
    out, bs, _ = train_step(student, method, dev, opt, None, 0, teacher, _args(skip_scheduler=True))
    loss = float(out["loss"].detach())
    g = student.visual.engine.grad
    gn = float(g.double().norm())
    assert torch.isfinite(g).all() and gn > 0
    out2, _, _ = train_step(student, method, dev, opt, None, 1, teacher, _args(skip_scheduler=True))
    loss2 = float(out2["loss"].detach())
    _log(f"cfg3 L/14-336 CLIPSelf full batch (16 images x 32 crops): loss {loss:.6f} vs fp64 on the same features {want:.6f}; "
         f"|grad| {gn:.4e}; loss after one AdamW step {loss2:.6f}")
    assert bs == 16 and abs(loss - want) < 2e-5
    assert np.isfinite(loss2) and loss2 < loss + 1e-3
    idx = torch.tensor([0, 1, 100, 255, 300, 511])
    crops = batch[2].flatten(0, 1)[idx]
    ref = eva_ref.encode_image(seeded_visual_state(cfg, 0), cfg, crops)
    got = t[idx.cuda()]
    r, c = rel(got, ref), one_minus_cos(got, ref)
    _log(f"cfg3 L/14-336 teacher, 6 crops sampled from the 512-crop pass vs oracle: rel-L2 {r:.3e}, max 1-cos {c:.2e}")
    assert r < 1.9e-2 and c < 1e-4


def test_cfg1_full_size_step_is_bit_reproducible():
    """BASELINE configs[1] at full size, twice from the same state and batch: identical loss bits and identical gradient bits.  Round 2's
    backward added RoIAlign contributions and bias column sums with float atomics (2.8e-4 relative run-to-run differences); the RoIAlign
    backward is now a gather in box order, the bias sums come out of the LayerNorm backwards / row-block partials in a fixed order, and
    the split-K weight gradients were already combined in a fixed order."""
    from clipself_amd.training.clipself import CLIPSelf
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    student, teacher = _pair(cfg, 0)
    batch = tuple(t.cuda() for t in synthetic_batch(64, 32, 224, 224, seed=4321))
    args = _args()
    runs = []
    for _ in range(2):
        student.visual.engine.zero_grad()
        for p in student.parameters():
            p.grad = None
        losses, _, _ = CLIPSelf()(batch, student, teacher, None, "cuda", None, False, args)
        total = sum(losses.values())
        total.backward()
        torch.cuda.synchronize()
        runs.append((total.detach().clone(), student.visual.engine.grad.clone()))
    assert torch.equal(runs[0][0], runs[1][0])
    diff = int((runs[0][1] != runs[1][1]).sum())
    _log(f"cfg1 full-size step twice: loss bits equal, {diff} of {runs[0][1].numel()} gradient elements differ, |grad| {float(runs[0][1].double().norm()):.4e}")
    assert diff == 0 and float(runs[0][1].abs().sum()) > 0
