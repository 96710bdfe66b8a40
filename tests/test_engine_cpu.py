"""CPU: the step engine's op schedule (clipself_amd/engine.py) -- forward decomposition and the hand-written
backward chain -- driven by the per-kernel references (oracle/ops_ref.py, injected here as test infrastructure),
checked against the monolithic autograd oracle (oracle/eva_ref.py) and the reference-derived golden vectors."""
import json

import numpy as np
import pytest
import torch

from clipself_amd.config import tiny_cfg
from clipself_amd.engine import EvaEngine
from clipself_amd.init import seeded_visual_state, synthetic_batch
from oracle import eva_ref
from oracle.ops_ref import RefOps


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _engine(cfg, seed, trainable):
    eng = EvaEngine(cfg, RefOps(), trainable=trainable)
    eng.load_state(seeded_visual_state(cfg, seed))
    if trainable:
        eng.set_trainable_blocks(cfg.layers)
    return eng


def _rois(boxes):
    rows = [torch.cat([torch.full((len(b), 1), float(i)), b[:, :4]], dim=1) for i, b in enumerate(boxes)]
    return torch.cat(rows)


@pytest.fixture(scope="module")
def gold(golden_dir):
    g = np.load(golden_dir / "tiny_step.npz")
    return g, json.loads(str(g["recipe"]))


def test_forward_matches_bf16_oracle_and_reference(gold):
    g, rec = gold
    cfg = tiny_cfg()
    sd = seeded_visual_state(cfg, rec["seed_w"])
    images, boxes, crops = synthetic_batch(rec["batch"], rec["boxes"], cfg.image_size, cfg.image_size, seed=rec["seed_b"])
    eng = _engine(cfg, rec["seed_w"], False)
    teacher = eng.encode_image(crops.flatten(0, 1), chunk=4)
    dense, grid = eng.encode_dense(images)
    pooled = eng.roi_pool(dense, _rois(boxes), grid)
    with torch.no_grad():
        t_bf = eva_ref.encode_image(sd, cfg, crops.flatten(0, 1), emulate_bf16=True)
        s_bf = eva_ref.encode_pseudo_boxes(sd, cfg, images, [b[:, :4] for b in boxes], emulate_bf16=True)
    # Two independent bf16-operand implementations differ by the same ~5e-3..1e-2 relative L2 as either differs
    # from fp32 (operand rounding floor, 2^-9 per product); what is tight is the *direction* of each feature
    # vector (the only thing the loss sees after F.normalize): 1 - cos <= 1e-3 by a wide margin.
    def one_minus_cos(a, b):
        return float((1 - torch.nn.functional.cosine_similarity(torch.as_tensor(a).double(), torch.as_tensor(b).double(), dim=-1)).max())
    assert rel(teacher, t_bf) < 1.5e-2 and rel(pooled, s_bf) < 1.5e-2
    assert rel(teacher, g["teacher"]) < 2e-2
    assert rel(pooled, g["student_roi"]) < 2e-2
    assert rel(dense[:, 1:], g["dense"]) < 2e-2
    assert one_minus_cos(teacher, g["teacher"]) < 2e-4
    assert one_minus_cos(pooled, g["student_roi"]) < 2e-4


def test_cls_only_last_teacher_block_is_the_same_function(gold):
    """encode_image() runs the last block for the CLS query only (engine.cls_only_last_block); the full-token last block
    must give the same embedding -- every skipped row is dead in forward_features() (eva_vit_model.py:505-519)."""
    g, rec = gold
    cfg = tiny_cfg()
    _, _, crops = synthetic_batch(rec["batch"], rec["boxes"], cfg.image_size, cfg.image_size, seed=rec["seed_b"])
    eng = _engine(cfg, rec["seed_w"], False)
    assert eng.cls_only_last_block and eng.fold_sub_ln
    eng.fold_sub_ln = False                       # isolate the schedule change: identical arithmetic per row
    fast = eng.encode_image(crops.flatten(0, 1), chunk=3)
    eng.cls_only_last_block = False
    full = eng.encode_image(crops.flatten(0, 1), chunk=3)
    assert rel(fast, full) < 1e-6
    assert rel(fast, g["teacher"]) < 2e-2


def test_folded_sub_layernorms_match_the_unfolded_teacher(gold):
    """Frozen towers fold inner_attn_ln / ffn_ln into proj / w3 (gamma into the bf16 weight, statistics from the producers'
    epilogues).  Same function up to bf16 operand rounding: compared with the unfolded schedule and with the reference golden."""
    g, rec = gold
    cfg = tiny_cfg()
    _, _, crops = synthetic_batch(rec["batch"], rec["boxes"], cfg.image_size, cfg.image_size, seed=rec["seed_b"])
    eng = _engine(cfg, rec["seed_w"], False)
    eng.cls_only_last_block = False               # every block through the folded path
    assert eng.fold_block_ln                      # norm1 / norm2 folded as well (bf16 copy + statistics from the residual GEMMs)
    folded = eng.encode_image(crops.flatten(0, 1), chunk=5)
    eng.fold_block_ln = False                     # only the two sub-LayerNorms folded
    sub_only = eng.encode_image(crops.flatten(0, 1), chunk=5)
    eng.fold_sub_ln = False
    plain = eng.encode_image(crops.flatten(0, 1), chunk=5)
    for other in (sub_only, folded):
        cos = torch.nn.functional.cosine_similarity(other.double(), plain.double(), dim=-1)
        assert rel(other, plain) < 1e-2 and float((1 - cos).max()) < 1e-4
    assert rel(folded, g["teacher"]) < 2e-2
    cosg = torch.nn.functional.cosine_similarity(folded.double(), torch.as_tensor(g["teacher"]).double(), dim=-1)
    assert float((1 - cosg).max()) < 2e-4
    # the trainable tower never folds (its backward needs the normalised activations)
    assert not _engine(cfg, rec["seed_w"], True).fold_sub_ln


def test_backward_chain_is_the_gradient(gold):
    g, rec = gold
    cfg = tiny_cfg()
    images, boxes, crops = synthetic_batch(rec["batch"], rec["boxes"], cfg.image_size, cfg.image_size, seed=rec["seed_b"])
    eng = _engine(cfg, rec["seed_w"], True)
    ops = eng.ops
    rois = _rois(boxes)
    dense, grid = eng.encode_dense(images, need_grad=True)
    pooled = eng.roi_pool(dense, rois, grid)
    teacher = torch.from_numpy(g["teacher"])
    K, E = pooled.shape
    stats, loss, dpool = torch.empty(K, 3), torch.empty(1), torch.empty(K, E)
    ops.cosine_loss_fwd(pooled, teacher, stats, loss, 1.0)
    ops.cosine_loss_bwd(pooled, teacher, stats, dpool, 1.0, 1.0)
    eng.zero_grad()
    fired = []
    eng.grad_ready_hook = fired.append
    eng.backward_dense(eng.roi_pool_backward(dpool, rois, images.shape[0], dense.shape[1], grid))
    assert fired == list(range(cfg.layers - 1, -1, -1))
    assert abs(float(loss) - g["losses"][0]) < 5e-3
    none = {str(n) for n in g["grad_none"]}
    worst = 0.0
    for name in eng.trainable_names():
        got = eng.g[name]
        if name in none:
            assert float(got.abs().max()) == 0.0, f"{name} must not receive a gradient"
            continue
        r = rel(got, g["grad/" + name])
        worst = max(worst, r)
        assert r < 6e-2, f"{name}: rel {r:.3e}"
    assert float(eng.g["visual.blocks.0.attn._k_bias_zero"].abs().max()) == 0.0
    print("worst grad rel", worst)


def test_three_steps_track_reference_losses(gold):
    g, rec = gold
    cfg = tiny_cfg()
    eng = _engine(cfg, rec["seed_w"], True)
    teacher_eng = _engine(cfg, rec["seed_w"], False)
    ops = eng.ops
    losses = []
    for step in range(rec["steps"]):
        images, boxes, crops = synthetic_batch(rec["batch"], rec["boxes"], cfg.image_size, cfg.image_size, seed=rec["seed_b"] + step)
        rois = _rois(boxes)
        teacher = teacher_eng.encode_image(crops.flatten(0, 1))
        dense, grid = eng.encode_dense(images, need_grad=True)
        pooled = eng.roi_pool(dense, rois, grid)
        K, E = pooled.shape
        stats, loss, dpool = torch.empty(K, 3), torch.empty(1), torch.empty(K, E)
        ops.cosine_loss_fwd(pooled, teacher, stats, loss, 1.0)
        ops.cosine_loss_bwd(pooled, teacher, stats, dpool, 1.0, 1.0)
        eng.zero_grad()
        eng.backward_dense(eng.roi_pool_backward(dpool, rois, images.shape[0], dense.shape[1], grid))
        lr = eva_ref.cosine_lr_value(step, rec["lr"], rec["warmup"], rec["total"])
        eng.adamw_step(step + 1, lr, rec["wd"])
        losses.append(float(loss))
    assert np.allclose(losses, g["losses"], atol=1e-2), (losses, g["losses"])
    # frozen tensors untouched, skipped tensors untouched (no decay either)
    sd0 = seeded_visual_state(cfg, rec["seed_w"])
    for n in ("visual.head.weight", "visual.pos_embed", f"visual.blocks.{cfg.layers - 1}.attn.q_proj.weight"):
        assert torch.equal(eng.p[n], sd0[n].reshape(eng.p[n].shape)), n
    r = rel(eng.p["visual.blocks.0.mlp.w1.weight"], g["final/visual.blocks.0.mlp.w1.weight"])
    assert r < 2e-2, r


def test_split_residual_stream_is_lossless_and_changes_nothing(gold):
    """The frozen tower keeps its fp32 residual stream as two 16-bit planes of bits(x) + 0x8000 between the residual GEMMs
    (cs_gemm_nt_ln_split; mirror: RefOps.split_planes / join_planes).  The planes rebuild every fp32 bit pattern, the upper plane is x rounded
    to bf16 (halves away from zero), and the teacher's features are those of the fp32-stream schedule."""
    x = torch.randn(4096, 64) * 10
    x[0, :4] = torch.tensor([0.0, -0.0, float("inf"), -float("inf")])
    tie = (torch.tensor([1.0]).view(torch.int32) | 0x8000).view(torch.float32)[0]           # low 16 bits exactly 0x8000
    x[1, :2] = torch.stack([tie, -tie])
    x[2, 0] = torch.finfo(torch.float32).tiny                                               # smallest normal
    hi, lo = RefOps.split_planes(x)
    assert hi.dtype == torch.bfloat16 and lo.dtype == torch.int16
    assert torch.equal(RefOps.join_planes(hi, lo).view(torch.int32), x.view(torch.int32))
    rne = x.to(torch.bfloat16)
    differs = hi.view(torch.int16) != rne.view(torch.int16)
    low16 = x.view(torch.int32) & 0xFFFF
    assert bool((low16[differs] == 0x8000).all())                                           # only exact ties round differently from RNE
    assert float(hi[1, 0]) > float(tie) and float(hi[1, 1]) < -float(tie)                   # ... and they round away from zero

    g, rec = gold
    cfg = tiny_cfg()
    _, _, crops = synthetic_batch(rec["batch"], rec["boxes"], cfg.image_size, cfg.image_size, seed=rec["seed_b"])
    eng = _engine(cfg, rec["seed_w"], False)
    assert eng.split_stream and eng.fold_block_ln
    for cls_only in (True, False):
        eng.cls_only_last_block = cls_only
        eng.split_stream = True
        split = eng.encode_image(crops.flatten(0, 1), chunk=5)
        eng.split_stream = False
        fp32 = eng.encode_image(crops.flatten(0, 1), chunk=5)
        # the two schedules differ only where an exact tie makes the upper plane round away from zero instead of to even (2^-16 of the
        # values): at width 64 one such flip of a bf16 operand is visible, far below the bf16 noise of the path itself
        cos = torch.nn.functional.cosine_similarity(split.double(), fp32.double(), dim=-1)
        assert rel(split, fp32) < 3e-3 and float((1 - cos).max()) < 1e-5, (rel(split, fp32), float((1 - cos).max()))
    assert not _engine(cfg, rec["seed_w"], True).split_stream                               # the trainable tower keeps fp32 + autograd saves
    # one residual GEMM, both forms, same operands: the fp32 stream is bit-identical and so are the statistics
    ops = RefOps()
    M, N, K = 37, 64, 128
    gen = torch.Generator().manual_seed(5)
    A, B = torch.randn(M, K, generator=gen).to(torch.bfloat16), (torch.randn(N, K, generator=gen) * 0.1).to(torch.bfloat16)
    bias, cs = torch.randn(N, generator=gen), torch.randn(N, generator=gen)
    mean, rstd = torch.randn(M, generator=gen) * 0.1, torch.rand(M, generator=gen) + 0.5
    x0 = torch.randn(M, N, generator=gen) * 3
    xa, pa, xb = x0.clone(), torch.zeros(1, M, 2), torch.zeros(M, N, dtype=torch.bfloat16)
    ops.gemm_nt_ln(A, B, xa, bias=bias, extra=xa, ln_mean=mean, ln_rstd=rstd, ln_colsum=cs, stats_part=pa, xb_out=xb, epi=6)
    hi, lo, pb = torch.zeros(M, N, dtype=torch.bfloat16), torch.zeros(M, N, dtype=torch.int16), torch.zeros(1, M, 2)
    ops.gemm_nt_ln_split(A, B, hi, lo, bias, mean, rstd, cs, x_in=x0, stats_part=pb)
    assert torch.equal(RefOps.join_planes(hi, lo).view(torch.int32), xa.view(torch.int32)) and torch.equal(pa, pb)
    out = torch.zeros(M, N)
    ops.gemm_nt_ln_split(A, B, hi, lo, bias, mean, rstd, cs, x_out=out)
    xa2 = xa.clone()
    ops.gemm_nt_ln(A, B, xa2, bias=bias, extra=xa2, ln_mean=mean, ln_rstd=rstd, ln_colsum=cs, epi=6)
    assert torch.equal(out.view(torch.int32), xa2.view(torch.int32))


def test_kernel_rounding_oracles_track_the_schedule_twin_and_the_reference(gold):
    """oracle/eva_ref.py's bf16 emulation at the kernels' own rounding points (`encode_image_frozen_schedule`, `emulate_bf16="kernel"`) is an
    independent restatement of what the HIP schedules compute: on the 2-block tiny tower it agrees with the engine driven by the
    per-kernel references to <= 2e-3 (the generic `emulate_bf16=True` points sit 4e-3..9e-3 away -- where bf16 rounds is most of the
    difference between two bf16 implementations), and it stays as close to the reference's fp32 golden as any bf16 evaluation."""
    g, rec = gold
    cfg = tiny_cfg()
    sd = seeded_visual_state(cfg, rec["seed_w"])
    images, boxes, crops = synthetic_batch(rec["batch"], rec["boxes"], cfg.image_size, cfg.image_size, seed=rec["seed_b"])
    flat = crops.flatten(0, 1)
    frozen = _engine(cfg, rec["seed_w"], False).encode_image(flat, chunk=4)
    train_eng = _engine(cfg, rec["seed_w"], True)
    dense, grid = train_eng.encode_dense(images, need_grad=True)          # the training forward (activations kept)
    pooled = train_eng.roi_pool(dense, _rois(boxes), grid)
    with torch.no_grad():
        t_k = eva_ref.encode_image_frozen_schedule(sd, cfg, flat)
        t_g = eva_ref.encode_image(sd, cfg, flat, emulate_bf16=True)
        s_k = eva_ref.encode_pseudo_boxes(sd, cfg, images, [b[:, :4] for b in boxes], emulate_bf16="kernel")
    assert rel(frozen, t_k) < 2e-3 < rel(frozen, t_g)
    assert rel(pooled, s_k) < 2e-3
    assert rel(t_k, g["teacher"]) < 2e-2 and rel(s_k, g["student_roi"]) < 2e-2
    # the split stream's hi plane: round to nearest, halves away from zero; exact on bf16-representable values
    x = torch.tensor([1.0, -1.0, 1.00390625, -1.00390625, 1.001953125, 1.005859375, 65280.0])      # ties: 1 + 2^-8 lies halfway to 1 + 2^-7
    want = torch.tensor([1.0, -1.0, 1.0078125, -1.0078125, 1.0, 1.0078125, 65280.0])
    assert torch.equal(eva_ref._plane_round(x), want)
    assert float(x[2:3].to(torch.bfloat16)) == 1.0                                                    # RNE sends the tie to the even neighbour
