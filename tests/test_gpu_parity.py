"""`-m gpu`: the HIP path against the bf16-EMULATING oracles (oracle/eva_ref.py) -- the comparison the north star's "within 1e-3 on bf16
normalised features and loss" is quoted against.  Three facts, each measured here and written to gpurun_out/parity_metrics.txt
(tracked copy: profiles/r03_parity.md):

  1. Where bf16 rounds decides ~5e-3 of the result: the oracle has to round where the kernels round.  `emulate_bf16="kernel"` and
     `encode_image_frozen_schedule` restate the kernels' rounding points (un-normalised P in bf16, folded LayerNorms on the split stream,
     fused SiLU*mul) independently of the engine's code; `emulate_bf16=True` keeps the generic points.
  2. A 12-block tower with bf16 storage is chaotic at the 5e-3..1e-2 level: the SAME bf16-emulating oracle evaluated with fp64 instead of
     fp32 accumulation -- a 1e-7 perturbation of every sum -- moves the features by 8e-3 (each flipped rounding is a 2^-8 step that the
     following blocks amplify).  End-to-end distances between any two bf16 implementations therefore sit at that level; they are logged and
     bounded here, but they cannot separate kernel error from rounding chaos.
  3. What does separate them: ONE block at a time on identical inputs ("teacher forcing").  Each HIP block is fed the oracle's own fp32
     stream and compared with the oracle's next stream -- frozen schedule (all LayerNorms folded, split stream) and training schedule --
     at <= 2e-3 on the stream and <= 3e-3 on the block's update (measured: <= 1.2e-3 / 1.5e-3, typically 3e-4 / 6e-4); the 2-block
     tiny tower (no chaos yet) end to end at <= 2e-3.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clipself_amd.config import get_tower_cfg, tiny_cfg          # noqa: E402
from clipself_amd.init import seeded_visual_state, synthetic_batch  # noqa: E402
from test_gpu_step import _pair, one_minus_cos, rel               # noqa: E402

F32, BF16 = torch.float32, torch.bfloat16


def _log(msg):
    from pathlib import Path
    p = Path(__file__).resolve().parent.parent / "gpurun_out"
    p.mkdir(exist_ok=True)
    with open(p / "parity_metrics.txt", "a") as f:
        f.write(msg + "\n")


def _nrm(t):
    return torch.nn.functional.normalize(torch.as_tensor(t).detach().double().cpu(), dim=-1)


def test_tiny_tower_end_to_end_against_the_kernel_rounding_oracles():
    """2-block tower (no rounding chaos yet): teacher features against the frozen-schedule oracle, student RoI features and loss against
    the training-schedule oracle, at the north star's tolerance scale."""
    from oracle import eva_ref
    cfg = tiny_cfg()
    sd = seeded_visual_state(cfg, 3)
    student, teacher = _pair(cfg, 3)
    images, boxes, crops = synthetic_batch(4, 5, cfg.image_size, cfg.image_size, seed=11)
    flat = crops.flatten(0, 1)
    rois = [b[:, :4] for b in boxes]
    # the student runs its TRAINING forward (grad enabled: W1|W2 output stored in bf16, then SiLU*mul -- what the backward differentiates)
    s = student.encode_pseudo_boxes(images.cuda(), [r.cuda() for r in rois]).detach()
    with torch.no_grad():
        t = teacher.encode_image(flat.cuda())
        t_k = eva_ref.encode_image_frozen_schedule(sd, cfg, flat)
        t_g = eva_ref.encode_image(sd, cfg, flat, emulate_bf16=True)
        t_f = eva_ref.encode_image(sd, cfg, flat)
        s_k = eva_ref.encode_pseudo_boxes(sd, cfg, images, rois, emulate_bf16="kernel")
        s_g = eva_ref.encode_pseudo_boxes(sd, cfg, images, rois, emulate_bf16=True)
        s_f = eva_ref.encode_pseudo_boxes(sd, cfg, images, rois)
    loss = lambda a, b: float(1.0 - (_nrm(a) * _nrm(b)).sum(-1).mean())
    l_hip, l_k, l_f = loss(s, t), loss(s_k, t_k), loss(s_f, t_f)
    _log(f"tiny end to end, normalised features rel-L2 -- teacher vs kernel-points oracle {rel(_nrm(t), _nrm(t_k)):.3e} | generic bf16 oracle "
         f"{rel(_nrm(t), _nrm(t_g)):.3e} | fp32 oracle {rel(_nrm(t), _nrm(t_f)):.3e}; student RoI vs kernel-points {rel(_nrm(s), _nrm(s_k)):.3e} | "
         f"generic bf16 {rel(_nrm(s), _nrm(s_g)):.3e} | fp32 {rel(_nrm(s), _nrm(s_f)):.3e}; loss HIP {l_hip:.6f} kernel-points {l_k:.6f} fp32 {l_f:.6f}")
    assert rel(_nrm(t), _nrm(t_k)) < 2e-3 and rel(_nrm(s), _nrm(s_k)) < 2e-3
    assert abs(l_hip - l_k) / l_k < 1e-3 and abs(l_hip - l_f) / l_f < 1e-3


def _frozen_blocks_teacher_forced(sd, cfg, teacher, flat, tag, fold_block=True, bound_s=2e-3, bound_u=3e-3, versus_fp32=False):
    """Every block 0..L-2 of the frozen schedule on the ORACLE's input stream, against the oracle that rounds where that schedule rounds
    (fold_block: norm1 / norm2 folded on the split stream, or -- the row-statistics guard's fallback -- LayerNorm kernels with only the
    sub-LayerNorms folded).  versus_fp32: also the distance of every block UPDATE to the fp32 oracle's, next to the generic bf16 oracle's own
    distance to fp32 (returned as two lists)."""
    from oracle import eva_ref
    from oracle.ops_ref import RefOps
    B = flat.shape[0]
    L, C = cfg.layers, cfg.width
    with torch.no_grad():
        _, stream = eva_ref.encode_image_frozen_schedule(sd, cfg, flat, return_stream=True, fold_block=fold_block)
    N = stream[0].shape[1]
    eng = teacher.visual.engine
    ops = eng.ops
    g = int(round((N - 1) ** 0.5))
    cos, sin = eng.rope_tables(g)
    ocos, osin = eva_ref.rope_tables(g, cfg.head_width, cfg.pt_hw_seq_len)
    worst_s = worst_u = 0.0
    d_hip, d_gen = [], []
    for i in range(L - 1):
        xin = stream[i].reshape(B * N, C)
        want = stream[i + 1].reshape(B * N, C)
        with torch.no_grad():
            if not fold_block:
                got = eng._block_fwd(i, xin.cuda().contiguous(), B, N, cos, sin, True, None, True).cpu()
            elif i == 0:
                x = xin.cuda().contiguous()
                lo = ops.empty((B * N, C), torch.int16)
                xb, st = eng._teacher_block_folded(0, x, None, None, B, N, cos, sin, emit_next=True, lo=lo)
                got = RefOps.join_planes(xb.cpu(), lo.cpu())
            else:
                hi_h, lo_h = RefOps.split_planes(xin)
                mu = xin.mean(-1)
                rstd = torch.rsqrt(((xin - mu[:, None]) ** 2).mean(-1) + cfg.ln_eps)
                xb, lo = hi_h.cuda().contiguous(), lo_h.cuda().contiguous()
                xb, st = eng._teacher_block_folded(i, None, xb, (mu.cuda(), rstd.cuda()), B, N, cos, sin, emit_next=True, lo=lo)
                got = RefOps.join_planes(xb.cpu(), lo.cpu())
        rs, ru = rel(got, want), rel(got - xin, want - xin)
        worst_s, worst_u = max(worst_s, rs), max(worst_u, ru)
        msg = f"{tag} frozen block {i:2d} teacher-forced vs its schedule's oracle: stream rel-L2 {rs:.2e}, update rel-L2 {ru:.2e}"
        if versus_fp32:
            with torch.no_grad():
                x3 = stream[i]
                f32 = eva_ref.block(sd, cfg, x3, i, ocos, osin, eva_ref._Round(False), True).reshape(B * N, C)
                gen = eva_ref.block(sd, cfg, x3, i, ocos, osin, eva_ref._Round(True), True).reshape(B * N, C)
            d_hip.append(rel(got - xin, f32 - xin))
            d_gen.append(rel(gen - xin, f32 - xin))
            msg += f" | update vs fp32 oracle: HIP {d_hip[-1]:.2e}, generic bf16 oracle {d_gen[-1]:.2e}"
        _log(msg)
        assert rs < bound_s and ru < bound_u, (tag, i, rs, ru)
    return worst_s, worst_u, d_hip, d_gen


def test_b16_blocks_teacher_forced_against_the_kernel_rounding_oracles():
    """EVA02-CLIP-B-16, BASELINE configs[0] crops: every block of both schedules on the ORACLE's input stream.
    Frozen schedule: blocks 1..L-2 through engine._teacher_block_folded on the split stream (hi / lo planes + fp32 row statistics built
    from the oracle's stream), block 0 through its fp32-in form.  Training schedule: engine._block_fwd with the activations kept (the
    student's forward), last block without attention."""
    from oracle import eva_ref
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    sd = seeded_visual_state(cfg, 0)
    student, teacher = _pair(cfg, 0)
    _, _, crops = synthetic_batch(2, 8, 224, 224, seed=1234)
    flat = crops.flatten(0, 1)[:6]
    B = flat.shape[0]
    L, C = cfg.layers, cfg.width
    worst_s, worst_u, _, _ = _frozen_blocks_teacher_forced(sd, cfg, teacher, flat, "B/16")
    N = cfg.tokens
    cos, sin = teacher.visual.engine.rope_tables(cfg.grid)
    # training schedule (student forward)
    eng = student.visual.engine
    rq = eva_ref._Round("kernel")
    _, _, crops = synthetic_batch(2, 8, 224, 224, seed=1234)
    with torch.no_grad():
        x, gg = eva_ref.stem(sd, cfg, flat, rq)
        ocos, osin = eva_ref.rope_tables(gg, cfg.head_width, cfg.pt_hw_seq_len)
        for i in range(L):
            with_attn = i < L - 1
            want = eva_ref.block(sd, cfg, x, i, ocos, osin, rq, with_attn)
            xin = x.reshape(B * N, C)
            got = eng._block_fwd(i, xin.cuda().contiguous(), B, N, cos, sin, with_attn=with_attn, save={}, inplace=False).cpu()
            rs, ru = rel(got, want.reshape(B * N, C)), rel(got - xin, want.reshape(B * N, C) - xin)
            worst_s, worst_u = max(worst_s, rs), max(worst_u, ru)
            _log(f"B/16 training block {i:2d} teacher-forced vs kernel-points oracle: stream rel-L2 {rs:.2e}, update rel-L2 {ru:.2e}")
            assert rs < 2e-3 and ru < 3e-3, (i, rs, ru)
            x = want
    _log(f"B/16 teacher-forced blocks, both schedules: worst stream rel-L2 {worst_s:.2e}, worst update rel-L2 {worst_u:.2e}")


def test_b16_cfg1_end_to_end_against_fp32_bf16_and_kernel_rounding_oracles():
    """EVA02-CLIP-B-16 at BASELINE configs[0] (2 images x 8 boxes): normalised teacher / student features and the loss against the fp32
    oracle, the generic bf16 oracle and the kernel-points oracles side by side -- and the oracle against ITSELF with fp64 accumulation, the
    yardstick for how far two correct bf16 implementations of a 12-block tower sit apart."""
    from oracle import eva_ref
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    sd = seeded_visual_state(cfg, 0)
    student, teacher = _pair(cfg, 0)
    images, boxes, crops = synthetic_batch(2, 8, 224, 224, seed=1234)
    flat = crops.flatten(0, 1)
    rois = [b[:, :4] for b in boxes]
    s = student.encode_pseudo_boxes(images.cuda(), [r.cuda() for r in rois]).detach()          # training forward (see the tiny test)
    with torch.no_grad():
        t = teacher.encode_image(flat.cuda())
        t_f = eva_ref.encode_image(sd, cfg, flat)
        t_g = eva_ref.encode_image(sd, cfg, flat, emulate_bf16=True)
        t_k = eva_ref.encode_image_frozen_schedule(sd, cfg, flat)
        s_f = eva_ref.encode_pseudo_boxes(sd, cfg, images, rois)
        s_g = eva_ref.encode_pseudo_boxes(sd, cfg, images, rois, emulate_bf16=True)
        s_k = eva_ref.encode_pseudo_boxes(sd, cfg, images, rois, emulate_bf16="kernel")
        sd64 = {k: v.double() for k, v in sd.items()}
        t_g64 = eva_ref.encode_image(sd64, cfg, flat.double(), emulate_bf16=True)
    loss = lambda a, b: float(1.0 - (_nrm(a) * _nrm(b)).sum(-1).mean())
    l_hip, l_f, l_g, l_k = loss(s, t), loss(s_f, t_f), loss(s_g, t_g), loss(s_k, t_k)
    noise = rel(_nrm(t_g), _nrm(t_g64))
    _log(f"B/16 cfg1 end to end, normalised features rel-L2 -- teacher vs fp32 oracle {rel(_nrm(t), _nrm(t_f)):.3e} | generic bf16 oracle "
         f"{rel(_nrm(t), _nrm(t_g)):.3e} | frozen-schedule oracle {rel(_nrm(t), _nrm(t_k)):.3e} (max 1-cos {one_minus_cos(t, t_k):.1e}); student RoI vs fp32 "
         f"{rel(_nrm(s), _nrm(s_f)):.3e} | generic bf16 {rel(_nrm(s), _nrm(s_g)):.3e} | kernel-points {rel(_nrm(s), _nrm(s_k)):.3e}; "
         f"bf16 oracle vs ITSELF with fp64 accumulation {noise:.3e}")
    _log(f"B/16 cfg1 loss: HIP {l_hip:.6f} | fp32 oracle {l_f:.6f} ({abs(l_hip - l_f) / l_f:.1e}) | generic bf16 {l_g:.6f} ({abs(l_hip - l_g) / l_g:.1e}) | "
         f"kernel-points {l_k:.6f} ({abs(l_hip - l_k) / l_k:.1e})")
    assert noise > 2e-3, "the chaos yardstick itself: a bf16 12-block tower is not reproducible to 2e-3 under a 1e-7 perturbation"
    # end to end every bf16 implementation sits within ~1.5x of the oracle's own fp32-vs-fp64-accumulation distance
    # (round 6: bounds cut to ~1.8x the measured 5.6e-3 / 3.6e-3 and 2.3e-5 of profiles/r06_parity.md; they were 1.5e-2 / 2e-4 since round 3)
    assert rel(_nrm(t), _nrm(t_k)) < max(1.0e-2, 1.4 * noise) and rel(_nrm(s), _nrm(s_k)) < max(1.0e-2, 1.4 * noise)
    assert one_minus_cos(t, t_k) < 6e-5 and one_minus_cos(s, s_k) < 6e-5
    for ref in (l_f, l_g, l_k):
        assert abs(l_hip - ref) / ref < 1e-3, (l_hip, ref)


def test_full_size_teacher_pass_sampled_against_the_frozen_schedule_oracle():
    """BASELINE configs[1]: all 2048 crops in one pass (M = 403 456 rows per GEMM); 63 crops sampled over the whole row range against the
    fp32 oracle and the frozen-schedule bf16 oracle side by side."""
    from oracle import eva_ref
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    sd = seeded_visual_state(cfg, 0)
    _, teacher = _pair(cfg, 0)
    _, _, crops = synthetic_batch(64, 32, 224, 224, seed=1234)
    crops = crops.flatten(0, 1)
    with torch.no_grad():
        teacher.visual.teacher_chunk = 2048
        got_all = teacher.encode_image(crops.cuda())
    idx = torch.unique(torch.cat([torch.arange(0, 2048, 37), torch.tensor([1, 255, 256, 1023, 1024, 2046, 2047])]))[:63]
    with torch.no_grad():
        want_f = eva_ref.encode_image(sd, cfg, crops[idx])
        want_k = eva_ref.encode_image_frozen_schedule(sd, cfg, crops[idx])
    got = got_all[idx.cuda()]
    _log(f"cfg1 full-size teacher pass, {len(idx)} sampled crops, normalised features rel-L2: vs fp32 oracle {rel(_nrm(got), _nrm(want_f)):.3e} | "
         f"vs frozen-schedule bf16 oracle {rel(_nrm(got), _nrm(want_k)):.3e}; max 1-cos {one_minus_cos(got, want_k):.1e}")
    assert rel(_nrm(got), _nrm(want_k)) < 1.0e-2 and one_minus_cos(got, want_k) < 6e-5      # measured 5.3e-3 / 2.2e-5 (profiles/r06_parity.md)


def test_b16_block_backward_teacher_forced_against_the_kernel_rounding_oracle():
    """The hand-written backward, one block at a time (tests/test_block_backward_cpu.py): EVA02-CLIP-B-16, 2 images x 8 boxes at 224^2.  Every
    block gets the oracle's input stream and the oracle's upstream gradient; dL/dx and all 21 parameter gradients of the block are compared
    with autograd of the kernel-points oracle.  Bounds: 5e-3 relative L2 -- 8e-3 for the q / k projection weights and q_bias, whose gradients
    pass through the attention backward's bf16 probabilities and dS (the per-kernel CPU references, which round at the same points, sit at
    5.4e-3 there).  The end-to-end gradient bounds (3e-2 in tests/test_gpu_step.py) measure these errors after 12 blocks of amplification."""
    from test_block_backward_cpu import block_backward_teacher_forced, oracle_chain
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    sd = seeded_visual_state(cfg, 0)
    student, _ = _pair(cfg, 0)
    images, boxes, _ = synthetic_batch(2, 8, 224, 224, seed=1234)
    sdg, xs = oracle_chain(sd, cfg, images, [b[:, :4] for b in boxes])
    loose = ("attn.q_proj.weight", "attn.k_proj.weight", "attn.q_bias")
    worst_dx = worst_p = 0.0
    for i in range(cfg.layers - 1, -1, -1):
        res = block_backward_teacher_forced(student, sdg, xs, cfg, i, "cuda")
        wp, wn = max((v, k) for k, v in res.items() if k != "dx")
        worst_dx, worst_p = max(worst_dx, res["dx"]), max(worst_p, wp)
        _log(f"B/16 block {i:2d} backward teacher-forced vs kernel-points oracle autograd: dL/dx rel-L2 {res['dx']:.2e}, worst parameter gradient "
             f"{wp:.2e} ({wn}), w3.weight {res['mlp.w3.weight']:.2e}, w1.weight {res['mlp.w1.weight']:.2e}, norm1.weight {res['norm1.weight']:.2e}")
        assert len(res) == (22 if i < cfg.layers - 1 else 19), sorted(res)           # last block: q / k / q_bias never differentiated
        for k, v in res.items():
            assert v < (8e-3 if k in loose else 5e-3), (i, k, v)
    _log(f"B/16 backward, all blocks teacher-forced: worst dL/dx {worst_dx:.2e}, worst parameter gradient {worst_p:.2e}")


def _pair_from(cfg, sd):
    from clipself_amd.open_clip.model import CustomCLIP
    student, teacher = CustomCLIP(cfg, trainable=True), CustomCLIP(cfg, trainable=False)
    for m in (student, teacher):
        m.visual.engine.load_state(sd)
    student.lock_image_tower(unlocked_groups=cfg.layers)
    student.train()
    teacher.eval()
    return student, teacher


@pytest.mark.parametrize("ros", [1.5, 5.0])
def test_b16_trained_statistics_stress(golden_dir, ros):
    """Weights with the activation statistics of a TRAINED ViT (oracle/stress_weights.py: four channels at x50..x200 the typical magnitude in
    every block's stream, rows with |mean| / sigma = `ros`), pinned on the real reference (tests/golden/b16_stress.npz).
      * ros = 1.5: the frozen schedule keeps all four LayerNorms folded; ros = 5: the engine's row-statistics guard measures the stream and
        keeps norm1 / norm2 as LayerNorm kernels (engine.block_folds_active);
      * every frozen block, teacher-forced, against the oracle of the schedule in use (<= 2e-3 stream / 3e-3 update) AND against the fp32
        oracle: the HIP block's distance to fp32 must stay within 2x the generic bf16 oracle's own distance to fp32 (the judge's criterion);
      * every training block (the student's forward) teacher-forced against the kernel-points oracle;
      * end to end: teacher and student features against the REFERENCE's, next to the generic bf16 oracle's distance."""
    from oracle import eva_ref
    from oracle.stress_weights import row_statistics, trained_statistics_state
    cfg = get_tower_cfg("EVA02-CLIP-B-16")
    sd = trained_statistics_state(cfg, 0, row_offset_sigmas=ros)
    gold = np.load(golden_dir / "b16_stress.npz")
    student, teacher = _pair_from(cfg, sd)
    images, boxes, crops = synthetic_batch(2, 8, 224, 224, seed=1234)
    flat = crops.flatten(0, 1)[:4]
    rois = [b[:, :4] for b in boxes]
    eng = teacher.visual.engine
    active = eng.block_folds_active(flat.cuda())
    with torch.no_grad():
        x0, _ = eva_ref.stem(sd, cfg, flat, eva_ref._Round(False))
    ratio, outlier = row_statistics(x0)
    tag = f"B/16 trained-statistics (|mean|/sigma {ratio:.2f}, largest / median deviation {outlier:.0f})"
    _log(f"{tag}: engine statistic {eng.block_fold_ratio:.2f} (limit {eng.block_fold_limit}) -> norm1 / norm2 folded: {active}")
    assert active == (ros < 2.0) and abs(ratio - ros) < 0.1 and outlier > 100
    ws, wu, d_hip, d_gen = _frozen_blocks_teacher_forced(sd, cfg, teacher, flat[:3], tag, fold_block=active, versus_fp32=True)
    worst = max(h / g for h, g in zip(d_hip, d_gen))
    _log(f"{tag}: frozen blocks vs fp32, HIP / generic-bf16-oracle distance ratio: worst {worst:.2f}, mean {sum(d_hip) / sum(d_gen):.2f}")
    assert worst < 2.0
    # training schedule (student forward), teacher-forced
    B, N, C, L = 3, cfg.tokens, cfg.width, cfg.layers
    seng = student.visual.engine
    cos, sin = seng.rope_tables(cfg.grid)
    rq = eva_ref._Round("kernel")
    wt = 0.0
    with torch.no_grad():
        x, gg = eva_ref.stem(sd, cfg, flat[:3], rq)
        ocos, osin = eva_ref.rope_tables(gg, cfg.head_width, cfg.pt_hw_seq_len)
        for i in range(L):
            want = eva_ref.block(sd, cfg, x, i, ocos, osin, rq, i < L - 1).reshape(B * N, C)
            xin = x.reshape(B * N, C)
            got = seng._block_fwd(i, xin.cuda().contiguous(), B, N, cos, sin, with_attn=i < L - 1, save={}, inplace=False).cpu()
            ru = rel(got - xin, want - xin)
            wt = max(wt, ru)
            assert ru < 3e-3, (i, ru)
            x = want.reshape(B, N, C)
    _log(f"{tag}: training blocks teacher-forced vs kernel-points oracle, worst update rel-L2 {wt:.2e}")
    # end to end against the reference's own features
    with torch.no_grad():
        t = teacher.encode_image(flat.cuda())
        t_g = eva_ref.encode_image(sd, cfg, flat, emulate_bf16=True)
        t_k = eva_ref.encode_image_frozen_schedule(sd, cfg, flat, fold_block=active)
        s_g = eva_ref.encode_pseudo_boxes(sd, cfg, images, rois, emulate_bf16=True)
    s = student.encode_pseudo_boxes(images.cuda(), [r.cuda() for r in rois]).detach()
    t_ref, s_ref = torch.from_numpy(gold[f"ros{ros:g}/teacher"]), torch.from_numpy(gold[f"ros{ros:g}/student_roi"])
    d_t, d_tg, d_tk = rel(_nrm(t), _nrm(t_ref)), rel(_nrm(t_g), _nrm(t_ref)), rel(_nrm(t), _nrm(t_k))
    d_s, d_sg = rel(_nrm(s), _nrm(s_ref)), rel(_nrm(s_g), _nrm(s_ref))
    _log(f"{tag}: end to end, normalised features rel-L2 vs the REFERENCE -- teacher HIP {d_t:.3e} | generic bf16 oracle {d_tg:.3e} | HIP vs its "
         f"schedule's oracle {d_tk:.3e} (max 1-cos {one_minus_cos(t, t_ref):.1e}); student RoI HIP {d_s:.3e} | generic bf16 oracle {d_sg:.3e} "
         f"(max 1-cos {one_minus_cos(s, s_ref):.1e})")
    assert d_t < 2.0 * d_tg + 2e-3 and d_s < 2.0 * d_sg + 2e-3
    assert one_minus_cos(t, t_ref) < 1e-3 and one_minus_cos(s, s_ref) < 1e-3
