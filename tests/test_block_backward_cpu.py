"""Per-block BACKWARD parity, teacher-forced: the forward blocks are compared one at a time on the oracle's own
stream (tests/test_gpu_parity.py); this does the same for the hand-written backward.  The kernel-points oracle (oracle/eva_ref.py,
emulate_bf16="kernel": bf16 where the training schedule stores bf16, straight-through gradients) is evaluated end to end ONCE with
autograd; for every block i the engine then gets the oracle's input stream x_i and the oracle's upstream gradient dL/dx_{i+1}, runs its own
forward of that one block (the saved activations it differentiates) and `_block_bwd`, and the results -- dL/dx_i and every parameter gradient
of the block -- are compared with the oracle's.  One block's kernel error (bf16 gradient operands, summation order) is thus measured before
the next blocks amplify it.

Here: the engine's schedule through the per-kernel CPU references (oracle/ops_ref.py) on the tiny tower; tests/test_gpu_parity.py runs the same
helper through the HIP kernels on EVA02-CLIP-B-16.  Reference call sites: eva_vit_model.py:300-332 (block), clipself.py:37-47 (loss)."""
import torch
import torch.nn.functional as F

from clipself_amd.config import tiny_cfg
from clipself_amd.init import _rng, seeded_visual_state, synthetic_batch
from oracle import eva_ref
from oracle.roi_align_ref import roi_align_1x1

BF16 = torch.bfloat16


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def oracle_chain(sd, cfg, images, rois_list, mode="kernel"):
    """The student's dense path + a cosine loss against seeded unit targets through the oracle, keeping the stream in front of every block.
    Returns (sd with .grad on every block parameter, [x_0 .. x_L] with .grad)."""
    rq = eva_ref._Round(mode)
    sdg = {k: v.clone().requires_grad_(k.startswith("visual.blocks.")) for k, v in sd.items()}
    with torch.no_grad():
        x0, g = eva_ref.stem(sdg, cfg, images, rq)
    cos, sin = eva_ref.rope_tables(g, cfg.head_width, cfg.pt_hw_seq_len)
    xs = [x0.clone().requires_grad_(True)]
    L = cfg.layers
    for i in range(L):
        y = eva_ref.block(sdg, cfg, xs[-1], i, cos, sin, rq, i < L - 1)
        y.retain_grad()
        xs.append(y)
    x = rq(eva_ref.layer_norm(xs[-1][:, 1:], sdg["visual.norm.weight"], sdg["visual.norm.bias"], cfg.ln_eps))
    dense = F.normalize(x @ rq(sdg["visual.head.weight"]).T + sdg["visual.head.bias"], dim=-1)
    B = images.shape[0]
    roi = roi_align_1x1(dense.reshape(B, g, g, -1), eva_ref.rois_from_list(rois_list, g))
    tgt = torch.from_numpy(_rng("block_backward.targets", 0).standard_normal(tuple(roi.shape)).astype("float32"))
    loss = 1.0 - (F.normalize(roi, dim=-1) * F.normalize(tgt, dim=-1)).sum(-1).mean()
    loss.backward()
    return sdg, xs


def block_backward_teacher_forced(student, sdg, xs, cfg, i, device):
    """Engine forward + backward of block i on the oracle's x_i / dL/dx_{i+1}.  Returns {name: rel-L2} for 'dx' and every parameter gradient
    of the block that the oracle differentiates."""
    eng = student.visual.engine
    ops = eng.ops
    B, N, C = xs[i].shape
    M = B * N
    g = int(round((N - 1) ** 0.5))
    cos, sin = eng.rope_tables(g)
    b = f"visual.blocks.{i}."
    with torch.no_grad():
        save = {}
        eng._block_fwd(i, xs[i].detach().reshape(M, C).to(device).contiguous(), B, N, cos, sin, with_attn=(i < cfg.layers - 1), save=save, inplace=False)
        eng.zero_grad()
        gup = xs[i + 1].grad.reshape(M, C).to(device).contiguous().clone()
        gb = gup.to(BF16)
        ws_bytes = max(ops.layernorm_bwd_workspace(M, max(C, eng.Hp)), ops.attn_bwd_workspace(B, N, cfg.heads))
        ws = (ops.empty((ws_bytes,), torch.uint8), ops.empty((max(ops.colsum_workspace(M, max(2 * eng.Hp, 3 * C)), 4),), torch.uint8))
        # in the step the LayerNorm backward that PRODUCED this block's upstream gradient has already summed its bf16 copy into the w3 bias gradient
        ops.colsum_bf16(gb, eng.g[b + "mlp.w3.bias"], ws[1])
        eng._block_bwd(i, save, gup, gb, B, N, cos, sin, ws, next_bias=None)
    out = {"dx": rel(gup.cpu(), xs[i].grad.reshape(M, C))}
    for name, p in sdg.items():
        if name.startswith(b) and p.grad is not None:
            out[name[len(b):]] = rel(eng.g[name].cpu(), p.grad)
    return out


def test_tiny_block_backward_teacher_forced_cpu():
    from clipself_amd.open_clip.model import CustomCLIP
    from oracle.ops_ref import RefOps
    cfg = tiny_cfg()
    sd = seeded_visual_state(cfg, 3)
    student = CustomCLIP(cfg, ops=RefOps(), trainable=True)
    student.visual.engine.load_state(sd)
    student.lock_image_tower(unlocked_groups=cfg.layers)
    images, boxes, _ = synthetic_batch(3, 4, cfg.image_size, cfg.image_size, seed=21)
    sdg, xs = oracle_chain(sd, cfg, images, [bx[:, :4] for bx in boxes])
    for i in range(cfg.layers):
        res = block_backward_teacher_forced(student, sdg, xs, cfg, i, "cpu")
        worst = max(res.values())
        print(f"tiny block {i}: dx {res['dx']:.2e}, worst parameter gradient {max(v for k, v in res.items() if k != 'dx'):.2e} "
              f"({max((v, k) for k, v in res.items() if k != 'dx')[1]})")
        # the last block has no attention: q / k never get a gradient in either implementation
        assert ("attn.q_proj.weight" in res) == (i < cfg.layers - 1)
        assert worst < 1e-2, res
