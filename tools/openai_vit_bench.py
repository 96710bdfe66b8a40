#!/usr/bin/env python
"""One-GPU timing of the CLIPSelf step on the OpenAI-CLIP ViT family, same batch shape as BASELINE configs[1]:
usage (GPU box): python tools/openai_vit_bench.py [ViT-B-16 [images [crops [size [plain]]]]]     ("plain" = teacher without the folded
LayerNorms / CLS-only last block, the A/B switch of engine_openai.py)."""
import sys
import time
from pathlib import Path
from types import SimpleNamespace

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.init import synthetic_batch  # noqa: E402
from clipself_amd.open_clip import create_model  # noqa: E402
from clipself_amd.training.clipself import CLIPSelf  # noqa: E402
from clipself_amd.training.optim import FlatAdamW  # noqa: E402
from clipself_amd.training.train import train_step  # noqa: E402

argv = sys.argv[1:]
MODEL = argv[0] if argv else "ViT-B-16"
B, K = (int(argv[1]) if len(argv) > 1 else 64), (int(argv[2]) if len(argv) > 2 else 32)
dev = "cuda:0"
student = create_model(MODEL, "", precision="amp_bf16", device=dev)
teacher = create_model(MODEL, "", precision="amp_bf16", device=dev, trainable=False)
cfg = student.visual.cfg
S = int(argv[3]) if len(argv) > 3 else cfg.image_size
plain = len(argv) > 4 and argv[4] == "plain"
if plain:
    teacher.visual.engine.fold_block_ln = teacher.visual.engine.cls_only_last_block = False
if len(argv) > 4 and argv[4] == "nocls":                     # the CLS-only last block with its own LayerNorm pass over the whole stream (round 3 form)
    teacher.visual.engine.fold_cls_block = False
if len(argv) > 4 and argv[4] == "nosplit":                  # the folded schedule with the fp32 stream + bf16 copy instead of the two 16-bit planes
    teacher.visual.engine.split_stream = False
student.lock_image_tower(unlocked_groups=cfg.layers)
student.train(); teacher.eval()
opt = FlatAdamW(student, lr=1e-5, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
args = SimpleNamespace(device=dev, precision="amp_bf16", distributed=False, skip_scheduler=True, grad_clip_norm=None, multiscale=False,
                       extract_type="v2", cosine_weight=1.0)
batches = [tuple(t.to(dev) for t in synthetic_batch(B, K, S, S, seed=5 + j)) for j in range(2)]
method = CLIPSelf()
for i in range(3):
    out, _, _ = train_step(student, method, batches[i % 2], opt, None, i, teacher, args, next_batch=batches[(i + 1) % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 6
for i in range(n):
    out, _, _ = train_step(student, method, batches[(i + 1) % 2], opt, None, 3 + i, teacher, args, next_batch=batches[i % 2])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
g = S // cfg.patch_size
N, C, Hd, E, L, p = g * g + 1, cfg.width, cfg.hidden, cfg.embed_dim, cfg.layers, cfg.patch_size
pe = 2 * (N - 1) * 3 * p * p * C
blk = 8 * N * C * C + 4 * N * N * C + 4 * N * C * Hd                 # in_proj + out_proj, attention, c_fc + c_proj
blk_na = 4 * N * C * C + 4 * N * C * Hd                                # last dense block: value third of in_proj + out_proj, MLP
blk_cls = 4 * N * C * C + 4 * C * C + 4 * N * C + 4 * C * Hd          # teacher's last block for the CLS query only
T = pe + (L - 1) * blk + (blk if plain else blk_cls) + 2 * C * E
F = K * T + (pe + (L - 1) * blk + blk_na + 2 * (N - 1) * C * E) + 2 * ((L - 1) * blk + blk_na) + 2 * (N - 1) * C * E
print(f"{MODEL}{' (plain teacher schedule)' if plain else ''}: {B} images x {K} crops at {S}^2: {1e3 * dt:.1f} ms/step, {B / dt:.1f} images/s, "
      f"{F * B / dt / 1e12:.0f} TFLOP/s of executed matmul FLOPs ({F * B / dt / 2.5e15:.1%} of the 2.5 PFLOP/s MFMA peak), "
      f"loss {float(out['loss'].detach()):.4f}", flush=True)
