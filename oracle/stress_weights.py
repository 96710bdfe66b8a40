"""TEST INFRASTRUCTURE -- seeded weights with the activation statistics of a TRAINED vision transformer, for parity tests only.

The seeded N(0, 0.02) recipe (clipself_amd/init.py) gives a residual stream with row means ~ 0 and no outlier channels.  Trained ViTs have
both: a handful of "massive activation" channels 50-200x the typical magnitude, and rows whose mean is several row-sigmas away from zero.
The frozen tower's schedule depends on that being harmless: every LayerNorm is folded as rstd * (bf16(x) . W gamma - mean * colsum) + b
(csrc/gemm_stream.hip), i.e. the GEMM contracts the UN-centred bf16 row and the mean is removed afterwards in fp32 -- a cancellation whose
error grows with |mean| / sigma -- and the sub-LayerNorm statistics are one-pass sums of x and x^2 in fp32.

`trained_statistics_state` = seeded_visual_state with the stream that ENTERS block 0 reshaped (through pos_embed, cls_token) and kept that
way through the tower (through every block's mlp.w3.bias, which each block's second residual GEMM adds to the stream):
  * `channels`: pos_embed / cls_token entries set to +-`scales` x the typical |stream value| (sign fixed per channel, 10 % jitter per token:
    massive activations sit at the same channels with the same sign at every token), every block's w3.bias entry at those channels
    scaled by the same factors;
  * a constant added to every entry of pos_embed such that mean over rows of |row mean| / row sigma = `row_offset_sigmas`, the row sigma
    taken WITH the outlier channels.
Both are measured on the stem output of a seeded batch (`calibration_images`), so the numbers hold for the stream, not for pos_embed alone.
Pinned like every other fixture: oracle/gen_golden.py::gen_stress runs the REAL reference on these weights (tests/golden/b16_stress.npz)."""
from __future__ import annotations

import numpy as np
import torch

from clipself_amd.init import _rng, seeded_visual_state, synthetic_batch

CHANNELS = (7, 130, 401, 700)
SCALES = (200.0, -100.0, 50.0, -100.0)


def calibration_images(cfg, n: int = 2):
    return synthetic_batch(n, 1, cfg.image_size, cfg.image_size, seed=4242)[0]


def trained_statistics_state(cfg, seed: int = 0, prefix: str = "visual.", channels=CHANNELS, scales=SCALES, row_offset_sigmas: float = 5.0):
    from . import eva_ref
    sd = seeded_visual_state(cfg, seed, prefix)
    C = cfg.width
    chans = [c % C for c in channels]
    with torch.no_grad():
        x0, _ = eva_ref.stem(sd, cfg, calibration_images(cfg), eva_ref._Round(False), prefix)
    typical = float(x0.abs().median())
    pos = sd[prefix + "pos_embed"]
    for ch, s in zip(chans, scales):
        jitter = 1.0 + 0.1 * torch.from_numpy(_rng(f"stress.{ch}", seed).standard_normal(pos.shape[1]).astype(np.float32))
        pos[0, :, ch] = s * typical * jitter
        sd[prefix + "cls_token"][..., ch] = 0.0                          # CLS row = cls_token + pos_embed[0]: the outlier is in pos_embed
        for i in range(cfg.layers):
            sd[f"{prefix}blocks.{i}.mlp.w3.bias"][ch] *= abs(s)
    if row_offset_sigmas:
        with torch.no_grad():
            x0, _ = eva_ref.stem(sd, cfg, calibration_images(cfg), eva_ref._Round(False), prefix)
        rows = x0.reshape(-1, C).double()
        mu, sg = rows.mean(-1), rows.std(-1)
        lo, hi = 0.0, 1e4 * typical                                      # bisection on c: mean(|mu + c| / sigma) = row_offset_sigmas (sigma does not move)
        for _ in range(60):
            c = 0.5 * (lo + hi)
            if float(((mu + c).abs() / sg).mean()) < row_offset_sigmas:
                lo = c
            else:
                hi = c
        pos += float(0.5 * (lo + hi))
    return sd


def row_statistics(x: torch.Tensor):
    """(mean over rows of |row mean| / row sigma, largest |value| / median |value - row mean|) of a [..., C] stream: what the fixture stresses."""
    x = x.detach().double().reshape(-1, x.shape[-1])
    mu, sd = x.mean(-1), x.std(-1)
    return float((mu.abs() / sd).mean()), float(x.abs().max() / (x - mu[:, None]).abs().median())
