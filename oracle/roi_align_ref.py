"""TEST INFRASTRUCTURE -- CPU oracle, never imported by the product path.

RoIAlign restatement.  The reference calls the un-vendored third-party
``torchvision.ops.roi_align`` (unpinned: requirements.txt:2,
requirements-training.txt:2) at
  src/open_clip/eva_clip/eva_vit_model.py:628-629
     roi_align(x_NCHW, list[Tensor[k,4]], (1,1), 1.0, -1, True)[..., 0, 0]
torchvision is absent from /root/reference and from this image, so this file
restates torchvision's *published* algorithm (roi_align_kernel.cpp:
``roi_align_forward_kernel_impl`` / ``bilinear_interpolate`` /
``pre_calc_for_bilinear_interpolate``):

  * aligned=True  -> box coords shifted by -0.5 after scaling, no min-size clamp
  * sampling_ratio=-1 -> adaptive grid ceil(roi_h/ph) x ceil(roi_w/pw)
  * sample (y,x): outside [-1,H]x[-1,W] contributes 0; y<=0 -> 0;
    y_low>=H-1 -> y_low=y_high=H-1, y=H-1; same for x; bilinear weights
  * mean over max(gh*gw, 1) samples

PARITY UNPINNED at this boundary: the reference holds no test or golden vector
for roi_align.  It is pinned only by hand-derived
known-answer cases in tests/test_oracle_roialign.py.

Two implementations: ``roi_align_1x1`` (torch ops, differentiable, used as the
stand-in for torchvision when the reference is imported) and
``roi_align_1x1_loops`` (pure-python scalar loops, the line-by-line restatement,
small cases only) -- they are cross-checked in the tests.
"""
from __future__ import annotations

import math

import numpy as np
import torch

_f = np.float32   # torchvision computes box/sample coordinates in the input dtype (float32)


def _sample_points(lo, extent, grid: int):
    # y = roi_start + ph*bin + (iy + .5f) * bin_size / grid ; pooled size 1 => ph = 0, bin = extent
    return [_f(lo + _f(_f(_f(i) + _f(0.5)) * extent) / _f(grid)) for i in range(grid)]


def _box_f32(roi_row):
    """(x0,y0,x1,y1) - 0.5 and extents, all in float32 like torchvision's T=float."""
    x0, y0, x1, y1 = (_f(_f(v) - _f(0.5)) for v in roi_row[1:5])
    return x0, y0, _f(x1 - x0), _f(y1 - y0)


def _axis_weights(coords, size: int):
    """For each 1-D sample coordinate: (valid, low, high, w_low, w_high)."""
    out = []
    for c in coords:
        c = _f(c)
        if c < -1.0 or c > size:
            out.append((False, 0, 0, 0.0, 0.0))
            continue
        if c <= 0:
            c = _f(0.0)
        low = int(c)
        if low >= size - 1:
            high = low = size - 1
            c = _f(low)
        else:
            high = low + 1
        l = _f(c - _f(low))
        out.append((True, low, high, float(_f(1.0) - l), float(l)))
    return out


def roi_align_1x1_loops(feat_nhwc: torch.Tensor, rois: torch.Tensor) -> torch.Tensor:
    """feat_nhwc [B,H,W,C]; rois [K,5] = (batch, x0,y0,x1,y1) in *feature-map pixels*.
    Returns [K,C].  Scalar loops; follows roi_align_forward_kernel_impl."""
    B, H, W, C = feat_nhwc.shape
    out = torch.zeros(rois.shape[0], C, dtype=feat_nhwc.dtype)
    for k in range(rois.shape[0]):
        b = int(rois[k, 0])
        x0, y0, rw, rh = _box_f32([float(v) for v in rois[k]])
        gh, gw = math.ceil(rh), math.ceil(rw)
        count = max(gh * gw, 1)
        ys = _axis_weights(_sample_points(y0, rh, gh), H) if gh > 0 else []
        xs = _axis_weights(_sample_points(x0, rw, gw), W) if gw > 0 else []
        acc = torch.zeros(C, dtype=torch.float64)
        for (vy, yl, yh, hy, ly) in ys:
            for (vx, xl, xh, hx, lx) in xs:
                if not (vy and vx):
                    continue
                acc += (hy * hx) * feat_nhwc[b, yl, xl].double() + (hy * lx) * feat_nhwc[b, yl, xh].double() \
                    + (ly * hx) * feat_nhwc[b, yh, xl].double() + (ly * lx) * feat_nhwc[b, yh, xh].double()
        out[k] = (acc / count).to(out.dtype)
    return out


def roi_align_1x1(feat_nhwc: torch.Tensor, rois: torch.Tensor) -> torch.Tensor:
    """Differentiable (w.r.t. feat) torch version of the same algorithm."""
    B, H, W, C = feat_nhwc.shape
    flat = feat_nhwc.reshape(B * H * W, C)
    rows = []
    for k in range(rois.shape[0]):
        b = int(rois[k, 0])
        x0, y0, rw, rh = _box_f32([float(v) for v in rois[k]])
        gh, gw = math.ceil(rh), math.ceil(rw)
        count = max(gh * gw, 1)
        idx, wts = [], []
        if gh > 0 and gw > 0:
            ys = _axis_weights(_sample_points(y0, rh, gh), H)
            xs = _axis_weights(_sample_points(x0, rw, gw), W)
            for (vy, yl, yh, hy, ly) in ys:
                for (vx, xl, xh, hx, lx) in xs:
                    if not (vy and vx):
                        continue
                    base = b * H * W
                    idx += [base + yl * W + xl, base + yl * W + xh, base + yh * W + xl, base + yh * W + xh]
                    wts += [hy * hx, hy * lx, ly * hx, ly * lx]
        if idx:
            w = torch.tensor(wts, dtype=feat_nhwc.dtype) / count
            rows.append((flat[torch.tensor(idx)] * w[:, None]).sum(0))
        else:
            rows.append(flat.new_zeros(C) + 0.0 * flat[0])
    return torch.stack(rows) if rows else flat.new_zeros(0, C)


def torchvision_roi_align_standin(input_nchw, boxes, output_size, spatial_scale=1.0,
                                  sampling_ratio=-1, aligned=False):
    """Signature-compatible stand-in for ``torchvision.ops.roi_align`` restricted to
    the one configuration the reference uses (eva_vit_model.py:628-629)."""
    assert tuple(output_size) == (1, 1) and spatial_scale == 1.0 and sampling_ratio == -1 and aligned is True
    if isinstance(boxes, (list, tuple)):
        parts = [torch.cat([torch.full((len(b), 1), float(i), dtype=b.dtype), b.detach().cpu()], dim=1)
                 for i, b in enumerate(boxes)]
        rois = torch.cat(parts) if parts else input_nchw.new_zeros(0, 5)
    else:
        rois = boxes
    feat = input_nchw.permute(0, 2, 3, 1)
    return roi_align_1x1(feat, rois.to(torch.float64))[..., None, None]
