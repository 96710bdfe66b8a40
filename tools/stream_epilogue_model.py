#!/usr/bin/env python
"""CPU model of the lane/register data movement in csrc/gemm_stream.hip's register epilogues (index arithmetic only: MFMA 32x32
accumulator layout with swapped operands, ds_bpermute broadcast of per-column constants, v_permlane32_swap widening, store offsets).
Run on CPU: python tools/stream_epilogue_model.py -- asserts that every output element lands where the GEMM contract says."""
import numpy as np

LANES = np.arange(64)
L31, HF = LANES & 31, LANES >> 5


def acc_of(tile):            # tile [128 rows, 64 cols] of one wave -> acc[i][j][e][lane] (swapped operands: lane = row)
    acc = np.zeros((4, 2, 16, 64), tile.dtype)
    for i in range(4):
        for j in range(2):
            for e in range(16):
                acc[i, j, e] = tile[i * 32 + L31, j * 32 + 8 * (e >> 2) + 4 * HF + (e & 3)]
    return acc


def bperm(addr, v):          # ds_bpermute_b32: lane l receives v[addr[l] / 4]
    return v[(addr >> 2) & 63]


def swap32(x, y):            # v_permlane32_swap: lanes 32-63 of x <-> lanes 0-31 of y
    nx, ny = x.copy(), y.copy()
    nx[32:], ny[:32] = y[:32], x[32:]
    return nx, ny


def check_bf16_like(ncols_per_j):
    """bf16 epilogue (ncols_per_j = 32, two column tiles) and SwiGLU (one 32-hidden tile): packed 4-column groups -> 16-byte stores."""
    rng = np.random.default_rng(0)
    ldc = 200
    tile = rng.integers(1, 1 << 20, size=(128, 64)).astype(np.int64)
    bias = rng.integers(1, 1 << 10, size=64).astype(np.int64)
    acc = acc_of(tile)
    out = np.zeros((128, ldc), np.int64)                      # units: one bf16 element per slot
    sl0 = HF << 4
    for j in range(2):
        pk = np.zeros((4, 4, 4, 64), np.int64)                # [i][q][r] values (two per dword in the kernel)
        for q in range(4):
            for r in range(4):
                sl = sl0 + ((j * 32 + 8 * q + r) << 2)
                b = bperm(sl, bias)                            # lane = column constant vector: bias[lane]
                for i in range(4):
                    pk[i, q, r] = acc[i, j, q * 4 + r] + b
        for i in range(4):
            rowoff = (i * 32 + L31) * ldc + j * 32 + 8 * HF   # in elements
            for pr in range(2):
                x0, y0 = swap32(np.stack([pk[i, 2 * pr, 0], pk[i, 2 * pr, 1]], -1), np.stack([pk[i, 2 * pr + 1, 0], pk[i, 2 * pr + 1, 1]], -1))
                x1, y1 = swap32(np.stack([pk[i, 2 * pr, 2], pk[i, 2 * pr, 3]], -1), np.stack([pk[i, 2 * pr + 1, 2], pk[i, 2 * pr + 1, 3]], -1))
                vec = np.concatenate([x0, x1, y0, y1], -1)    # 8 elements = 16 bytes per lane
                off = rowoff + 16 * pr
                for l in range(64):
                    out[off[l] // ldc, off[l] % ldc:off[l] % ldc + 8] = vec[l]
    want = tile + bias[None, :]
    assert np.array_equal(out[:, :64], want), "bf16 epilogue mapping"


def check_swiglu():
    rng = np.random.default_rng(1)
    ldc = 100
    tile = rng.integers(1, 1 << 20, size=(128, 64)).astype(np.int64)     # cols 0-31 = x1 of hidden 0..31, cols 32-63 = x2
    cb = rng.integers(1, 1 << 10, size=64).astype(np.int64)              # lanes 0-31: b1[hidden], lanes 32-63: b2[hidden]
    acc = acc_of(tile)
    out = np.zeros((128, ldc), np.int64)
    sl0 = HF << 4
    pk = np.zeros((4, 4, 4, 64), np.int64)
    for q in range(4):
        for r in range(4):
            sl = sl0 + ((8 * q + r) << 2)
            b1, b2 = bperm(sl, cb), bperm(sl + 128, cb)
            for i in range(4):
                pk[i, q, r] = (acc[i, 0, q * 4 + r] + b1) * 1000003 + (acc[i, 1, q * 4 + r] + b2)       # any injective f(x1, x2)
    for i in range(4):
        rowoff = (i * 32 + L31) * ldc + 8 * HF
        for pr in range(2):
            x0, y0 = swap32(np.stack([pk[i, 2 * pr, 0], pk[i, 2 * pr, 1]], -1), np.stack([pk[i, 2 * pr + 1, 0], pk[i, 2 * pr + 1, 1]], -1))
            x1, y1 = swap32(np.stack([pk[i, 2 * pr, 2], pk[i, 2 * pr, 3]], -1), np.stack([pk[i, 2 * pr + 1, 2], pk[i, 2 * pr + 1, 3]], -1))
            vec = np.concatenate([x0, x1, y0, y1], -1)
            off = rowoff + 16 * pr
            for l in range(64):
                out[off[l] // ldc, off[l] % ldc:off[l] % ldc + 8] = vec[l]
    want = (tile[:, :32] + cb[None, :32]) * 1000003 + (tile[:, 32:] + cb[None, 32:])
    assert np.array_equal(out[:, :32], want), "SwiGLU epilogue mapping"


def check_resid():
    rng = np.random.default_rng(2)
    ldc = 96
    tile = rng.integers(1, 1 << 20, size=(128, 64)).astype(np.int64)
    bias = rng.integers(1, 1 << 10, size=64).astype(np.int64)
    acc = acc_of(tile)
    out = np.zeros((128, ldc), np.int64)
    sl0 = HF << 4
    for j in range(2):
        bq = [bperm(sl0 + ((j * 32 + 8 * (e >> 2) + (e & 3)) << 2), bias) for e in range(16)]
        for i in range(4):
            rowoff = (i * 32 + L31) * ldc + j * 32 + 4 * HF                # fp32 elements
            for q in range(4):
                o = np.stack([acc[i, j, q * 4 + r] + bq[q * 4 + r] for r in range(4)], -1)
                off = rowoff + 8 * q
                for l in range(64):
                    out[off[l] // ldc, off[l] % ldc:off[l] % ldc + 4] = o[l]
    assert np.array_equal(out[:, :64], tile + bias[None, :]), "residual epilogue mapping"


if __name__ == "__main__":
    check_bf16_like(32)
    check_swiglu()
    check_resid()
    print("stream epilogue index model: ok")
