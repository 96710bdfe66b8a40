#!/usr/bin/env python
"""The frozen tower's four per-block GEMMs exactly as `engine._teacher_block_folded` launches them on the split stream (q|k|v with norm1 folded,
proj on the (hi, lo) planes, W1|W2 with norm2 + SiLU*mul + ffn_ln partials, W3 on the planes), timed with HIP events, plus a bitwise
repeatability screen of every launch (hand-counted vmcnt waits: a race shows up as run-to-run differences).

usage (GPU box): python tools/gemm_ab.py [crops=2048] [rounds=2] [tag]
A/B of two builds: run it once per library with CLIPSELF_HIP_LIB=<path to libclipself_hip.so> (tools/ab_libs.sh interleaves them)."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.hip import HipOps  # noqa: E402

BF, F32 = torch.bfloat16, torch.float32


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    ops = HipOps()
    ops.gemm_flags = int(os.environ.get("GEMM_AB_FLAGS", "0"), 0)      # e.g. 0x1000: slab form of the bf16 / SwiGLU epilogues
    crops = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    tag = sys.argv[3] if len(sys.argv) > 3 else os.environ.get("CLIPSELF_HIP_LIB", "default")
    M, C, Hd = crops * 197, 768, 2048
    g = torch.Generator(device="cuda").manual_seed(7)
    rn = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc
    xb = rn(M, C).to(BF)
    mean, rstd = rn(M, sc=0.1), torch.rand(M, device="cuda", generator=g) + 0.5
    Wq, cq, dq = rn(3 * C, C, sc=0.05).to(BF), rn(3 * C), rn(3 * C)
    qkv = torch.empty(M, 3 * C, dtype=BF, device="cuda")
    Wp, cp, dp = rn(C, C, sc=0.05).to(BF), rn(C), rn(C)
    att = rn(M, C).to(BF)
    hi0, lo0 = rn(M, C).to(BF), torch.randint(-32768, 32767, (M, C), device="cuda", dtype=torch.int16, generator=g)
    hi, lo = hi0.clone(), lo0.clone()
    part_x = torch.empty(C // 64, M, 2, device="cuda")
    W12, c12, d12 = rn(2 * Hd, C, sc=0.05).to(BF), rn(2 * Hd), rn(2 * Hd)
    hid, part_h = torch.empty(M, Hd, dtype=BF, device="cuda"), torch.empty(4 * (Hd // 128), M, 2, device="cuda")
    W3, c3, d3 = rn(C, Hd, sc=0.05).to(BF), rn(C), rn(C)
    hin = rn(M, Hd).to(BF)
    x32 = rn(M, C)
    cases = [
        ("qkv  N=2304 K=768  bf16 + norm1", 2.0 * M * 3 * C * C, (qkv,), None,
         lambda: ops.gemm_nt_ln(xb, Wq, qkv, bias=dq, ln_mean=mean, ln_rstd=rstd, ln_colsum=cq, epi=0)),
        ("proj N=768  K=768  split stream + stats", 2.0 * M * C * C, (hi, lo, part_x), (hi, lo),
         lambda: ops.gemm_nt_ln_split(att, Wp, hi, lo, dp, mean, rstd, cp, stats_part=part_x)),
        ("w12  N=4096 K=768  norm2 + swiglu + stats", 2.0 * M * 2 * Hd * C, (hid, part_h), None,
         lambda: ops.gemm_nt_ln(xb, W12, hid, bias=d12, ln_mean=mean, ln_rstd=rstd, ln_colsum=c12, stats_part=part_h, epi=3, group=Hd)),
        ("w3   N=768  K=2048 split stream + stats", 2.0 * M * C * Hd, (hi, lo, part_x), (hi, lo),
         lambda: ops.gemm_nt_ln_split(hin, W3, hi, lo, d3, mean, rstd, c3, stats_part=part_x)),
        ("proj N=768  K=768  fp32 in -> split", 2.0 * M * C * C, (hi, lo, part_x), None,
         lambda: ops.gemm_nt_ln_split(att, Wp, hi, lo, dp, mean, rstd, cp, x_in=x32, stats_part=part_x)),
        ("w3   N=768  K=2048 split -> fp32 out", 2.0 * M * C * Hd, (x32,), None,
         lambda: ops.gemm_nt_ln_split(hin, W3, hi0, lo0, d3, mean, rstd, c3, x_out=x32)),
    ]
    for r in range(rounds):
        for name, flops, outs, inplace, run in cases[:4]:
            us = timeit(run)
            print(f"[{tag}] round {r} {name} M={M}: {us:8.1f} us  {flops / us / 1e6:6.0f} TF/s", flush=True)
    if os.environ.get("GEMM_AB_NOREP"):
        return
    for name, flops, outs, inplace, run in cases:
        def once():
            if inplace is not None:
                hi.copy_(hi0)
                lo.copy_(lo0)
            for o in outs:
                if inplace is None or all(o is not t for t in inplace):
                    o.view(torch.int16 if o.element_size() == 2 else torch.int32).fill_(-1)
            run()
            torch.cuda.synchronize()
            return [o.clone() for o in outs]
        ref = once()
        bad = 0
        for _ in range(5):
            got = once()
            bad += int(not all(torch.equal(a.view(torch.int16 if a.element_size() == 2 else torch.int32),
                                           b.view(torch.int16 if b.element_size() == 2 else torch.int32)) for a, b in zip(ref, got)))
        csum = sum(float(o.float().nan_to_num(0.0, 0.0, 0.0).abs().sum()) for o in ref)
        print(f"[{tag}] repeatability {name}: {bad}/5 runs differ from the first; checksum {csum:.6e}", flush=True)


if __name__ == "__main__":
    main()
