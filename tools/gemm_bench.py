#!/usr/bin/env python
"""GPU micro-benchmark of cs_gemm_nt tile configurations on the teacher's GEMM shapes (A/B evidence for DESIGN.md).
usage (GPU box): python tools/gemm_bench.py [chunk_crops]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.hip import HipOps  # noqa: E402

BF, F32 = torch.bfloat16, torch.float32


def main():
    ops = HipOps()
    crops = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    M = crops * 197
    shapes = [("qkv  N=2304 K=768  epi0", 2304, 768, 0), ("proj N=768  K=768  epi2", 768, 768, 2),
              ("w12  N=4096 K=768  epi3", 4096, 768, 3), ("w3   N=768  K=2048 epi2", 768, 2048, 2)]
    for name, N, K, epi in shapes:
        A = torch.randn(M, K, device="cuda").to(BF)
        B = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
        bias = torch.randn(N, device="cuda")
        if epi == 0:
            C, extra, group = torch.empty(M, N, dtype=BF, device="cuda"), None, 0
        elif epi == 2:
            C = torch.randn(M, N, device="cuda")
            extra, group = C, 0
        else:
            C, extra, group = torch.empty(M, N // 2, dtype=BF, device="cuda"), None, N // 2
        line = f"{name} M={M}: "
        for cfg, gm, dbg in ((3, 8, 0), (3, 8, 0), (7, 8, 0), (8, 8, 0), (3, 8, 4), (3, 8, 5)):   # dbg: 1 no DMA, 2 no MFMA, 3 neither
            flags = (cfg << 4) | (gm << 8) | (dbg << 12)
            for _ in range(3):
                ops.gemm_nt(A, B, C, bias, extra, epi=epi, group=group, flags=flags)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_nt(A, B, C, bias, extra, epi=epi, group=group, flags=flags)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            line += f" cfg{cfg}/gm{gm}/dbg{dbg}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.0f} TF/s |"
        print(line, flush=True)
        # race screen for the ping-pong schedule: it accumulates in the same order as the lockstep kernel -> bit-identical
        if epi != 2:
            ref = torch.empty_like(C)
            ops.gemm_nt(A, B, ref, bias, extra, epi=epi, group=group, flags=(3 << 4))
            bad = 0
            for _ in range(20):
                C.fill_(float("nan"))
                ops.gemm_nt(A, B, C, bias, extra, epi=epi, group=group, flags=(5 << 4))
                bad += int(not torch.equal(C, ref))
            print(f"   ping-pong vs lockstep bitwise mismatches in 20 runs: {bad}", flush=True)


if __name__ == "__main__":
    main()
