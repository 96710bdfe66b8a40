#!/usr/bin/env python
"""Main loop / epilogue split of the streaming GEMM on the tower shapes: cs_gemm_nt schedule 11 with the timing ablations of a
-DCS_ABLATION_SWITCHES build (dbg 4 = no epilogue, dbg 8 = every store masked, dbg 2 = no barrier; wrong results by construction).
env ABLATE_DBG (list of dbg values), ABLATE_RESERVE (compute units left free: grid = 256 - reserve; is the epilogue's cost per CU or per chip?),
ABLATE_SHAPES (substring filter), ABLATE_DATA (randn | zeros | const | sparse operand values).   usage (GPU box): CLIPSELF_HIP_LIB=<ablation build> python tools/stream_ablate.py [crops=2048] [tag]"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.hip import HipOps  # noqa: E402

BF = torch.bfloat16


def main():
    ops = HipOps()
    crops = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    tag = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(os.environ.get("CLIPSELF_HIP_LIB", "default"))
    M = crops * 197
    shapes = [("qkv N=2304 K=768 epi0", 2304, 768, 0), ("proj N=768 K=768 epi2", 768, 768, 2),
              ("w12 N=4096 K=768 epi3", 4096, 768, 3), ("w3 N=768 K=2048 epi2", 768, 2048, 2)]
    reserve = int(os.environ.get("ABLATE_RESERVE", "0"))
    only = os.environ.get("ABLATE_SHAPES", "")
    for name, N, K, epi in shapes:
        if only and not any(o in name for o in only.split(",")):
            continue
        A = torch.randn(M, K, device="cuda").to(BF)
        B = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
        data = os.environ.get("ABLATE_DATA", "randn")           # operand values: the clock under matrix load follows the power they draw
        if data == "zeros":
            A.zero_(); B.zero_()
        elif data == "const":
            A.fill_(1.0); B.fill_(0.5)
        elif data == "sparse":                                   # 3 of 4 elements zero
            A[:, torch.arange(K, device="cuda") % 4 != 0] = 0
        bias = torch.randn(N, device="cuda")
        if epi == 0:
            C, extra, group = torch.empty(M, N, dtype=BF, device="cuda"), None, 0
        elif epi == 2:
            C = torch.randn(M, N, device="cuda")
            extra, group = C, 0
        else:
            C, extra, group = torch.empty(M, N // 2, dtype=BF, device="cuda"), None, N // 2
        line = f"[{tag} reserve={reserve} data={os.environ.get('ABLATE_DATA', 'randn')}] {name}:"
        for dbg in [int(x) for x in os.environ.get("ABLATE_DBG", "0,4,8,0,4").split(",")]:
            flags = (11 << 4) | (8 << 8) | (dbg << 12) | (reserve << 20)
            for _ in range(2):
                ops.gemm_nt(A, B, C, bias, extra, epi=epi, group=group, flags=flags)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm_nt(A, B, C, bias, extra, epi=epi, group=group, flags=flags)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 10
            line += f"  dbg{dbg} {us:7.1f} us ({2.0 * M * N * K / us / 1e6:5.0f} TF/s)"
        print(line, flush=True)


if __name__ == "__main__":
    main()
