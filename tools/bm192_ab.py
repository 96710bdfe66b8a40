#!/usr/bin/env python
"""Student GEMM shapes (M = images * 197 rows) on the streaming kernel with 256-row tiles (CS_NO_BM192=1) and with the launcher's choice
(192-row tiles where the 256-row tiling fills the chip badly), interleaved in one process.   usage (GPU box): python tools/bm192_ab.py [images=64]"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.hip import HipOps  # noqa: E402

BF = torch.bfloat16


def timeit(fn, n=30, warm=3):
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
    M = (int(sys.argv[1]) if len(sys.argv) > 1 else 64) * 197
    shapes = [("qkv fwd   N=2304 K=768  bf16", 2304, 768, 0), ("proj fwd  N=768  K=768  resid", 768, 768, 2), ("w3 fwd    N=768  K=2048 resid", 768, 2048, 2),
              ("dgrad w3  N=2048 K=768  bf16", 2048, 768, 0), ("dgrad w12 N=768  K=4096 bf16", 768, 4096, 0), ("dgrad qkv N=768  K=2304 bf16", 768, 2304, 0),
              ("dgrad prj N=768  K=768  bf16", 768, 768, 0), ("w12 fwd   N=4096 K=768  swiglu", 4096, 768, 3)]
    for name, N, K, epi in shapes:
        A = torch.randn(M, K, device="cuda").to(BF)
        B = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
        bias = torch.randn(N, device="cuda")
        if epi == 2:
            C = torch.randn(M, N, device="cuda")
            run = lambda: ops.gemm_nt(A, B, C, bias, C, epi=2)
        elif epi == 3:
            C = torch.empty(M, N // 2, dtype=BF, device="cuda")
            run = lambda: ops.gemm_nt(A, B, C, bias, epi=3, group=N // 2)
        else:
            C = torch.empty(M, N, dtype=BF, device="cuda")
            run = lambda: ops.gemm_nt(A, B, C, bias, epi=0)
        res = {"256": [], "auto": []}
        for rep in range(3):
            os.environ["CS_NO_BM192"] = "1"
            res["256"].append(timeit(run))
            os.environ.pop("CS_NO_BM192")
            res["auto"].append(timeit(run))
        a, b = min(res["256"]), min(res["auto"])
        print(f"{name} M={M}: 256-row tiles {a:6.1f} us | launcher's choice {b:6.1f} us ({100 * (b / a - 1):+.1f} %)", flush=True)


if __name__ == "__main__":
    main()
