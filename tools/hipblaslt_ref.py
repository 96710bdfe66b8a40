"""Where the vendor library lands on the teacher's GEMM shapes (plain bf16 output, no fused epilogue): torch.nn.functional.linear
(hipBLASLt / rocBLAS behind it) on the same random operands as tools/gemm_bench.py, beside cs_gemm_nt's plain bf16 epilogue.
Reference point for DESIGN.md only -- the product never calls it.   usage: python tools/hipblaslt_ref.py [crops]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.hip import HipOps  # noqa: E402

BF = torch.bfloat16


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    crops = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    M = crops * 197
    ops = HipOps()
    for name, N, K in (("W1|W2", 4096, 768), ("q|k|v", 2304, 768), ("proj", 768, 768), ("W3", 768, 2048)):
        A = torch.randn(M, K, device="cuda").to(BF)
        W = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
        bias = torch.randn(N, device="cuda")
        C = torch.empty(M, N, dtype=BF, device="cuda")
        t_lib = timeit(lambda: torch.nn.functional.linear(A, W, out=None))
        t_libb = timeit(lambda: torch.nn.functional.linear(A, W, bias.to(BF)))
        t_own = timeit(lambda: ops.gemm_nt(A, W, C, bias, None, epi=0))
        f = 2.0 * M * N * K
        print(f"{name} M={M} N={N} K={K}: F.linear {t_lib:7.1f} us ({f / t_lib / 1e6:5.0f} TF/s) | F.linear+bias {t_libb:7.1f} us ({f / t_libb / 1e6:5.0f} TF/s)"
              f" | cs_gemm_nt bf16+bias {t_own:7.1f} us ({f / t_own / 1e6:5.0f} TF/s)", flush=True)


if __name__ == "__main__":
    main()
