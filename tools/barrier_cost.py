"""What the per-K-tile workgroup barrier of the streaming GEMM costs, and what waves drifting apart would buy: cs_gemm_nt flags bit 13
(dbg 2) drops the s_barrier (results are wrong: the ring slots race), bit 14 (dbg 4) the epilogue.  The barrier switch is compiled in
only with `CS_EXTRA_FLAGS=-DCS_ABLATION_SWITCHES bash clipself_amd/csrc/build.sh` (after touching gemm_stream.hip): as a run-time test it
cost 0.35 % of the step.  usage: python tools/barrier_cost.py"""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.hip import HipOps
ops = HipOps(); BF = torch.bfloat16
M = 2048 * 197
def t(fn, n=8):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, N, K in (("W1|W2 bf16 out", 4096, 768), ("W3-shaped bf16 out", 768, 2048)):
    A = torch.randn(M, K, device="cuda").to(BF); W = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    C = torch.empty(M, N, dtype=BF, device="cuda"); bias = torch.randn(N, device="cuda")
    line = name + ": "
    for label, fl in (("full", 0xB0), ("no epilogue", 0x40B0), ("no epilogue, no barrier", 0x60B0), ("no barrier", 0x20B0)):
        us = t(lambda: ops.gemm_nt(A, W, C, bias, None, epi=0, flags=fl))
        line += f"{label} {us:7.1f} us ({2.0*M*N*K/us/1e6:5.0f} TF/s) | "
    print(line, flush=True)
