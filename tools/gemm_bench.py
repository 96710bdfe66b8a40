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
        for cfg, gm, dbg in ((7, 8, 0), (9, 8, 0), (11, 8, 0), (7, 8, 0), (9, 8, 0), (11, 8, 0), (3, 8, 0)):   # dbg needs a -DCS_ABLATION_SWITCHES build
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
        # race screen for the hand-counted waits of the streaming kernel: same accumulation order as the lockstep kernel -> bit-identical
        if epi != 2:
            ref = torch.empty_like(C)
            ops.gemm_nt(A, B, ref, bias, extra, epi=epi, group=group, flags=(3 << 4))
            bad = 0
            for _ in range(20):
                C.fill_(float("nan"))
                ops.gemm_nt(A, B, C, bias, extra, epi=epi, group=group, flags=(11 << 4))
                bad += int(not torch.equal(C, ref))
            print(f"   streaming vs lockstep bitwise mismatches in 20 runs: {bad}", flush=True)


def raster_ab():
    """Persistent kernel: grouped raster vs B-stationary raster (N parts 1/2/4) vs non-temporal operand loads, interleaved twice
    (usage: python tools/gemm_bench.py 2048 raster)."""
    ops = HipOps()
    M = int(sys.argv[1]) * 197
    for name, N, K, epi in (("w12 N=4096 K=768 epi3", 4096, 768, 3), ("qkv N=2304 K=768 epi0", 2304, 768, 0)):
        A = torch.randn(M, K, device="cuda").to(BF)
        B = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
        bias = torch.randn(N, device="cuda")
        C = torch.empty(M, N // 2 if epi == 3 else N, dtype=BF, device="cuda")
        group = N // 2 if epi == 3 else 0
        modes = [("grouped", 0x90), ("bstat/auto", 0x10090), ("bstat/1", 0x10190), ("bstat/2", 0x10290), ("bstat/4", 0x10490),
                 ("bstat+ntA/auto", 0x20090), ("bstat+ntA/1", 0x20190), ("bstat+ntA/2", 0x20290), ("grouped+ntB", 0x30090)]
        for rep in range(2):
            line = f"{name} M={M} pass {rep}: "
            for tag, flags in modes:
                for _ in range(2):
                    ops.gemm_nt(A, B, C, bias, epi=epi, group=group, flags=flags)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.gemm_nt(A, B, C, bias, epi=epi, group=group, flags=flags)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 10
                line += f"{tag} {us:7.1f} us ({2.0 * M * N * K / us / 1e6:4.0f} TF/s) | "
            print(line, flush=True)


def square():
    """Calibration against published square-GEMM numbers: M = N = K in {4096, 8192}, bf16 out, uniform random [-1, 1) operands
    (usage: python tools/gemm_bench.py 0 square)."""
    ops = HipOps()
    for n in (4096, 8192):
        A = (torch.rand(n, n, device="cuda") * 2 - 1).to(BF)
        B = (torch.rand(n, n, device="cuda") * 2 - 1).to(BF)
        C = torch.empty(n, n, dtype=BF, device="cuda")
        line = f"{n}^3: "
        for cfg in (11, 9, 7, 3):
            flags = cfg << 4
            for _ in range(3):
                ops.gemm_nt(A, B, C, epi=0, flags=flags)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_nt(A, B, C, epi=0, flags=flags)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            line += f"cfg{cfg} {us:7.1f} us ({2.0 * n ** 3 / us / 1e6:5.0f} TF/s) | "
        print(line, flush=True)


def student_shapes():
    """The student's GEMM shapes (M = images * 197 rows: forward, dgrad and the v-only last block) across tile configurations -- 50 M panels of
    256 rows leave the N = 768 products at 150 tiles (usage: python tools/gemm_bench.py 64 student)."""
    ops = HipOps()
    M = int(sys.argv[1]) * 197
    shapes = [("qkv fwd   N=2304 K=768  epi0", 2304, 768, 0), ("proj fwd  N=768  K=768  epi2", 768, 768, 2), ("w12 fwd   N=4096 K=768  epi0", 4096, 768, 0),
              ("w3 fwd    N=768  K=2048 epi2", 768, 2048, 2), ("dgrad w3  N=2048 K=768  epi0", 2048, 768, 0), ("dgrad w12 N=768  K=4096 epi0", 768, 4096, 0),
              ("dgrad qkv N=768  K=2304 epi0", 768, 2304, 0), ("head      N=512  K=768  epi1", 512, 768, 1)]
    for name, N, K, epi in shapes:
        A = torch.randn(M, K, device="cuda").to(BF)
        B = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
        bias = torch.randn(N, device="cuda")
        C = torch.randn(M, N, device="cuda") if epi in (1, 2) else torch.empty(M, N, dtype=BF, device="cuda")
        extra = C if epi == 2 else None
        line = f"{name} M={M}: "
        for cfg in (0, 1, 2, 3, 7, 9, 11):
            flags = cfg << 4
            for _ in range(3):
                ops.gemm_nt(A, B, C, bias, extra, epi=epi, flags=flags)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_nt(A, B, C, bias, extra, epi=epi, flags=flags)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            line += f"cfg{cfg} {us:6.1f} us ({2.0 * M * N * K / us / 1e6:4.0f} TF/s) | "
        print(line, flush=True)


def fold_ab():
    """A/B of the folded-LayerNorm operands: SwiGLU GEMM with / without the statistics output, residual GEMM with / without the
    folded epilogue, and the finalize kernel (usage: python tools/gemm_bench.py 512 fold)."""
    ops = HipOps()
    crops = int(sys.argv[1])
    M = crops * 197

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    A = torch.randn(M, 768, device="cuda").to(BF)
    W = (torch.randn(4096, 768, device="cuda") * 0.05).to(BF)
    bias = torch.randn(4096, device="cuda")
    hid = torch.empty(M, 2048, dtype=BF, device="cuda")
    part = torch.empty(64, M, 2, device="cuda")
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    for rep in range(2):
        t0 = timeit(lambda: ops.gemm_nt(A, W, hid, bias, epi=3, group=2048))
        t1 = timeit(lambda: ops.gemm_nt_ln(A, W, hid, bias=bias, stats_part=part, epi=3, group=2048))
        t2 = timeit(lambda: ops.ln_stats_finalize(part, 32, 2048, mean, rstd))
        print(f"w12 swiglu: plain {t0:7.1f} us | +stats {t1:7.1f} us | finalize(64 slices) {t2:6.1f} us", flush=True)
    W3 = (torch.randn(768, 2048, device="cuda") * 0.05).to(BF)
    x = torch.randn(M, 768, device="cuda")
    b3, cs = torch.randn(768, device="cuda"), torch.randn(768, device="cuda")
    ln_out = torch.empty(M, 2048, dtype=BF, device="cuda")
    g, be = torch.ones(2048, device="cuda"), torch.zeros(2048, device="cuda")
    for rep in range(2):
        t0 = timeit(lambda: ops.gemm_nt(hid, W3, x, b3, x, epi=2))
        t1 = timeit(lambda: ops.gemm_nt_ln(hid, W3, x, bias=b3, extra=x, ln_mean=mean, ln_rstd=rstd, ln_colsum=cs))
        t2 = timeit(lambda: ops.layernorm_fwd(hid, g, be, ln_out))
        print(f"w3 resid: plain {t0:7.1f} us | folded-LN epilogue {t1:7.1f} us | the LayerNorm pass it replaces {t2:6.1f} us", flush=True)




def wgrad_ab():
    """Student wgrad shapes (contraction = 64*197 tokens padded to 12672): automatic split-K vs forced splits / tile configs
    (usage: python tools/gemm_bench.py 64 wgrad)."""
    ops = HipOps()
    Mp = ((int(sys.argv[1]) * 197 + 63) // 64) * 64
    for name, N, K in (("w12 4096x768", 4096, 768), ("w3 768x2048", 768, 2048), ("qkv 2304x768", 2304, 768), ("proj 768x768", 768, 768)):
        A = torch.randn(N, Mp, device="cuda").to(BF)
        B = torch.randn(K, Mp, device="cuda").to(BF)
        C = torch.zeros(N, K, device="cuda")
        line = f"{name} Mp={Mp}: "
        for cfg, splits in ((0, 0), (3, 2), (2, 4), (1, 8)):
            def run():
                ops.gemm_nt(A, B, C, epi=4, splits=splits, flags=cfg << 4)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            line += f"c{cfg}/s{splits}: {us:6.1f} ({2.0 * N * K * Mp / us / 1e6:4.0f}) | "
        ws = torch.empty(ops.gemm_wgrad_workspace(N, K, Mp), dtype=torch.uint8, device="cuda")
        for _ in range(3):
            ops.gemm_wgrad(A, B, C, ws)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm_wgrad(A, B, C, ws)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        line += f"cs_gemm_wgrad (partials, ws {ws.numel() >> 20} MiB): {us:6.1f} ({2.0 * N * K * Mp / us / 1e6:4.0f})"
        # the same gradient from the token-major operands: no transposed copies (old path = 2 transposes + cs_gemm_wgrad)
        T = int(sys.argv[1]) * 197
        dY, X = torch.randn(T, N, device="cuda").to(BF), torch.randn(T, K, device="cuda").to(BF)
        dYt, Xt = torch.empty(N, Mp, dtype=BF, device="cuda"), torch.empty(K, Mp, dtype=BF, device="cuda")
        need = ops.gemm_wgrad_tn_workspace(N, K, T)
        if need:
            ws2 = torch.empty(need, dtype=torch.uint8, device="cuda")

            def old_path():
                ops.transpose_bf16(dY, dYt)
                ops.transpose_bf16(X, Xt)
                ops.gemm_wgrad(dYt, Xt, C, ws)
            for tag, fn in (("transposes + NT", old_path), ("TN", lambda: ops.gemm_wgrad_tn(dY, X, C, ws2))):
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 20
                line += f" | {tag}: {us:6.1f} ({2.0 * N * K * T / us / 1e6:4.0f})"
        print(line, flush=True)


def stream_ab():
    """Streaming kernel with register epilogues (cfg 11) against the persistent slab-epilogue
    kernel (cfg 9) on the teacher's four GEMMs exactly as `_teacher_block_folded` calls them (folded LayerNorms, statistics / bf16 copy
    outputs), two interleaved passes + a bitwise race screen (usage: python tools/gemm_bench.py 2048 stream)."""
    ops = HipOps()
    M = int(sys.argv[1]) * 197
    C, Hd = 768, 2048
    xb = torch.randn(M, C, device="cuda").to(BF)
    mean, rstd = torch.randn(M, device="cuda") * 0.1, torch.rand(M, device="cuda") + 0.5
    cases = []
    Wq, cq, dq = (torch.randn(3 * C, C, device="cuda") * 0.05).to(BF), torch.randn(3 * C, device="cuda"), torch.randn(3 * C, device="cuda")
    qkv = torch.empty(M, 3 * C, dtype=BF, device="cuda")
    cases.append(("qkv  N=2304 K=768  bf16+LN", 2.0 * M * 3 * C * C, qkv,
                  lambda f: ops.gemm_nt_ln(xb, Wq, qkv, bias=dq, ln_mean=mean, ln_rstd=rstd, ln_colsum=cq, epi=0, flags=f)))
    Wp, cp, dp = (torch.randn(C, C, device="cuda") * 0.05).to(BF), torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    x = torch.randn(M, C, device="cuda")
    part_x, xb2 = torch.empty(C // 64, M, 2, device="cuda"), torch.empty(M, C, dtype=BF, device="cuda")
    cases.append(("proj N=768  K=768  resid+LN+copy+stats", 2.0 * M * C * C, xb2,
                  lambda f: ops.gemm_nt_ln(xb, Wp, x, bias=dp, extra=x, ln_mean=mean, ln_rstd=rstd, ln_colsum=cp, stats_part=part_x, xb_out=xb2, epi=6, flags=f)))
    W12, c12, d12 = (torch.randn(2 * Hd, C, device="cuda") * 0.05).to(BF), torch.randn(2 * Hd, device="cuda"), torch.randn(2 * Hd, device="cuda")
    hid, part_h = torch.empty(M, Hd, dtype=BF, device="cuda"), torch.empty(4 * (Hd // 128), M, 2, device="cuda")
    cases.append(("w12  N=4096 K=768  swiglu+LN+stats", 2.0 * M * 2 * Hd * C, hid,
                  lambda f: ops.gemm_nt_ln(xb, W12, hid, bias=d12, ln_mean=mean, ln_rstd=rstd, ln_colsum=c12, stats_part=part_h, epi=3, group=Hd, flags=f)))
    W3, c3, d3 = (torch.randn(C, Hd, device="cuda") * 0.05).to(BF), torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    hin = torch.randn(M, Hd, device="cuda").to(BF)
    cases.append(("w3   N=768  K=2048 resid+LN+copy+stats", 2.0 * M * C * Hd, xb2,
                  lambda f: ops.gemm_nt_ln(hin, W3, x, bias=d3, extra=x, ln_mean=mean, ln_rstd=rstd, ln_colsum=c3, stats_part=part_x, xb_out=xb2, epi=6, flags=f)))
    for name, flops, out, run in cases:
        line = f"{name} M={M}: "
        for rep in range(2):
            # the last two columns are timing ablations (wrong results): they exist in builds with CS_EXTRA_FLAGS=-DCS_ABLATION_SWITCHES only
            for tag, f in (("cfg9", 0x90), ("cfg11", 0xB0), ("cfg11 slab", 0x10B0), ("cfg11 no epilogue", 0x40B0), ("cfg11 stores masked", 0x80B0)):
                for _ in range(2):
                    run(f)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run(f)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 10
                line += f"{tag} {us:7.1f} us ({flops / us / 1e6:4.0f} TF/s) | "
        print(line, flush=True)
        if "resid" not in name:            # race screen: every run of the streaming kernel must reproduce its own first result bit for bit
            for f in (0xB0, 0x10B0):
                run(f)
                ref = out.clone()
                bad = 0
                for _ in range(10):
                    out.fill_(float("nan"))
                    run(f)
                    bad += int(not torch.equal(out, ref))
                run(0x90)
                d = (out.float() - ref.float()).abs().max().item()
                print(f"   flags {f:#x}: {bad}/10 runs differ from the first; max |diff| to cfg 9 = {d:.3e}", flush=True)


if __name__ == "__main__":
    mode = sys.argv[2] if len(sys.argv) > 2 else ""
    {"fold": fold_ab, "wgrad": wgrad_ab, "raster": raster_ab, "square": square, "student": student_shapes, "stream": stream_ab}.get(mode, main)()
