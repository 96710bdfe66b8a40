#!/usr/bin/env python
"""Floor table of the headline step: for every kernel class of one optimizer step (BASELINE configs[1], inline schedule) the measured
microseconds (tools/trace_sequence.py output of a `rocprofv3 --kernel-trace` run) next to the floor THIS DESIGN admits, computed from rates that were
measured in earlier rounds -- not from the measurement in the same row:

  R_LOOP   1.45 PFLOP/s   K loop of the streaming GEMM at the power-limited clock on non-trivial operands (profiles/r04_s_stream_ablate.txt: W1|W2 loop
                          only 1750 us for 2.538 TFLOP; r04_y_power_limit.md: 1.36-1.37 on randn; the same binary reaches 2.1-2.2 on zeros)
  R_TN     1.16 PFLOP/s   K loop of the token-major wgrad kernel: the NT loop x 0.8 (twice the LDS read instructions per fragment)
  BW_EPI   6.6 TB/s       epilogue bytes through the CUs' vector-memory pipes: 512 KB per tile and CU in 19.8 us x 256 CUs (r04_u_epilogue_per_cu.md)
  BW       6.3 TB/s       streaming kernels (AdamW measured at it; MI355X_MICROARCH.md: achievable HBM rate)
  VALU     4 cycles per wave64 instruction and SIMD, 16 for v_exp / v_rcp (quarter rate); 1024 SIMDs
  CLK_MEM  2.4 GHz        clock of phases that are not matrix-bound (r03/r04 PMC: AdamW 2.47, LayerNorm 2.4-2.5)
  CLK_ATT  1.75 GHz       clock observed in the attention forward inside the step (GRBM_GUI_ACTIVE / 8 / duration, profiles/r05_zz_pmc_mfma.md)
  BOUNDARY 1.4 us         dependent kernel boundary on one stream (MI355X_MICROARCH.md: 1.1-1.9)

The design is what it is: epilogues are NOT overlapped with K loops (floor = loop + epilogue), attention does NOT overlap its VALU and MFMA phases across
units (floor = the busier pipe at 100 %), every launch is its own kernel.  usage: python tools/floor_table.py profiles/rNN_sequence_inline.txt > profiles/rNN_floor.md"""
import re
import sys

R_LOOP, R_TN, BW_EPI, BW, CLK_MEM, CLK_ATT, BOUNDARY = 1.45e15, 1.16e15, 6.6e12, 6.3e12, 2.4e9, 1.75e9, 1.4
SIMDS = 1024
M_T, M_S = 2048 * 197, 64 * 197                 # teacher / student rows
C, HD3, HID = 768, 2304, 2048


def gemm(M, N, K, epi_bytes, valu_cycles_per_cu=0.0):
    """K loop at the power-limited rate + the epilogue at the vector-memory rate (or its VALU time when that is longer), in us"""
    return (2.0 * M * N * K / R_LOOP + max(epi_bytes / BW_EPI, valu_cycles_per_cu / CLK_MEM)) * 1e6


def stream(nbytes):
    return nbytes / BW * 1e6


def swiglu_epilogue_cycles(M):
    # 720 VALU instructions per wave and 128 x 64 tile slice, 128 of them v_exp / v_rcp; two waves per SIMD, tiles per CU = M/256 x 16 / 256
    return (592 * 4 + 128 * 16) * 2 * (M / 256.0) * 16 / 256.0


def attn_fwd_valu(units):
    # profiles/r06_c_attention_pipes.md: 8 818 VALU wave instructions per (crop, head) unit, 728 of them v_exp_f32
    return units * ((8818 - 728) * 4 + 728 * 16) / SIMDS / CLK_ATT * 1e6


def attn_bwd_unit_cycles(n):
    scores = n * n / 64.0
    valu = scores * (11.5 * 4 + 2 * 16)          # dQ: fma, exp, fma, mul + half a cvt; dK/dV: the same + a second cvt
    mfma = 7 * 2.0 * n * n * 64 / 32768 * 32    # seven products (S and dP twice), 32 x 32 x 16 MFMAs of 32 SIMD cycles
    return valu + mfma                          # the design runs them one after the other


# class -> (matcher on (phase, kernel name, avg us), floor in us per launch, what the floor is made of)
CLASSES = [
    ("teacher W1|W2 (+ norm2 fold, SiLU*mul, ffn_ln partials) -- DOMINANT", lambda ph, k, us: ph == "teacher" and "gemm_stream_kernel<3, true, true" in k,
     gemm(M_T, 2 * HID, C, M_T * HID * 2, swiglu_epilogue_cycles(M_T)), "loop 1751 + SwiGLU epilogue (VALU-bound: 64 exp + 64 rcp per wave-tile) 363"),
    ("teacher proj / W3 on the split stream (sub-LN folded, next block's statistics out)", lambda ph, k, us: ph == "teacher" and "gemm_stream_kernel<2, true, true, true, false, 3" in k,
     (gemm(M_T, C, C, M_T * C * 8 + M_T * 13 * 8) * 11 + gemm(M_T, C, HID, M_T * C * 8 + M_T * 13 * 8) * 10) / 21.0,
     "proj: loop 328 + 8 B/element epilogue 382; W3: loop 875 + 382 (average of 11 + 10 launches)"),
    ("teacher q|k|v (+ norm1 fold)", lambda ph, k, us: ph == "teacher" and "gemm_stream_kernel<0, true, false" in k, gemm(M_T, HD3, C, M_T * HD3 * 2), "loop 985 + bf16 store epilogue 282"),
    ("teacher attention forward (2048 crops x 12 heads x 197 tokens)", lambda ph, k, us: ph == "teacher" and "attn_fwd8" in k, attn_fwd_valu(2048 * 12),
     "VALU at 100 % (8 818 wave instructions per unit at the 1.75 GHz of this kernel); HBM floor 394"),
    ("teacher first / last residual GEMM, CLS-only block, head (fp32 stream in / out)", lambda ph, k, us: ph == "teacher" and ("gemm_stream_kernel<2, true, true, true, false, 2" in k or ("gemm_stream_kernel<0, false" in k and us > 500)),
     None, None),
    ("patch embed: im2row + GEMM (both towers)", lambda ph, k, us: "im2row" in k or "gemm_nt_kernel<5" in k, None, None),
    ("ln_stats_finalize2 (4 per teacher block)", lambda ph, k, us: "ln_stats_finalize" in k, stream((42e6 * 3 + 210e6) / 4) + BOUNDARY, "12 / 64 partial slices at 6.3 TB/s + boundary"),
    ("teacher CLS-query attention", lambda ph, k, us: "attn_cls" in k, stream(1.24e9), "1.24 GB at 6.3 TB/s"),
    ("student forward GEMMs (q|k|v, W1|W2, proj, W3 at 12 608 rows)", lambda ph, k, us: ph == "student fwd" and "gemm_stream" in k and us < 500,
     (gemm(M_S, HD3, C, M_S * HD3 * 2) + gemm(M_S, 2 * HID, C, M_S * 2 * HID * 2) + gemm(M_S, C, C, M_S * C * 8) + gemm(M_S, C, HID, M_S * C * 8)) / 4.0 + BOUNDARY,
     "FLOPs at R_LOOP + epilogue bytes (no tile quantisation: 49.25 row tiles), average of the four"),
    ("student dgrad GEMMs", lambda ph, k, us: ph == "student bwd" and "gemm_stream" in k,
     (gemm(M_S, C, HD3, M_S * C * 2) + gemm(M_S, C, 2 * HID, M_S * C * 2) + gemm(M_S, C, C, M_S * C * 2) + gemm(M_S, HID, C, M_S * HID * 2)) / 4.0 + BOUNDARY, "as above"),
    ("token-major wgrad (4 per block)", lambda ph, k, us: "gemm_tn_kernel" in k,
     (2.0 * M_S * (HD3 * C + 2 * HID * C + C * C + C * HID) / 4.0 / R_TN + 4 * (HD3 * C + 2 * HID * C + C * C + C * HID) / 4.0 * 4 / BW) * 1e6 + BOUNDARY,
     "K loop at R_TN + ~4 split-K partial slices written once"),
    ("split-K reduction", lambda ph, k, us: "splitk_reduce" in k, stream(6 * 4 * (HD3 * C + 2 * HID * C + C * C + C * HID) / 4.0) + BOUNDARY, "~5 partials read + 1 gradient written"),
    ("student attention forward", lambda ph, k, us: ph == "student fwd" and "attn_fwd8" in k, attn_fwd_valu(64 * 12) + 6.0, "VALU at 100 % + one unit's load latency (768 units on 512 slots)"),
    ("student attention backward (prep + dQ + dK/dV)", lambda ph, k, us: "attn_bwd" in k, None, None),
    ("LayerNorm forward / backward, SwiGLU forward / backward, column sums, parameter reductions", lambda ph, k, us: any(s in k for s in ("ln_fwd", "ln_bwd", "swiglu_", "colsum", "ln_param_reduce", "l2norm", "cls_row")), None, None),
    ("AdamW + W^T shadows", lambda ph, k, us: "adamw" in k or "transpose_bf16_batched" in k, None, None),
    ("RoIAlign, loss, torch index / fill glue", lambda ph, k, us: True, None, None),
]
# floors of the classes whose launches are heterogeneous: given as a total per step
TOTAL_FLOORS = {
    "teacher first / last residual GEMM, CLS-only block, head (fp32 stream in / out)": ((gemm(M_T, C, C, M_T * C * 10) + gemm(M_T, 2 * C, C, M_T * 2 * C * 2)), "first proj with the fp32 stream in (10 B / element) + the K|V GEMM of the CLS-only block"),
    "patch embed: im2row + GEMM (both towers)": (stream(1.85e9 * 33 / 32) + 2.0 * (M_T + M_S) * C * C / R_LOOP * 1e6 + (M_T + M_S) * C * 8 / BW_EPI * 1e6 + 4 * BOUNDARY, "im2row bytes at 6.3 TB/s + GEMM loop + fp32 epilogue"),
    "student attention backward (prep + dQ + dK/dV)": (11 * (768 * attn_bwd_unit_cycles(197) / SIMDS / CLK_ATT * 1e6 + 12.0 + 3 * BOUNDARY), "VALU + MFMA cycles of 768 units one after the other at 1.75 GHz + two units' load latency"),
    "LayerNorm forward / backward, SwiGLU forward / backward, column sums, parameter reductions": (None, "bytes at 6.3 TB/s + a boundary per launch"),
    "AdamW + W^T shadows": (stream(2.55e9) + stream(2 * 170e6) + 2 * BOUNDARY, "30 B / parameter + shadow transposes"),
    "RoIAlign, loss, torch index / fill glue": (None, "measured (launch-bound)"),
}
# bytes of the small streaming kernels, by name fragment: (phase, fragment) -> bytes per launch
SMALL_BYTES = [("swiglu_fwd", 155e6), ("swiglu_bwd", 258e6), ("ln_bwd_kernel<float", 155e6), ("ln_bwd_kernelIDF16bLi0ELi8", 155e6), ("ln_bwd_kernelIDF16bLi0ELi4", 77e6),
               ("ln_fwd_kernelIfLi4", 58e6), ("ln_fwd_kernelIDF16bLi8", 103e6), ("ln_fwd_kernelIDF16bLi4", 39e6), ("colsum_bf16", 58e6)]


def main():
    rows, phase = [], None
    for line in open(sys.argv[1]):
        m = re.match(r"# (student forward|teacher \+ loss|student backward \+ AdamW):", line)
        if m:
            phase = {"student forward": "student fwd", "teacher + loss": "teacher", "student backward + AdamW": "student bwd"}[m.group(1)]
            continue
        m = re.match(r"#\s+([\d.]+) us\s+(\d+) x\s+([\d.]+)\s+(.*)", line)
        if m and phase:
            rows.append((phase, m.group(4).strip(), int(m.group(2)), float(m.group(3)), float(m.group(1))))
        m = re.match(r"# last full step: (\d+) kernels, wall ([\d.]+) ms, kernel time ([\d.]+) ms", line)
        if m:
            launches, wall, busy = int(m.group(1)), float(m.group(2)), float(m.group(3))
    acc = {c[0]: [0, 0.0, 0.0] for c in CLASSES}                     # launches, measured us, floor us
    for ph, k, n, avg, tot in rows:
        for name, match, floor, _ in CLASSES:
            if match(ph, k, avg):
                a = acc[name]
                a[0] += n
                a[1] += tot
                if floor is not None:
                    a[2] += n * floor
                elif TOTAL_FLOORS[name][0] is None:
                    b = next((b for frag, b in SMALL_BYTES if frag in k), None)
                    a[2] += n * (stream(b) + BOUNDARY) if b else tot      # glue kernels: no model, measured
                break
    for name, (tf, _) in TOTAL_FLOORS.items():
        if tf is not None:
            acc[name][2] = tf
    print(f"# Floor table of the headline step ({sys.argv[1]}: {launches} launches, {busy:.2f} ms of kernel time in {wall:.2f} ms of wall; inline schedule)\n")
    print(__doc__.split("usage:")[0].split("\n", 2)[2])
    print("| kernel class | launches | measured ms | design floor ms | gap ms | measured / floor | the floor is |")
    print("|---|---:|---:|---:|---:|---:|---|")
    tm = tf = 0.0
    for name, _, floor, what in CLASSES:
        n, meas, fl = acc[name]
        if n == 0:
            continue
        what = what or TOTAL_FLOORS[name][1]
        tm += meas
        tf += fl
        print(f"| {name} | {n} | {meas / 1e3:.2f} | {fl / 1e3:.2f} | {(meas - fl) / 1e3:+.2f} | {meas / fl:.2f} | {what} |")
    print(f"| **step** | {launches} | **{tm / 1e3:.2f}** | **{tf / 1e3:.2f}** | **{(tm - tf) / 1e3:+.2f}** | **{tm / tf:.3f}** | |")
    print(f"\nfloor {tf / 1e3:.1f} ms = {64e3 / tf * 1e3:.0f} images/s; measured {tm / 1e3:.1f} ms = {64e3 / tm * 1e3:.0f} images/s in this (profiled, inline) run; "
          f"the north star's 0.40 of the MFMA peak is 73.6 ms for the step's executed FLOPs.")


if __name__ == "__main__":
    main()
