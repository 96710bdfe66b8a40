// Streaming persistent bf16 MFMA GEMM with register-level epilogues (gfx950) -- the default schedule of the frozen tower's GEMMs.
//
//   C[M,N] (+epilogue) = A[M,K] . B[N,K]^T      same operands, LDS image and 256x256x64 tile / 8 waves of 128x64 as gemm_persist_kernel
//                                               (gemm.hip); what changes is everything around the MFMA loop:
//
//   * The MFMA operands are swapped: acc = mfma(B fragment, A fragment), i.e. every wave accumulates its tile TRANSPOSED.  A lane then
//     owns one output ROW (row = lane & 31 of the 32x32 block) and its 16 accumulator registers are four groups of four CONSECUTIVE
//     columns (col = 8*(e>>2) + 4*(lane>>5) + (e&3)).  Per-row LayerNorm statistics become per-lane scalars, four results pack into
//     8 bytes of bf16 in registers, one v_permlane32_swap per dword pairs the two half-waves' groups into 16 contiguous bytes, and
//     the output leaves as 16-byte row-segment stores straight from registers: no LDS slab, no ds_write/ds_read transposition, no
//     lgkmcnt round trips in the epilogue.  Per-column constants (bias, folded-LN column sums) are fetched with ONE coalesced dword
//     per lane and broadcast with ds_bpermute (the LDS crossbar, no LDS memory).
//   * The operand ring runs CONTINUOUSLY across output tiles: K tile g of the workgroup's tile sequence lives in A slot g % 3 /
//     B slot g & 1, and iteration g issues B(g+1) and A(g+2) whatever tile they belong to, so the first operands of the next output
//     tile are in flight two K tiles before the current tile's epilogue and there is no per-tile prologue, no extra barrier.
//   * Round 4: asymmetric DMA roles -- waves 0-3 issue the B tile (k-steps 0-1), waves 4-7 the A tile (k-steps 2-3); see ROLES below.
//   * The K step is ONE basic block with a fixed issue order (round 3): the eight DMA pieces are issued unconditionally (past its last tile
//     a cursor keeps cycling over that tile's K tiles: valid memory, slots nobody reads), the cursor advance sits behind the MFMAs, and the
//     six ds_read_b128 of k-step ks+1 go one behind each of the first six MFMAs of k-step ks (two fragment register sets,
//     sched_group_barrier); left to itself hipcc clusters reads and MFMAs into runs of 2-4 with lgkmcnt(0) between them.
//   * The epilogue's stores are never waited for: vmcnt retires in order on gfx9, so the first K tile after an epilogue waits
//     vmcnt(4 + S) with S = the fixed number of store instructions an epilogue issues (buffer stores with out-of-range offsets for
//     masked rows / columns keep that count exact).
//   * The epilogue's own operand loads (row statistics, column constants, residual rows) are ordinary loads: hipcc waits vmcnt(0) at
//     their first use, which also retires the next tile's operand DMA -- issued one to two K tiles earlier, it has landed by then.
//     (Measured: variants that fetched these operands during the tile's last K iteration -- by hand-counted inline-asm loads ahead of
//     that iteration's DMA, or by ordinary loads behind a burst issue of its DMA -- were not faster.)
//
// Epilogues: bf16 (+bias, optional GELU / QuickGELU, optional folded LayerNorm), fused SwiGLU (+folded LayerNorm, +row statistics of the
// hidden matrix), fp32 residual (+folded LayerNorm, + bf16 copy and row statistics of the new stream).  Reference call sites:
// eva_vit_model.py:99-103 (SwiGLU), :177-179 (q|k|v), :218-219 (proj), :306-307 (residual adds); open_clip/transformer.py:195-211.
#include <vector>
#include <cstdio>
#include "gemm_common.h"

namespace {

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr unsigned OOB = 0xfffffff0u;          // voffset of a masked lane: beyond num_records of every descriptor below -> the store is dropped

__device__ __forceinline__ float lane_bcast(int src_lane_x4, float v) {      // v of lane (src_lane_x4 / 4)
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane_x4, __float_as_int(v)));
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    union { bf16x2 v; unsigned u; } pk;
    pk.v[0] = f2bf(a);
    pk.v[1] = f2bf(b);
    return pk.u;
}
// half exchange: lanes 32-63 of x <-> lanes 0-31 of y (v_permlane32_swap)
__device__ __forceinline__ void swap32(unsigned& x, unsigned& y) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}
__device__ __forceinline__ float both_halves(float v) {                     // v(lane) + v(lane ^ 32)
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
// CP = cache policy of the epilogue's streaming accesses (gfx950 aux bits: 0 default | 2 = nt | 16 = sc1 | 17 = sc0 sc1): outputs are
// written once and read by a later kernel, the residual rows are read once -- neither should displace the operand panels in L2
template <int CP>
__device__ __forceinline__ void store16(const u32x4& v, __amdgpu_buffer_rsrc_t rs, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, CP);
}
template <int CP>
__device__ __forceinline__ void store8(const u32x2& v, __amdgpu_buffer_rsrc_t rs, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, CP);
}
// c + a.lo * b.lo + a.hi * b.hi on packed bf16 pairs (v_dot2c_f32_bf16)
__device__ __forceinline__ float dot2_bf16(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}
__device__ __forceinline__ float silu_mul(float u, float v) {               // hardware exp2 / rcp (1 ulp each; the result is rounded to bf16)
    return u * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * u)) * v;
}
template <int ACT>
__device__ __forceinline__ float activate_s(float v) {
    if (ACT == 1) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));                                            // nn.GELU
    if (ACT == 2) return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v));   // QuickGELU
    return v;
}

// Epilogue operands that do not depend on the accumulators: per-row LayerNorm statistics of the wave's four 32-row blocks (lane = row)
// and the per-column constants of its 64 columns (lane = column).
struct EpiOps {
    float mean[4], rstd[4];
    float cb, cc;            // bias / folded-LN column sum of column (lane) -- SwiGLU: lanes 0-31 = x1 columns, 32-63 = x2 columns
};

// ---------------------------------------------------------------------------------------------------------------- SwiGLU
// acc[i][0] = x1, acc[i][1] = x2 of hidden units hbase + col(e, lane>>5); out[row][hidden] = silu(x1) * x2 in bf16.
template <bool LN, bool AUX, int CP, int FM = 4>
__device__ __forceinline__ void epi_swiglu(const GemmArgs& p, const f32x16 (&acc)[FM][2], int lane, int row0, int tn, int wn, const EpiOps& eo) {
    const int l31 = lane & 31, hf = lane >> 5;
    const int hbase = tn * 128 + wn * 32;
    const bool colok = hbase < p.group;                                 // wave-uniform: group % 32 == 0 (checked by the launcher)
    float nm[FM], rs[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        rs[i] = LN ? eo.rstd[i] : 1.f;
        nm[i] = LN ? -eo.rstd[i] * eo.mean[i] : 0.f;
    }
    unsigned pk[FM][4][2];
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    const int sl0 = hf << 4;                                            // byte address of source lane 4*hf
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float hv[FM][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = q * 4 + r;
            const int sl = sl0 + ((8 * q + r) << 2);
            const float b1 = lane_bcast(sl, eo.cb), b2 = lane_bcast(sl + 128, eo.cb);
            float c1 = 0.f, c2 = 0.f;
            if (LN) { c1 = lane_bcast(sl, eo.cc); c2 = lane_bcast(sl + 128, eo.cc); }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                float u = acc[i][0][e], v = acc[i][1][e];
                if (LN) {                                               // value = rstd * acc - rstd * mean * colsum + bias: two FMAs
                    u = fmaf(rs[i], u, fmaf(nm[i], c1, b1));
                    v = fmaf(rs[i], v, fmaf(nm[i], c2, b2));
                } else {
                    u += b1;
                    v += b2;
                }
                hv[i][r] = silu_mul(u, v);
            }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            pk[i][q][0] = pack2(hv[i][0], hv[i][1]);
            pk[i][q][1] = pack2(hv[i][2], hv[i][3]);
            if (AUX) {                                                  // statistics of the ROUNDED outputs (what the next GEMM reads):
#pragma unroll                                                           // one v_dot2c_f32_bf16 per packed pair and moment, no unpacking
                for (int d = 0; d < 2; ++d) {
                    ssum[i] = dot2_bf16(pk[i][q][d], 0x3f803f80u, ssum[i]);
                    ssq[i] = dot2_bf16(pk[i][q][d], pk[i][q][d], ssq[i]);
                }
            }
        }
    }
    const __amdgpu_buffer_rsrc_t rc = make_rsrc((const __bf16*)p.C + (size_t)row0 * p.ldc + hbase);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const bool ok = colok && row0 + i * 32 + l31 < p.M && !CS_ABL(p, 8);      // dbg 8: timing ablation, every store masked
        const unsigned rowoff = (unsigned)((i * 32 + l31) * p.ldc + 8 * hf) * 2u;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            unsigned x0 = pk[i][2 * pr][0], x1 = pk[i][2 * pr][1], y0 = pk[i][2 * pr + 1][0], y1 = pk[i][2 * pr + 1][1];
            swap32(x0, y0);              // lower half: own group 2pr | upper's group 2pr  = columns 16pr .. 16pr+7
            swap32(x1, y1);              // upper half: lower's group 2pr+1 | own group 2pr+1 = columns 16pr+8 .. 16pr+15
            store16<CP>(u32x4{x0, x1, y0, y1}, rc, ok ? rowoff + 32u * pr : OOB);
        }
    }
    if (AUX) {
        // per-(32-hidden slice, row) partial (sum, sum of squares): slice = tn * 4 + wn; cs_ln_stats_finalize() pools the slices
        const size_t slice = (size_t)tn * 4 + wn;
        const __amdgpu_buffer_rsrc_t rst = make_rsrc(p.stats_part + (slice * p.M + row0) * 2);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const float s = both_halves(ssum[i]), q2 = both_halves(ssq[i]);
            const bool ok = colok && hf == 0 && row0 + i * 32 + l31 < p.M;
            store8<CP>(u32x2{__float_as_uint(s), __float_as_uint(q2)}, rst, ok ? (unsigned)(i * 32 + l31) * 8u : OOB);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- bf16 (+activation)
template <int ACT, bool LN, int CP, bool F8 = false, int FM = 4>
__device__ __forceinline__ void epi_bf16(const GemmArgs& p, const f32x16 (&acc)[FM][2], int lane, int row0, int colw, const EpiOps& eo) {
    const int l31 = lane & 31, hf = lane >> 5;
    float nm[FM], rs[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        rs[i] = (LN || F8) ? eo.rstd[i] : 1.f;            // F8: the row scale of the quantised A operand
        nm[i] = LN ? -eo.rstd[i] * eo.mean[i] : 0.f;
    }
    const __amdgpu_buffer_rsrc_t rc = make_rsrc((const __bf16*)p.C + (size_t)row0 * p.ldc + colw);
    const int sl0 = hf << 4;
    unsigned pk[2][FM][4][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float ov[FM][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sl = sl0 + ((j * 32 + 8 * q + r) << 2);
                const float b = lane_bcast(sl, eo.cb);
                const float c = (LN || F8) ? lane_bcast(sl, eo.cc) : 0.f;       // LN: folded column sum; F8: column scale of the B operand
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const float a = acc[i][j][q * 4 + r];
                    ov[i][r] = activate_s<ACT>(F8 ? fmaf(rs[i] * c, a, b) : LN ? fmaf(rs[i], a, fmaf(nm[i], c, b)) : a + b);
                }
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                pk[j][i][q][0] = pack2(ov[i][0], ov[i][1]);
                pk[j][i][q][1] = pack2(ov[i][2], ov[i][3]);
            }
        }
    }
    // stores in row-block order: the four 32-byte pieces of a row's 128-byte line (two column tiles x two group pairs) leave back to back
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const bool rowok = row0 + i * 32 + l31 < p.M && !CS_ABL(p, 8);        // dbg 8: timing ablation, every store masked
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = rowok && colw + j * 32 < p.N;                    // wave-uniform column test: N % 32 == 0
            const unsigned rowoff = (unsigned)((i * 32 + l31) * p.ldc + j * 32 + 8 * hf) * 2u;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                unsigned x0 = pk[j][i][2 * pr][0], x1 = pk[j][i][2 * pr][1], y0 = pk[j][i][2 * pr + 1][0], y1 = pk[j][i][2 * pr + 1][1];
                swap32(x0, y0);
                swap32(x1, y1);
                store16<CP>(u32x4{x0, x1, y0, y1}, rc, ok ? rowoff + 32u * pr : OOB);
            }
        }
    }
}

// ================================================================================================================ slab epilogues
// The register epilogues above store 32 bytes per row per instruction (quarter lines); the L2 merges the pieces, but every piece is a
// request of its own.  The slab variants send the packed results through a wave-private LDS slab -- the A / B ring slots the tile's
// last K iteration has just finished with (one extra workgroup barrier per tile), XOR-swizzled -- and read them back row-contiguous,
// so that 8 (bf16: 128 bytes), 4 (SwiGLU: 64 bytes) or 16 (fp32: 256 bytes) adjacent lanes store one whole row segment.
// Thanks to the transposed accumulators the slab is written with 8- / 16-byte DS stores (four consecutive columns per lane).

template <int ACT, bool LN, int CP>
__device__ __forceinline__ void epi_bf16_slab(const GemmArgs& p, const f32x16 (&acc)[4][2], int lane, int row0, int colw, const EpiOps& eo, char* slab) {
    const int l31 = lane & 31, hf = lane >> 5;
    float nm[4], rs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rs[i] = LN ? eo.rstd[i] : 1.f;
        nm[i] = LN ? -eo.rstd[i] * eo.mean[i] : 0.f;
    }
    const __amdgpu_buffer_rsrc_t rc = make_rsrc((const __bf16*)p.C + (size_t)row0 * p.ldc + colw);
    const int sl0 = hf << 4;
    const int piece = lane & 7, col = colw + piece * 8;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        // [64 rows][128 bytes]: 8-byte slot s of row r at r*128 + (s ^ ((r & 7) << 1)) * 8
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float bq[4], cq[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int sl = sl0 + ((j * 32 + 8 * q + r) << 2);
                    bq[r] = lane_bcast(sl, eo.cb);
                    cq[r] = LN ? lane_bcast(sl, eo.cc) : 0.f;
                }
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int i = half * 2 + ii, lrow = ii * 32 + l31;
                    float ov[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a = acc[i][j][q * 4 + r];
                        ov[r] = activate_s<ACT>(LN ? fmaf(rs[i], a, fmaf(nm[i], cq[r], bq[r])) : a + bq[r]);
                    }
                    const int s8 = (j * 8 + 2 * q + hf) ^ ((lrow & 7) << 1);
                    *(uint2*)(slab + lrow * 128 + s8 * 8) = make_uint2(pack2(ov[0], ov[1]), pack2(ov[2], ov[3]));
                }
            }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int lrow = (lane >> 3) + 8 * k, rrel = half * 64 + lrow;
            const uint4 w = *(const uint4*)(slab + lrow * 128 + ((piece ^ (lrow & 7)) << 4));
            const bool ok = row0 + rrel < p.M && col < p.N && !CS_ABL(p, 8);
            store16<CP>(u32x4{w.x, w.y, w.z, w.w}, rc, ok ? (unsigned)(rrel * p.ldc + piece * 8) * 2u : OOB);
        }
    }
}

template <bool LN, bool AUX, int CP>
__device__ __forceinline__ void epi_swiglu_slab(const GemmArgs& p, const f32x16 (&acc)[4][2], int lane, int row0, int tn, int wn, const EpiOps& eo,
                                                char* slab) {
    const int l31 = lane & 31, hf = lane >> 5;
    const int hbase = tn * 128 + wn * 32;
    const bool colok = hbase < p.group;
    float nm[4], rs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rs[i] = LN ? eo.rstd[i] : 1.f;
        nm[i] = LN ? -eo.rstd[i] * eo.mean[i] : 0.f;
    }
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    const int sl0 = hf << 4;
    // [128 rows][64 bytes]: 8-byte slot s of row r at r*64 + (s ^ (((r >> 1) & 3) << 1)) * 8
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float hv[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = q * 4 + r;
            const int sl = sl0 + ((8 * q + r) << 2);
            const float b1 = lane_bcast(sl, eo.cb), b2 = lane_bcast(sl + 128, eo.cb);
            float c1 = 0.f, c2 = 0.f;
            if (LN) { c1 = lane_bcast(sl, eo.cc); c2 = lane_bcast(sl + 128, eo.cc); }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float u = acc[i][0][e], v = acc[i][1][e];
                if (LN) {
                    u = fmaf(rs[i], u, fmaf(nm[i], c1, b1));
                    v = fmaf(rs[i], v, fmaf(nm[i], c2, b2));
                } else {
                    u += b1;
                    v += b2;
                }
                hv[i][r] = silu_mul(u, v);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned d0 = pack2(hv[i][0], hv[i][1]), d1 = pack2(hv[i][2], hv[i][3]);
            if (AUX) {
                ssum[i] = dot2_bf16(d1, 0x3f803f80u, dot2_bf16(d0, 0x3f803f80u, ssum[i]));
                ssq[i] = dot2_bf16(d1, d1, dot2_bf16(d0, d0, ssq[i]));
            }
            const int lrow = i * 32 + l31;
            const int s8 = (2 * q + hf) ^ (((lrow >> 1) & 3) << 1);
            *(uint2*)(slab + lrow * 64 + s8 * 8) = make_uint2(d0, d1);
        }
    }
    const __amdgpu_buffer_rsrc_t rc = make_rsrc((const __bf16*)p.C + (size_t)row0 * p.ldc + hbase);
    const int piece = lane & 3;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int lrow = (lane >> 2) + 16 * k;
        const uint4 w = *(const uint4*)(slab + lrow * 64 + ((piece ^ ((lrow >> 1) & 3)) << 4));
        const bool ok = colok && row0 + lrow < p.M && !CS_ABL(p, 8);
        store16<CP>(u32x4{w.x, w.y, w.z, w.w}, rc, ok ? (unsigned)(lrow * p.ldc + piece * 8) * 2u : OOB);
    }
    if (AUX) {
        const size_t slice = (size_t)tn * 4 + wn;
        const __amdgpu_buffer_rsrc_t rst = make_rsrc(p.stats_part + (slice * p.M + row0) * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s = both_halves(ssum[i]), q2 = both_halves(ssq[i]);
            const bool ok = colok && hf == 0 && row0 + i * 32 + l31 < p.M && !CS_ABL(p, 8);
            store8<CP>(u32x2{__float_as_uint(s), __float_as_uint(q2)}, rst, ok ? (unsigned)(i * 32 + l31) * 8u : OOB);
        }
    }
}

#ifdef CS_ABLATION_SWITCHES
// Epilogue timeline (env CS_GEMM_TRACE=<file>, tools/epi_trace.py): per workgroup and wave 16 x u64 of the LAST tile's residual epilogue -- the
// 100 MHz clock at entry and, per 32-row block, after (loads issued + slab written), after every outstanding memory access has landed
// (an extra s_waitcnt vmcnt(0): the trace build waits for a block's rows at once) and at the block's end.
__device__ unsigned long long* g_epi_trace = nullptr;
#define EPI_TRACE(slot)                                                                                                            \
    do {                                                                                                                           \
        if (g_epi_trace && (threadIdx.x & 63) == 0)                                                                                \
            g_epi_trace[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime();           \
    } while (0)
#define EPI_TRACE_WAIT() do { if (g_epi_trace) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#else
#define EPI_TRACE(slot) do { } while (0)
#define EPI_TRACE_WAIT() do { } while (0)
#endif
// fp32 residual through the slab: the accumulators of one 32-row block go [32 rows][256 bytes] (16-byte slot s of row r at
// r*256 + (s ^ (r & 15)) * 16, written with 16-byte DS stores while a lane still owns a row) and come back EIGHT lanes per row, eight
// consecutive columns per lane: every global access of the epilogue is a 16-byte-per-lane request over whole 128- / 256-byte row segments
// (hi / lo planes, bf16 copy: one 128-byte line per 8 lanes; fp32 rows: two).  Round 2 read the slab 16 lanes per row with 8-byte plane
// accesses -- twice the memory instructions for the same bytes, and 8-byte requests run at 0.54-0.70x the rate of 16-byte ones
// (MI355X_MICROARCH.md): the epilogue is bound by the CU's vector-memory path.
// SPLIT bit 0 / bit 1: the stream arrives / leaves as the two 16-bit planes (p.xb_out, p.lo) of y = bits(x) + 0x8000 (GemmArgs::split).
template <bool LN, bool AUX, int CP, bool F8 = false, int SPLIT = 0, int FM = 4>
__device__ __forceinline__ void epi_resid_slab(const GemmArgs& p, const f32x16 (&acc)[FM][2], int lane, int row0, int colw, int tn, int wn,
                                               const EpiOps& eo, char* slab) {
    static_assert(!(SPLIT & 2) || AUX, "a split stream leaves together with the row statistics");
    const int l31 = lane & 31, hf = lane >> 5;
    const int rrow = lane >> 3, piece = lane & 7, col = colw + piece * 8;
    const bool colok = col < p.N;                                           // N % 32 == 0: a lane's eight columns are inside or outside together
    const __amdgpu_buffer_rsrc_t rc = make_rsrc((SPLIT & 2) ? (const void*)p.xb_out : (const void*)((const float*)p.C + (size_t)row0 * p.ldc + colw));
    const __amdgpu_buffer_rsrc_t rx = make_rsrc((SPLIT & 1) ? (const void*)p.xb_out : (const void*)(p.extra + (size_t)row0 * p.ldc + colw));
    const __amdgpu_buffer_rsrc_t rb = make_rsrc((AUX || SPLIT) ? (const void*)(p.xb_out + (size_t)row0 * p.ldxb + colw) : (const void*)p.C);
    const __amdgpu_buffer_rsrc_t rl = make_rsrc(SPLIT ? (const void*)(p.lo + (size_t)row0 * p.ldxb + colw) : (const void*)p.A);
    const size_t slice = (size_t)tn * 4 + wn;
    const __amdgpu_buffer_rsrc_t rst = make_rsrc(AUX ? (const void*)(p.stats_part + (slice * p.M + row0) * 2) : (const void*)p.C);
    // The epilogue arithmetic runs AFTER the slab, where a lane owns 8 columns: the column constants are four 16-byte loads per tile and the
    // row's rstd and rstd * mean cross over with one ds_bpermute each per output row.  In front of the slab a lane owns a row and every
    // accumulator register needs its column's bias / column sum from another lane: 256 ds_bpermute per tile.  Same operations per element
    // as gemm.hip's epilogue_at.
    f32x4 cb4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, cc4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const unsigned off = colok ? (unsigned)(col + 4 * h) * 4u : OOB;
        if (p.bias) cb4[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(p.bias), off, 0, 0));
        if (LN || F8) cc4[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(p.ln_colsum), off, 0, 0));
    }
    EPI_TRACE(0);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        // the residual rows of the block are requested before the accumulators go through the slab
        f32x4 xin[4][2];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rrel = i * 32 + rrow + 8 * it;
            const bool ok = colok && row0 + rrel < p.M;
            if constexpr (SPLIT & 1) {
                const unsigned off = ok ? (unsigned)(rrel * p.ldxb + piece * 8) * 2u : OOB;
                const u32x4 h = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, off, 0, CP));
                const u32x4 l = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rl, off, 0, CP));
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    xin[it][w >> 1][2 * (w & 1)] = __uint_as_float(__builtin_amdgcn_perm(h[w], l[w], 0x05040100u) - 0x8000u);
                    xin[it][w >> 1][2 * (w & 1) + 1] = __uint_as_float(__builtin_amdgcn_perm(h[w], l[w], 0x07060302u) - 0x8000u);
                }
            } else {
                const unsigned off = ok ? (unsigned)(rrel * p.ldc + piece * 8) * 4u : OOB;
                xin[it][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, CP));
                xin[it][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off + 16u : OOB, 0, CP));
            }
        }
        const float rs_row = (LN || F8) ? eo.rstd[i] : 1.f, nm_row = LN ? -eo.rstd[i] * eo.mean[i] : 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 t;
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = acc[i][j][q * 4 + r];
                const int s16 = (j * 8 + 2 * q + hf) ^ (l31 & 15);
                *(f32x4*)(slab + l31 * 256 + s16 * 16) = t;
            }
        EPI_TRACE(1 + 3 * i);
        EPI_TRACE_WAIT();
        EPI_TRACE(2 + 3 * i);
        const int sel = lane & 7;
        float st_s = 0.f, st_q = 0.f;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int lrow = rrow + 8 * it, rrel = i * 32 + lrow;
            const bool ok = colok && row0 + rrel < p.M;
            float rs = 1.f, nm = 0.f;
            if (LN || F8) rs = lane_bcast(lrow << 2, rs_row);                 // rstd, -rstd * mean of row lrow live in lanes lrow, lrow + 32
            if (LN) nm = lane_bcast(lrow << 2, nm_row);
            f32x4 o[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 sv = *(const f32x4*)(slab + lrow * 256 + (((2 * piece + h) ^ (lrow & 15)) << 4));
                if (LN) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[h][t] = xin[it][h][t] + fmaf(rs, sv[t], fmaf(nm, cc4[h][t], cb4[h][t]));
                } else if (F8) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[h][t] = xin[it][h][t] + fmaf(rs * cc4[h][t], sv[t], cb4[h][t]);
                } else {
                    o[h] = xin[it][h] + (sv + cb4[h]);
                }
            }
            if constexpr (SPLIT & 2) {
                const unsigned off = ok ? (unsigned)(rrel * p.ldxb + piece * 8) * 2u : OOB;
                unsigned y[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) y[t] = __float_as_uint(o[t >> 2][t & 3]) + 0x8000u;
                store16<CP>(u32x4{__builtin_amdgcn_perm(y[1], y[0], 0x07060302u), __builtin_amdgcn_perm(y[3], y[2], 0x07060302u),
                                  __builtin_amdgcn_perm(y[5], y[4], 0x07060302u), __builtin_amdgcn_perm(y[7], y[6], 0x07060302u)}, rb, off);
                store16<CP>(u32x4{__builtin_amdgcn_perm(y[1], y[0], 0x05040100u), __builtin_amdgcn_perm(y[3], y[2], 0x05040100u),
                                  __builtin_amdgcn_perm(y[5], y[4], 0x05040100u), __builtin_amdgcn_perm(y[7], y[6], 0x05040100u)}, rl, off);
            } else {
                const unsigned off = ok ? (unsigned)(rrel * p.ldc + piece * 8) * 4u : OOB;
                store16<CP>(__builtin_bit_cast(u32x4, o[0]), rc, off);
                store16<CP>(__builtin_bit_cast(u32x4, o[1]), rc, ok ? off + 16u : OOB);
            }
            if (AUX) {
                if constexpr (!(SPLIT & 2))
                    store16<CP>(u32x4{pack2(o[0][0], o[0][1]), pack2(o[0][2], o[0][3]), pack2(o[1][0], o[1][1]), pack2(o[1][2], o[1][3])}, rb,
                                ok ? (unsigned)(rrel * p.ldxb + piece * 8) * 2u : OOB);
                // a row's 64 columns sit in 8 adjacent lanes (DPP butterfly); lane (lane & 7) == it keeps iteration it's row, so the
                // 32 rows of the block leave in one 256-byte store below
                float ps = 0.f, pq = 0.f;
                if (colok) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) { ps += o[t >> 2][t & 3]; pq = fmaf(o[t >> 2][t & 3], o[t >> 2][t & 3], pq); }
                }
                const float a = sum_lanes8(ps), b = sum_lanes8(pq);
                if (sel == it) { st_s = a; st_q = b; }
            }
        }
        if (AUX) {
            const int rrel = i * 32 + rrow + 8 * sel;
            const bool ok = sel < 4 && row0 + rrel < p.M && colw < p.N;
            store8<CP>(u32x2{__float_as_uint(st_s), __float_as_uint(st_q)}, rst, ok ? (unsigned)rrel * 8u : OOB);
        }
        EPI_TRACE(3 + 3 * i);
        __builtin_amdgcn_sched_barrier(0);       // keep the next block's residual loads (32 registers) out of this block
    }
}

// store instructions one epilogue issues per wave (exact: masked lanes keep their instruction, see OOB)
// (a count BELOW the real one would be safe -- a wait then retires more than it needs to --, one above it is not)
template <int EPI, bool AUX, bool SLAB, int SPLIT = 0, int FM = 4>
constexpr int epi_stores() {
    if (EPI == EPI_SWIGLU_BF16) return FM * (2 + (AUX ? 1 : 0));
    if (EPI == EPI_RESID_F32) return FM * (8 + (AUX ? ((SPLIT & 2) ? 0 : 4) + 1 : 0));      // per row block: four row groups x two fp32 or two plane stores (+ the bf16 copy), + statistics
    return FM * 4;
}

// cache policy of the operand DMA (gfx950 aux bits: 0 default | 1 = sc0 | 2 = nt | 16 = sc1); A/B experiments: -DCS_DMA_AUX_A=.. -DCS_DMA_AUX_B=..
#ifndef CS_DMA_AUX_A
#define CS_DMA_AUX_A 0
#endif
#ifndef CS_DMA_AUX_B
#define CS_DMA_AUX_B 0
#endif
#define CS_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

inline bool getenv_flag(const char* name) {       // A/B switch, read at every launch (~0.1 us) so that a test can flip it inside one process
    const char* v = getenv(name);
    return v && v[0] && v[0] != '0';
}

// EPI: EPI_BF16 / EPI_GELU_BF16 / EPI_QGELU_BF16 / EPI_SWIGLU_BF16 / EPI_RESID_F32 (the latter with LN = folded LayerNorm, i.e. epilogue 6)
// F8: A and B hold e4m3 bytes (K = bytes per row, a multiple of 128: one K tile = 128 contraction steps in the same 128-byte LDS rows),
// contracted with the block-scaled MFMA v_mfma_scale_f32_32x32x64_f8f6f4 at unit block scales (2x the bf16 MFMA rate at half the operand
// bytes); the per-row scale of A (p.ln_rstd) and per-row scale of B (p.ln_colsum) are applied in the epilogue.
// BMT: rows per output tile, 256 or 192 (round 4).  192 = six 32-row blocks, three per wave: launches whose 256-row tiles fill only part
// of the chip (the student's N = 768 GEMMs at M = 12 608: 150 tiles on 256 CUs) run 198 tiles of 3/4 the work instead.
template <int EPI, bool LN, bool AUX, bool SLAB, bool F8 = false, int SPLIT = 0, int BMT = 256>
__global__ __launch_bounds__(512) void gemm_stream_kernel(GemmArgs p) {
    static_assert(BMT == 256 || (BMT == 192 && !F8 && SPLIT == 0 && !LN && !AUX), "192-row tiles: the training schedule's plain epilogues");
    static_assert(SPLIT == 0 || (EPI == EPI_RESID_F32 && LN && !F8), "the split stream belongs to the folded-LayerNorm residual epilogue");
    static_assert(SLAB || EPI != EPI_RESID_F32, "the fp32 residual epilogue exists in the slab form only (whole-line traffic)");
    static_assert(!F8 || (!LN && !AUX && (EPI == EPI_BF16 || EPI == EPI_RESID_F32)), "fp8 operands: bf16 and fp32-residual outputs");
    constexpr int ES = F8 ? 1 : 2;               // operand element size in bytes
    constexpr int CP = 0;                      // default cache policy: the L2 must merge partial-line stores (nt / sc1 measured slower)
    constexpr bool SWI = EPI == EPI_SWIGLU_BF16, RES = EPI == EPI_RESID_F32;
    static_assert(SWI || RES || epi_is_bf16(EPI), "register epilogues");
    constexpr int BM = BMT, BN = 256, WN = 4, TM = BM / 2, TN = 64, FM = TM / 32, FN = 2;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int NPA = BM / 32;                 // DMA pieces of the A tile per A-loader wave (ROLES)
    constexpr int NM = FM * FN, NR = FM + FN;    // MFMAs / fragment reads per k-step and wave (8 / 6; 192-row tiles: 6 / 5)
    constexpr int S = epi_stores<EPI, AUX, SLAB, SPLIT, FM>() < 55 ? epi_stores<EPI, AUX, SLAB, SPLIT, FM>() : 55;     // vmcnt is a 6-bit counter; fewer = safe
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int hf = lane >> 5, l31 = lane & 31;
    char* const b_ring = smem + 3 * A_BYTES;
    const int ntiles = p.tiles_m * p.tiles_n, ktiles = p.K / (F8 ? 128 : BK);
    const int a_base = ((wm * TM + l31) >> 1) << 8, b_base = ((wn * TN + l31) >> 1) << 8;
    const int par8 = (l31 & 1) << 3, sw = l31 >> 1;
    const int G = gridDim.x;
#ifndef CS_NO_DMA_ROLES
    // Round 4: the two waves of a SIMD (w and w + 4) take different DMA roles.  Waves 0-3 put the whole B tile of K tile g+1 in flight
    // during the first two k-steps of iteration g (8 pieces each), waves 4-7 the whole A tile of K tile g+2 during the last two.  Issuing
    // a DMA piece stalls its wave for ~100 cycles beside LDS reads; with every wave issuing two pieces in every k-step (round 3) the two
    // waves of a SIMD stall together and the matrix pipe idles, now the stalled wave's partner has a k-step of MFMAs and LDS reads only.
    // Same tiles, same accumulation order: bit-identical outputs; step 704.7 -> 717.5 images/s on one box (profiles/r04_c_dma_roles.md).
    // (bf16 kernels.  The e4m3 kernels keep round 3's issue: with the roles' offset registers beside a 48-register fragment set they spilled 36
    // VGPRs into the K loop -- RegionCLIP L/14-336 fp8 forward + dgrad 451.0 / 452.7 -> 456.4 / 456.8 images/s without, profiles/r04_v_f8_roles.txt)
    constexpr bool ROLES = !F8;
#else
    constexpr bool ROLES = false;
#endif
    const bool role_a = ROLES && wave >= 4;
#ifndef CS_NO_MID_BARRIER
    // Round 4.  The per-K-tile barrier sits between k-steps 2 and 3 of the tile (register epilogues only): when a wave leaves it, the fragments of
    // k-step 3 are in its registers, so the eight MFMAs of that k-step start at once and the first fragments of the NEXT K tile are read
    // behind them -- no wave ever sits behind a barrier with nothing but LDS latency in front of its next MFMA.  Operand lead: B(g+2)
    // goes out in k-step 3 of iteration g (its slot, B(g)'s, is free once every wave has passed the barrier of iteration g), A(g+2) in
    // k-steps 1-2.  Bit-identical outputs; q|k|v 1500 -> 1473 us, W1|W2 2308 -> 2288 us, step +0.4 % (profiles/r04_f_mid_barrier.txt).
    // (Slab epilogues keep the top-of-tile barrier.  Their form of this schedule -- B(g+2) of an output tile's last K tile issued behind a
    // barrier that follows the epilogue, because it lands in the B slab -- was built in round 4: with the fragment sets live across the K
    // tiles the fp32-residual kernels, already at 241-252 VGPRs for their epilogue, spill 24-29 registers into the K loop.)
    constexpr bool MID = ROLES && !SLAB && !F8;
#else
    constexpr bool MID = false;
#endif

    // DMA cursors: position (tile, K tile) of the next A / B operand tile to put in flight.  A piece's source address is
    // wave-uniform base (operand + first row of the tile, advanced by 128 bytes per K tile: SGPRs) + a per-lane 32-bit byte offset
    // that is constant for the whole tile -> no vector address arithmetic in the K loop.
    unsigned avoff[ROLES ? 8 : 4], bvoff[ROLES ? 1 : 4];           // ROLES: avoff holds the 8 pieces of the wave's ONE operand
    const char *a_src = nullptr, *b_src = nullptr;
    int a_tile = blockIdx.x, a_kt = 0, b_tile = blockIdx.x, b_kt = 0;
    auto set_a = [&](int tile) {
        int tm, tn;
        tile_of_id(p, tile, ntiles, tm, tn);
        a_src = (const char*)p.A + (size_t)tm * BM * p.lda * ES;
#pragma unroll
        for (int i = 0; i < (ROLES ? NPA : 4); ++i) {
            int tr, chk;
            lane_source(ROLES ? (wave & 3) * NPA + i : wave * 4 + i, lane, tr, chk);
            avoff[i] = (unsigned)(min(tr, p.M - 1 - tm * BM) * p.lda * ES + chk * 16);
        }
    };
    auto set_b = [&](int tile) {
        int tm, tn;
        tile_of_id(p, tile, ntiles, tm, tn);
        if (SWI) {         // tile rows [w*64 + jj*32 + t] <- weight row jj*Hd + (tn*128 + w*32 + t): x1 | x2 of one hidden unit share lane and register
            b_src = (const char*)p.B + (size_t)tn * (BN / 2) * p.ldb * ES;
        } else {
            b_src = (const char*)p.B + (size_t)tn * BN * p.ldb * ES;
        }
#pragma unroll
        for (int i = 0; i < (ROLES ? 8 : 4); ++i) {
            int tr, chk;
            lane_source(ROLES ? (wave & 3) * 8 + i : wave * 4 + i, lane, tr, chk);
            int rel;
            if (SWI) {
                const int hrel = (tr >> 6) * 32 + (tr & 31);                                  // hidden unit relative to tn*128
                rel = ((tr >> 5) & 1) * p.group + min(hrel, p.group - 1 - tn * (BN / 2));
            } else {
                rel = min(tr, p.N - 1 - tn * BN);
            }
            (ROLES ? avoff : bvoff)[i] = (unsigned)(rel * p.ldb * ES + chk * 16);
        }
    };
    // Past the workgroup's last tile a cursor keeps cycling over that tile's K tiles: the DMA of the last two iterations then re-reads
    // valid memory into slots nobody reads any more, and the K loop issues its eight pieces unconditionally -- no branch inside it, one
    // basic block for the scheduler, fixed wait counts.
    auto adv_a = [&]() {
        if (++a_kt == ktiles) {
            a_kt = 0;
            if (a_tile + G < ntiles) {
                a_tile += G;
                set_a(a_tile);
            }
        }
    };
    auto adv_b = [&]() {
        if (++b_kt == ktiles) {
            b_kt = 0;
            if (b_tile + G < ntiles) {
                b_tile += G;
                set_b(b_tile);
            }
        }
    };
#define ISSUE_A(X, SLOT)                                                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src + (size_t)a_kt * (BK * 2) + avoff[X]), \
                                     (__attribute__((address_space(3))) void*)(smem + (SLOT) * A_BYTES + (wave * 4 + (X)) * 1024), 16, 0, CS_DMA_AUX_A)
#define ISSUE_B(X, SLOT)                                                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src + (size_t)b_kt * (BK * 2) + bvoff[X]), \
                                     (__attribute__((address_space(3))) void*)(b_ring + (SLOT) * B_BYTES + (wave * 4 + (X)) * 1024), 16, 0, CS_DMA_AUX_B)

    // ROLES: piece X of the wave's own operand tile (B: 8 pieces per wave, A: NPA) -- LDS rows (first piece + X) * 4 .. + 3 of the slot
#define ISSUE_RA(X, SLOT)                                                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src + (size_t)a_kt * (BK * 2) + avoff[X]), \
                                     (__attribute__((address_space(3))) void*)(smem + (SLOT) * A_BYTES + ((wave & 3) * NPA + (X)) * 1024), 16, 0, CS_DMA_AUX_A)
#define ISSUE_RB(X, SLOT)                                                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src + (size_t)b_kt * (BK * 2) + avoff[X]), \
                                     (__attribute__((address_space(3))) void*)(b_ring + (SLOT) * B_BYTES + ((wave & 3) * 8 + (X)) * 1024), 16, 0, CS_DMA_AUX_B)

    if constexpr (ROLES) {
        if (role_a) {
            set_a(a_tile);
#pragma unroll
            for (int x = 0; x < NPA; ++x) ISSUE_RA(x, 0);
            adv_a();
#pragma unroll
            for (int x = 0; x < NPA; ++x) ISSUE_RA(x, 1);
            adv_a();
        } else {
            set_b(b_tile);
#pragma unroll
            for (int x = 0; x < 8; ++x) ISSUE_RB(x, 0);
            adv_b();
            if constexpr (MID) {
#pragma unroll
                for (int x = 0; x < 8; ++x) ISSUE_RB(x, 1);
                adv_b();
            }
        }
        if constexpr (MID) {               // K tile 0 landed everywhere before the first fragment read (both roles: one tile stays in flight)
            if (role_a) CS_VMCNT(NPA);
            else CS_VMCNT(8);
            __builtin_amdgcn_s_barrier();
        }
    } else {
        set_a(a_tile);
        set_b(b_tile);
#pragma unroll
        for (int x = 0; x < 4; ++x) ISSUE_A(x, 0);
        adv_a();
#pragma unroll
        for (int x = 0; x < 4; ++x) ISSUE_B(x, 0);
        adv_b();
#pragma unroll
        for (int x = 0; x < 4; ++x) ISSUE_A(x, 1);
        adv_a();
    }

    int curA = 0, gpar = 0;                  // A slot of K tile g, parity of g
    bool after_epi = false;
    for (int tile = blockIdx.x; tile < ntiles; tile += G) {
        int tm, tn;
        tile_of_id(p, tile, ntiles, tm, tn);
        const int row0 = tm * BM + wm * TM, colw = tn * BN + wn * TN;
        f32x16 acc[FM][FN];
        EpiOps eo;
        // epilogue operands: LayerNorm statistics of the rows of the wave's four blocks (lane = row), constants of its 64 columns (lane = column)
        auto load_epi_ops = [&](int ln) {
            const int l31 = ln & 31, hf = ln >> 5, lane = ln;
#pragma unroll
            for (int i = 0; i < 4; ++i) eo.mean[i] = eo.rstd[i] = 0.f;
            eo.cb = eo.cc = 0.f;
            if (LN || F8) {
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int r = min(row0 + i * 32 + l31, p.M - 1);
                    if (LN) eo.mean[i] = p.ln_mean[r];
                    eo.rstd[i] = p.ln_rstd[r];
                }
            }
            int co;
            if (SWI) co = hf * p.group + min(tn * 128 + wn * 32 + l31, p.group - 1);
            else co = min(colw + lane, p.N - 1);
            if (!RES) {                                    // the residual epilogue reads its column constants 8 per lane behind the slab
                if (p.bias) eo.cb = p.bias[co];
                if (LN || F8) eo.cc = p.ln_colsum[co];
            }
        };

        // ZERO: first K tile of the output tile -- its first k-step accumulates onto the inline constant 0 (no 128 v_mov per tile)
        auto ktile = [&](auto zero_tag, auto role_tag) {
            constexpr bool ZERO = decltype(zero_tag)::value;
            constexpr int ROLE = decltype(role_tag)::value;         // 0: every wave issues 4 + 4 pieces; 1: B loader (waves 0-3); 2: A loader (4-7)
            // In issue order this wave's pending ops are ... A(g), B(g), A(g+1) [, the previous epilogue's S stores]: everything older
            // than A(g+1) must have landed; vmcnt retires in order, so the stores (newest) may stay in flight as well.
            // ROLES: a B loader's pending ops are B(g) [, stores]; an A loader's A(g), A(g+1) [, stores] -- 8 pieces each.
            constexpr int KEEP = ROLE == 0 ? 4 : ROLE == 1 ? 0 : NPA;
            constexpr int SS = S + KEEP > 63 ? 63 - KEEP : S;
            if (after_epi) CS_VMCNT(KEEP + SS);
            else CS_VMCNT(KEEP);
            after_epi = false;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!CS_ABL(p, 2))                  // build with CS_EXTRA_FLAGS=-DCS_ABLATION_SWITCHES for tools/barrier_cost.py: the run-time test costs 0.35 % of the step
                __builtin_amdgcn_s_barrier();     // K tile g landed everywhere; A slot (g+2)%3 and B slot (g+1)&1 are free
            const char* la = smem + curA * A_BYTES + a_base;
            const char* lb = b_ring + gpar * B_BYTES + b_base;
            const int slot_a2 = curA == 0 ? 2 : curA - 1, slot_b1 = gpar ^ 1;
            if constexpr (F8) {
                typedef __attribute__((ext_vector_type(8))) int i32x8;
                typedef __attribute__((ext_vector_type(4))) int i32x4;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {               // 64 contraction steps each: 32 bytes per lane and operand row
                    const int c0 = 4 * s2 + 2 * hf;
                    const int off0 = ((par8 | c0) ^ sw) << 4, off1 = ((par8 | (c0 + 1)) ^ sw) << 4;
                    i32x8 a8[FM], b8[FN];
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        const i32x4 lo = *(const i32x4*)(la + i * (16 * 256) + off0), hi = *(const i32x4*)(la + i * (16 * 256) + off1);
                        a8[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        const i32x4 lo = *(const i32x4*)(lb + j * (16 * 256) + off0), hi = *(const i32x4*)(lb + j * (16 * 256) + off1);
                        b8[j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
                    if constexpr (ROLE == 1) {              // B loaders: the whole B tile behind the first half's fragment reads
                        if (s2 == 0) {
#pragma unroll
                            for (int x = 0; x < 8; ++x) ISSUE_RB(x, slot_b1);
                        }
                    } else if constexpr (ROLE == 2) {       // A loaders: the whole A tile behind the second half's
                        if (s2 == 1) {
#pragma unroll
                            for (int x = 0; x < NPA; ++x) ISSUE_RA(x, slot_a2);
                        }
                    } else if (s2 == 0) {
                        ISSUE_B(0, slot_b1); ISSUE_B(1, slot_b1); ISSUE_B(2, slot_b1); ISSUE_B(3, slot_b1);
                    } else {
                        ISSUE_A(0, slot_a2); ISSUE_A(1, slot_a2); ISSUE_A(2, slot_a2); ISSUE_A(3, slot_a2);
                    }
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j) {     // e4m3 x e4m3, unit (2^0) block scales
                            if (ZERO && s2 == 0) {
                                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[j], a8[i], z, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                            } else {
                                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[j], a8[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                            }
                        }
                }
            } else {
                // fragments of k-step ks+1 are requested before the MFMAs of k-step ks: the wave's own LDS latency hides behind its own
                // MFMAs (two register sets), not only behind the other wave of the SIMD
                bf16x8 a[2][FM], b[2][FN];
                auto frags = [&](int ks, int buf) {
                    const int off = ((par8 | (ks * 2 + hf)) ^ sw) << 4;
#pragma unroll
                    for (int i = 0; i < FM; ++i) a[buf][i] = *(const bf16x8*)(la + i * (16 * 256) + off);
#pragma unroll
                    for (int j = 0; j < FN; ++j) b[buf][j] = *(const bf16x8*)(lb + j * (16 * 256) + off);
                };
                frags(0, 0);
#ifndef CS_NO_SGB
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#endif
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int cb = ks & 1;
                    if (ks < 3) frags(ks + 1, cb ^ 1);
                    if constexpr (ROLE == 1) {     // B(g+1): all 8 pieces in k-steps 0 and 1
                        if (ks == 0) { ISSUE_RB(0, slot_b1); ISSUE_RB(1, slot_b1); ISSUE_RB(2, slot_b1); ISSUE_RB(3, slot_b1); }
                        if (ks == 1) { ISSUE_RB(4, slot_b1); ISSUE_RB(5, slot_b1); ISSUE_RB(6, slot_b1); ISSUE_RB(7, slot_b1); }
                    } else if constexpr (ROLE == 2) {     // A(g+2): all NPA pieces in k-steps 2 and 3
                        if (ks == 2) {
#pragma unroll
                            for (int x = 0; x < NPA / 2; ++x) ISSUE_RA(x, slot_a2);
                        }
                        if (ks == 3) {
#pragma unroll
                            for (int x = NPA / 2; x < NPA; ++x) ISSUE_RA(x, slot_a2);
                        }
                    } else if (ks < 2) {           // two DMA pieces per k-step: B(g+1) first, then A(g+2)
                        if ((ks & 1) == 0) { ISSUE_B(0, slot_b1); ISSUE_B(1, slot_b1); } else { ISSUE_B(2, slot_b1); ISSUE_B(3, slot_b1); }
                    } else {
                        if ((ks & 1) == 0) { ISSUE_A(0, slot_a2); ISSUE_A(1, slot_a2); } else { ISSUE_A(2, slot_a2); ISSUE_A(3, slot_a2); }
                    }
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j) {
                            if (ZERO && ks == 0) {
                                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[cb][j], a[cb][i], z, 0, 0, 0);
                            } else {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[cb][j], a[cb][i], acc[i][j], 0, 0, 0);
                            }
                        }
                    // issue order of this k-step: one LDS read (of the next k-step's fragments) or one DMA piece behind every MFMA
#ifndef CS_NO_SGB
                    const bool DMA4 = (ROLE == 1 && ks < 2) || (ROLE == 2 && ks >= 2);      // this k-step carries DN = 4 (A at 192 rows: 3) pieces of this wave
                    constexpr int DN = ROLE == 2 ? NPA / 2 : 4;
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (ROLE == 0) {
                            if (m < NR) {
                                if (ks < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            } else {
                                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                            }
                        } else {
                            if (m < NR && ks < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            if (DMA4 && ks < 3 && m == NM - 2) __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);     // behind the last two MFMAs: 2 + 2 (2 + 1)
                            if (DMA4 && ks < 3 && m == NM - 1) __builtin_amdgcn_sched_group_barrier(0x010, DN - 2, 0);
                            if (DMA4 && ks == 3 && m >= 2 && m < 2 + DN) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // no reads in the last k-step
                        }
                    }
#endif
                }
            }
            if (ROLE != 2) adv_b();
            if (ROLE != 1) adv_a();
            curA = curA == 2 ? 0 : curA + 1;
            gpar ^= 1;
        };
        // ---- MID: barrier between k-steps 2 and 3 (see above); fragment sets live across the K tiles of the output tile
        bf16x8 fa[2][MID ? FM : 1], fb[2][MID ? FN : 1];
        auto ktile_mid = [&](auto zero_tag, auto role_tag) {
            constexpr bool ZERO = decltype(zero_tag)::value;
            constexpr int ROLE = decltype(role_tag)::value;         // 1: B loader (waves 0-3), 2: A loader (waves 4-7)
            const int nextA = curA == 2 ? 0 : curA + 1;
            const char* la = smem + curA * A_BYTES + a_base;
            const char* lb = b_ring + gpar * B_BYTES + b_base;
            const char* la_n = smem + nextA * A_BYTES + a_base;
            const char* lb_n = b_ring + (gpar ^ 1) * B_BYTES + b_base;
            const int slot_a2 = curA == 0 ? 2 : curA - 1, slot_b2 = gpar;      // A(g+2) -> slot of A(g-1); B(g+2) -> slot of B(g)
            auto frags = [&](const char* pa, const char* pb, int ks, int buf) {
                const int off = ((par8 | (ks * 2 + hf)) ^ sw) << 4;
#pragma unroll
                for (int i = 0; i < FM; ++i) fa[buf][i] = *(const bf16x8*)(pa + i * (16 * 256) + off);
#pragma unroll
                for (int j = 0; j < FN; ++j) fb[buf][j] = *(const bf16x8*)(pb + j * (16 * 256) + off);
            };
            auto mfmas = [&](int cb, bool zero) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        if (zero) {
                            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cb][j], fa[cb][i], z, 0, 0, 0);
                        } else {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cb][j], fa[cb][i], acc[i][j], 0, 0, 0);
                        }
                    }
            };
            if (ZERO) {                    // first K tile of an output tile: its first fragments were not prefetched across the epilogue
                frags(la, lb, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            }
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const int cb = ks & 1;
                frags(la, lb, ks + 1, cb ^ 1);
                // A(g+2) goes out in k-steps 1 and 2: away from the B loaders' k-step 3 burst, which runs into k-step 0 of their SIMD partners
                // (k-steps 0 and 1: 720.9 -> 724.2 images/s same box, dominant kernel 2360 -> 2331 us; profiles/r04_o_a_pieces_late.txt)
                constexpr int KA = 1;
                if constexpr (ROLE == 2) {                 // A(g+2): its NPA pieces in k-steps KA and KA + 1
                    if (ks == KA) {
#pragma unroll
                        for (int x = 0; x < NPA / 2; ++x) ISSUE_RA(x, slot_a2);
                    }
                    if (ks == KA + 1) {
#pragma unroll
                        for (int x = NPA / 2; x < NPA; ++x) ISSUE_RA(x, slot_a2);
                    }
                }
                mfmas(cb, ZERO && ks == 0);
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (m < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (ROLE == 2 && (ks == KA || ks == KA + 1) && m == NM - 2) __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
                    if (ROLE == 2 && (ks == KA || ks == KA + 1) && m == NM - 1) __builtin_amdgcn_sched_group_barrier(0x010, NPA / 2 - 2, 0);
                }
            }
            // K tile g+1 landed everywhere, every wave is done reading the LDS images of K tile g.  Pending ops of a B loader: B(g+1)
            // [, the stores of an epilogue that ran since]; of an A loader: A(g+1) [, those stores], A(g+2).
            {
                constexpr int KEEP = ROLE == 1 ? 0 : NPA;
                constexpr int SS = S + KEEP > 63 ? 63 - KEEP : S;
                if (after_epi) CS_VMCNT(KEEP + SS);
                else CS_VMCNT(KEEP);
                after_epi = false;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            // k-step 0 of K tile g+1 (past the output tile's last K tile: read, not used) and -- B loaders -- all 8 pieces of B(g+2), in the
            // order they are to issue: the compiler keeps LDS reads and LDS-DMA writes in program order (it cannot tell their slots apart)
            {
                const int off = ((par8 | hf) ^ sw) << 4;
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    if (x < FM) fa[0][x] = *(const bf16x8*)(la_n + x * (16 * 256) + off);
                    else if (x < NR) fb[0][x - FM] = *(const bf16x8*)(lb_n + (x - FM) * (16 * 256) + off);
                    if constexpr (ROLE == 1) ISSUE_RB(x, slot_b2);
                }
            }
            mfmas(1, false);
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (m < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (ROLE == 1 && m < NM - 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);          // 8 pieces over NM MFMAs:
                if (ROLE == 1 && m == NM - 1) __builtin_amdgcn_sched_group_barrier(0x010, 1 + 8 - NM, 0);   // the rest behind the last one
            }
            if (ROLE == 1) adv_b();
            else adv_a();
            curA = nextA;
            gpar ^= 1;
        };
        if constexpr (MID) {
            if (role_a) {
                ktile_mid(std::true_type{}, std::integral_constant<int, 2>{});
                for (int kt = 1; kt < ktiles; ++kt) ktile_mid(std::false_type{}, std::integral_constant<int, 2>{});
            } else {
                ktile_mid(std::true_type{}, std::integral_constant<int, 1>{});
                for (int kt = 1; kt < ktiles; ++kt) ktile_mid(std::false_type{}, std::integral_constant<int, 1>{});
            }
        } else if constexpr (ROLES) {
            if (role_a) {
                ktile(std::true_type{}, std::integral_constant<int, 2>{});
                for (int kt = 1; kt < ktiles; ++kt) ktile(std::false_type{}, std::integral_constant<int, 2>{});
            } else {
                ktile(std::true_type{}, std::integral_constant<int, 1>{});
                for (int kt = 1; kt < ktiles; ++kt) ktile(std::false_type{}, std::integral_constant<int, 1>{});
            }
        } else {
            ktile(std::true_type{}, std::integral_constant<int, 0>{});
            for (int kt = 1; kt < ktiles; ++kt) ktile(std::false_type{}, std::integral_constant<int, 0>{});
        }
        if (CS_ABL(p, 4)) continue;            // timing ablation (tools/gemm_bench.py): no epilogue, results are wrong
        after_epi = true;
        // Everything the epilogue derives from the lane id (swizzled slab addresses, store offsets, ds_bpermute sources: ~80 values) is
        // loop-invariant; hoisted out of the tile loop it would live -- spilled -- across the whole MFMA loop.  An opaque copy of the
        // lane id keeps those computations inside the epilogue.
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        if constexpr (SLAB) {
            // the slabs alias the A / B slots K tile g has just been read from: every wave must be done with them.  The barrier at the top
            // of the next K iteration then keeps that iteration's DMA (which refills exactly these slots) behind all slab traffic.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // 8 KB per wave: four in the B slot; four (192-row tiles: three, the fourth behind the rings -- 3 x 24 + 2 x 32 = 136 of 160 KB) in the A slot
            char* const slab = wave >= 4 ? b_ring + (gpar ^ 1) * B_BYTES + (wave - 4) * 8192
                               : (BM == 192 && wave == 3) ? b_ring + 2 * B_BYTES
                                                          : smem + (curA == 0 ? 2 : curA - 1) * A_BYTES + wave * 8192;
            load_epi_ops(lane_e);
#ifndef CS_NO_KERNARG_RELOAD
            constexpr bool RELOAD = RES && LN && AUX;
#else
            constexpr bool RELOAD = false;         // A/B switch (tools/build_variant.sh): the round-4 form with its SGPR spills
#endif
            if constexpr (RELOAD) {
                // The folded-LayerNorm residual epilogue with statistics uses eleven pointers of GemmArgs (seven buffer descriptors): held in
                // SGPRs across the K loop they were spilled to VGPR lanes (8-32 SGPRs, and 3 VGPRs to scratch in the fp32-stream form).  The
                // epilogue reads the arguments from the kernarg segment again instead -- scalar loads that hit the constant cache --
                // through a pointer the compiler cannot see through, so nothing of it is live in the loop.
                const __attribute__((address_space(4))) GemmArgs* pk = (const __attribute__((address_space(4))) GemmArgs*)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(pk));
                GemmArgs pe;                                     // field by field: scalar (s_load) reads of what the epilogue uses, the rest is dead
                pe.A = pk->A; pe.C = pk->C; pe.bias = pk->bias; pe.extra = pk->extra; pe.M = pk->M; pe.N = pk->N; pe.ldc = pk->ldc;
                pe.ln_colsum = pk->ln_colsum; pe.stats_part = pk->stats_part; pe.xb_out = pk->xb_out; pe.ldxb = pk->ldxb; pe.lo = pk->lo;
                pe.dbg = pk->dbg;
                epi_resid_slab<LN, AUX, CP, F8, SPLIT, FM>(pe, acc, lane_e, row0, colw, tn, wn, eo, slab);
            } else if constexpr (RES) {
                epi_resid_slab<LN, AUX, CP, F8, SPLIT, FM>(p, acc, lane_e, row0, colw, tn, wn, eo, slab);
            } else {
                static_assert(FM == 4, "the slab forms of the bf16 / SwiGLU epilogues exist for 256-row tiles");
                if constexpr (SWI) epi_swiglu_slab<LN, AUX, CP>(p, acc, lane_e, row0, tn, wn, eo, slab);
                else epi_bf16_slab<epi_act(EPI), LN, CP>(p, acc, lane_e, row0, colw, eo, slab);
            }
        } else {
            load_epi_ops(lane_e);
            if constexpr (SWI) epi_swiglu<LN, AUX, CP, FM>(p, acc, lane_e, row0, tn, wn, eo);
            else epi_bf16<epi_act(EPI), LN, CP, F8, FM>(p, acc, lane_e, row0, colw, eo);
        }
    }
    CS_VMCNT(0);      // the last two iterations' operand DMA (dead data, see adv_a) must not outlive the workgroup's LDS allocation
#undef ISSUE_A
#undef ISSUE_B
#undef ISSUE_RA
#undef ISSUE_RB
}

template <int EPI, bool LN, bool AUX, bool SLAB, int BMT = 256>
int launch_stream_v(const GemmArgs& a, unsigned grid, hipStream_t stream) {
    constexpr size_t lds = 160 * 1024;
    static bool once = ((void)hipFuncSetAttribute((const void*)gemm_stream_kernel<EPI, LN, AUX, SLAB, false, 0, BMT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL((gemm_stream_kernel<EPI, LN, AUX, SLAB, false, 0, BMT>), dim3(grid), dim3(512), lds, stream, a);
    CS_LAUNCH_CHECK();
    return 0;
}

// fp32 residual epilogues: always through the LDS slab (measured 6-25 % faster than quarter-line register stores).  bf16 / SwiGLU:
// straight from registers by default, through the slab with flags bit 12 (a.dbg & 1) -- a tie on the tower shapes, A/B switch.
template <int EPI, bool LN, bool AUX>
int launch_stream_t(const GemmArgs& a, unsigned grid, hipStream_t stream) {
    if constexpr (EPI == EPI_RESID_F32) return launch_stream_v<EPI, LN, AUX, true>(a, grid, stream);
    else return (a.dbg & 1) ? launch_stream_v<EPI, LN, AUX, true>(a, grid, stream) : launch_stream_v<EPI, LN, AUX, false>(a, grid, stream);
}

template <bool AUX, int SPLIT>
int launch_stream_split(const GemmArgs& a, unsigned grid, hipStream_t stream) {
    constexpr size_t lds = 160 * 1024;
    static bool once = ((void)hipFuncSetAttribute((const void*)gemm_stream_kernel<EPI_RESID_F32, true, AUX, true, false, SPLIT>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL((gemm_stream_kernel<EPI_RESID_F32, true, AUX, true, false, SPLIT>), dim3(grid), dim3(512), lds, stream, a);
    CS_LAUNCH_CHECK();
    return 0;
}

template <int EPI>
int launch_stream_bf16(const GemmArgs& a, unsigned grid, hipStream_t stream) {
    return a.ln_mean ? launch_stream_t<EPI, true, false>(a, grid, stream) : launch_stream_t<EPI, false, false>(a, grid, stream);
}

template <int EPI, bool SLAB>
int launch_stream_f8(const GemmArgs& a, unsigned grid, hipStream_t stream) {
    constexpr size_t lds = 160 * 1024;
    static bool once = ((void)hipFuncSetAttribute((const void*)gemm_stream_kernel<EPI, false, false, SLAB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL((gemm_stream_kernel<EPI, false, false, SLAB, true>), dim3(grid), dim3(512), lds, stream, a);
    CS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// fp8 (e4m3) operands with per-row scales: C = row_scale[m] * col_scale[n] * (A8 . B8^T) + bias (+ extra), epi 0 = bf16 out, 2 = fp32
// residual (in place allowed).  K = bytes per operand row = contraction length padded to a multiple of 128 with zeros.
int cs_gemm_stream_launch_f8(GemmArgs a, int epi, int reserve, hipStream_t stream) {
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.N + 255) / 256;
    const long ntiles = (long)a.tiles_m * a.tiles_n;
    const long cap = cs_persistent_cap(reserve);
    const unsigned grid = (unsigned)(ntiles < cap ? ntiles : cap);
    if (epi == EPI_BF16) return launch_stream_f8<EPI_BF16, false>(a, grid, stream);
    return launch_stream_f8<EPI_RESID_F32, true>(a, grid, stream);
}

// Returns 1 when the problem is outside what the register epilogues cover (the caller falls back to gemm_persist_kernel), 0 on launch,
// < 0 on error.  `a` arrives with M/N/K, leading dimensions, operands and epilogue operands set (gemm_nt_impl); reserve = CUs to leave free.
#ifdef CS_ABLATION_SWITCHES
static int stream_launch_impl(GemmArgs a, int epi, int reserve, hipStream_t stream);
int cs_gemm_stream_launch(GemmArgs a, int epi, int reserve, hipStream_t stream) {
    static const char* path = getenv("CS_GEMM_TRACE");
    static unsigned long long* buf = nullptr;
    constexpr size_t WORDS = 256 * 8 * 16;
    if (path && !buf && hipMalloc((void**)&buf, WORDS * 8) == hipSuccess) {
        (void)hipMemset(buf, 0, WORDS * 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_epi_trace), &buf, sizeof(buf));
    }
    const int rc = stream_launch_impl(a, epi, reserve, stream);
    if (path && buf && rc == 0 && (epi == EPI_RESID_F32 || epi == EPI_RESID_LN_F32)) {      // the last launch's timeline survives in the file
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> host(WORDS);
        (void)hipMemcpy(host.data(), buf, WORDS * 8, hipMemcpyDeviceToHost);
        if (FILE* f = fopen(path, "wb")) { fwrite(host.data(), 8, WORDS, f); fclose(f); }
    }
    return rc;
}
static int stream_launch_impl(GemmArgs a, int epi, int reserve, hipStream_t stream) {
#else
int cs_gemm_stream_launch(GemmArgs a, int epi, int reserve, hipStream_t stream) {
#endif
    const bool swi = epi == EPI_SWIGLU_BF16, res = epi == EPI_RESID_F32 || epi == EPI_RESID_LN_F32;
    if (!(swi || res || epi == EPI_BF16 || epi == EPI_QGELU_BF16)) return 1;      // exact GELU (erf) keeps the slab epilogue: register pressure
    if (a.M < 1 || a.K % BK != 0) return 1;
    if (swi ? (a.group % 32 != 0) : (a.N % 32 != 0)) return 1;
    if ((long)a.ldc * 128 * 4 >= 0x70000000L || (long)a.ldxb * 128 * 2 >= 0x70000000L) return 1;      // 32-bit offsets inside a wave tile
    const bool ln = a.ln_mean != nullptr;
    if ((epi == EPI_RESID_LN_F32) != (res && ln)) return 1;
    bool aux = false;
    if (swi) aux = a.stats_part != nullptr;
    else if (res) {
        if ((a.stats_part != nullptr) != (a.xb_out != nullptr)) return 1;
        aux = a.stats_part != nullptr;
        if (aux && (a.ldxb % 8 != 0 || ((uintptr_t)a.xb_out % 16) != 0)) return 1;        // the bf16 copy leaves in 16-byte pieces
    } else if (a.stats_part || a.xb_out) return 1;
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = swi ? (a.group + 127) / 128 : (a.N + 255) / 256;
    long ntiles = (long)a.tiles_m * a.tiles_n;
    const long cap = cs_persistent_cap(reserve);
    unsigned grid = (unsigned)(ntiles < cap ? ntiles : cap);
    // 192-row tiles (plain epilogues of the training schedule): when the 256-row tiling leaves a large part of the last round of
    // workgroups empty -- cost = rounds of the persistent grid x rows per tile; taken when it saves more than 10 %
    if (!ln && !aux && !(a.dbg & 1) && a.M >= 192 && (swi || res || epi == EPI_BF16) && !getenv_flag("CS_NO_BM192")) {
        const long t192 = (long)((a.M + 191) / 192) * a.tiles_n;
        const long c256 = ((ntiles + cap - 1) / cap) * 256, c192 = ((t192 + cap - 1) / cap) * 192;
        if (c192 * 10 < c256 * 9) {
            a.tiles_m = (a.M + 191) / 192;
            ntiles = t192;
            grid = (unsigned)(ntiles < cap ? ntiles : cap);
            if (swi) return launch_stream_v<EPI_SWIGLU_BF16, false, false, false, 192>(a, grid, stream);
            if (res) return launch_stream_v<EPI_RESID_F32, false, false, true, 192>(a, grid, stream);
            return launch_stream_v<EPI_BF16, false, false, false, 192>(a, grid, stream);
        }
    }
    if (swi) {
        if (ln) return aux ? launch_stream_t<EPI_SWIGLU_BF16, true, true>(a, grid, stream) : launch_stream_t<EPI_SWIGLU_BF16, true, false>(a, grid, stream);
        return aux ? launch_stream_t<EPI_SWIGLU_BF16, false, true>(a, grid, stream) : launch_stream_t<EPI_SWIGLU_BF16, false, false>(a, grid, stream);
    }
    if (res) {
        if (ln) return aux ? launch_stream_t<EPI_RESID_F32, true, true>(a, grid, stream) : launch_stream_t<EPI_RESID_F32, true, false>(a, grid, stream);
        if (aux) return 1;
        return launch_stream_t<EPI_RESID_F32, false, false>(a, grid, stream);
    }
    if (epi == EPI_BF16) return launch_stream_bf16<EPI_BF16>(a, grid, stream);
    return launch_stream_bf16<EPI_QGELU_BF16>(a, grid, stream);
}

// C ABI: the folded-LayerNorm residual GEMM (cs_gemm_nt_ln epilogue 6) on the split stream -- see include/clipself_hip.h.
//   x_in != NULL: the stream is read as fp32 [M, ldc];  NULL: as the planes (hi, lo)
//   x_out != NULL: it is written as fp32 [M, ldc] (stats_part optional: no bf16 copy is made, hi is only read);  NULL: as (hi, lo), with
//   stats_part required (the consumer of a split stream is a folded GEMM, which needs the row statistics).
extern "C" int cs_gemm_nt_ln_split(const void* A, const void* B, const float* bias, const float* ln_mean, const float* ln_rstd,
                                   const float* ln_colsum, const float* x_in, float* x_out, void* hi, void* lo, int ldxb, float* stats_part,
                                   int M, int N, int K, int lda, int ldb, int ldc, int flags, hipStream_t stream) {
    CS_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % BK == 0 && N % 32 == 0, "cs_gemm_nt_ln_split: M=%d N=%d (%% 32) K=%d (%% 64)", M, N, K);
    CS_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "cs_gemm_nt_ln_split: operands must be 16-byte aligned rows");
    CS_CHECK_ARG(ln_mean && ln_rstd && ln_colsum && bias && ((uintptr_t)ln_colsum % 16) == 0 && ((uintptr_t)bias % 16) == 0,
                 "cs_gemm_nt_ln_split: mean, rstd, column sums and bias are required (16-byte aligned vectors)");
    CS_CHECK_ARG(x_in == nullptr || x_out == nullptr, "cs_gemm_nt_ln_split: fp32 in and fp32 out is cs_gemm_nt_ln");
    CS_CHECK_ARG(hi && lo && ((uintptr_t)hi % 16) == 0 && ((uintptr_t)lo % 16) == 0 && ldxb % 8 == 0 && ldxb >= N, "cs_gemm_nt_ln_split: hi / lo planes: 16-byte aligned, ldxb %% 8 == 0");
    CS_CHECK_ARG((x_in == nullptr || (((uintptr_t)x_in % 16) == 0 && ldc % 4 == 0)) && (x_out == nullptr || (((uintptr_t)x_out % 16) == 0 && ldc % 4 == 0)),
                 "cs_gemm_nt_ln_split: the fp32 stream must be 16-byte aligned with ldc %% 4 == 0");
    CS_CHECK_ARG(x_out != nullptr || stats_part != nullptr, "cs_gemm_nt_ln_split: a stream that leaves split leaves with its row statistics");
    CS_CHECK_ARG(x_out == nullptr || stats_part == nullptr, "cs_gemm_nt_ln_split: fp32 out takes no stats_part (hi / lo are only read then; the statistics epilogue would overwrite hi)");
    CS_CHECK_ARG(!(x_in != nullptr && stats_part == nullptr), "cs_gemm_nt_ln_split: fp32 in -> split out needs stats_part");
    CS_CHECK_ARG((long)ldc * 128 * 4 < 0x70000000L && (long)ldxb * 128 * 2 < 0x70000000L, "cs_gemm_nt_ln_split: row stride too large");
    GemmArgs a;
    a.A = (const __bf16*)A; a.B = (const __bf16*)B; a.C = x_out; a.bias = bias; a.extra = x_in;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.group = 0;
    a.ktiles_per_split = K / BK; a.split_stride = 0; a.gm = 8; a.rm = 0; a.nsplit = 1;
    a.ln_mean = ln_mean; a.ln_rstd = ln_rstd; a.ln_colsum = ln_colsum; a.stats_part = stats_part; a.xb_out = (__bf16*)hi; a.ldxb = ldxb;
    a.lo = (unsigned short*)lo;
    a.split = (x_in == nullptr ? 1 : 0) | (x_out == nullptr ? 2 : 0);
    a.reserve = (flags >> 20) & 255;
    a.dbg = (flags >> 12) & 15;
    a.tiles_m = (M + 255) / 256;
    a.tiles_n = (N + 255) / 256;
    const long ntiles = (long)a.tiles_m * a.tiles_n;
    const long cap = cs_persistent_cap(a.reserve);
    const unsigned grid = (unsigned)(ntiles < cap ? ntiles : cap);
    if (a.split == 3) return launch_stream_split<true, 3>(a, grid, stream);
    if (a.split == 2) return launch_stream_split<true, 2>(a, grid, stream);
    return launch_stream_split<false, 1>(a, grid, stream);
}

// C ABI: fp8 (OCP e4m3) operands quantised row-wise by cs_quant_rows_fp8 -- BASELINE configs[4] "fp8 MFMA weights".
//   C[m, n] = row_scale[m] * col_scale[n] * sum_k A8[m, k] * B8[n, k] + bias[n]  (+ extra[m, n] for epi 2)
//   epi 0: C bf16 [M, ldc];  epi 2: C fp32 [M, ldc] = extra + ..., in place allowed.  K8 = bytes per operand row (contraction length padded
//   to a multiple of 128 with zero bytes), lda / ldb = row strides in bytes.  flags bits 20-27 as cs_gemm_nt (compute units left free).
extern "C" int cs_gemm_nt_f8(const void* A8, const void* B8, void* C, const float* bias, const float* extra, const float* row_scale,
                             const float* col_scale, int M, int N, int K8, int lda, int ldb, int ldc, int epi, int flags, hipStream_t stream) {
    CS_CHECK_ARG(M > 0 && N > 0 && K8 > 0 && K8 % 128 == 0, "cs_gemm_nt_f8: K8=%d must be a positive multiple of 128 (M=%d N=%d)", K8, M, N);
    CS_CHECK_ARG(epi == EPI_BF16 || epi == EPI_RESID_F32, "cs_gemm_nt_f8: epilogue %d (0 = bf16 out, 2 = fp32 residual)", epi);
    CS_CHECK_ARG(N % 32 == 0 && ldc % 4 == 0 && lda >= K8 && ldb >= K8 && lda % 16 == 0 && ldb % 16 == 0,
                 "cs_gemm_nt_f8: N %% 32, ldc %% 4, lda/ldb >= K8 and multiples of 16 bytes");
    CS_CHECK_ARG(((uintptr_t)A8 % 16) == 0 && ((uintptr_t)B8 % 16) == 0 && ((uintptr_t)C % 16) == 0, "cs_gemm_nt_f8: operands must be 16-byte aligned");
    CS_CHECK_ARG(row_scale && col_scale, "cs_gemm_nt_f8: row and column scales are required");
    CS_CHECK_ARG(epi != EPI_RESID_F32 || (extra && ((uintptr_t)extra % 16) == 0), "cs_gemm_nt_f8: epilogue 2 needs 16-byte aligned extra");
    CS_CHECK_ARG((long)ldc * 128 * 4 < 0x70000000L, "cs_gemm_nt_f8: ldc too large");
    GemmArgs a;
    a.A = (const __bf16*)A8; a.B = (const __bf16*)B8; a.C = C; a.bias = bias; a.extra = extra;
    a.M = M; a.N = N; a.K = K8; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.group = 0;
    a.tiles_m = a.tiles_n = 0; a.ktiles_per_split = K8 / 128; a.split_stride = 0; a.gm = 8; a.rm = 0; a.nsplit = 1;
    a.ln_mean = nullptr; a.ln_rstd = row_scale; a.ln_colsum = col_scale; a.stats_part = nullptr; a.xb_out = nullptr; a.ldxb = 0;
    a.reserve = (flags >> 20) & 255;
    a.dbg = (flags >> 12) & 15;
    return cs_gemm_stream_launch_f8(a, epi, a.reserve, stream);
}
