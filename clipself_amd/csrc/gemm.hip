// bf16 MFMA GEMM for the CLIPSelf hot path (gfx950).
//
//   C[M,N] (+epilogue) = A[M,K] . B[N,K]^T        A, B bf16 row-major with the contraction
//                                                  dimension contiguous ("NT"); fp32 accumulate.
//
// Every matmul of the EVA02 block is brought to this one form:
//   forward  y = x W^T          A = x [M,K],      B = W [N,K]            (reference: F.linear,
//                                                  eva_vit_model.py:99-103,177-179,219,617)
//   dgrad    dx = dy W          A = dy [M,N'],    B = W^T [K',N'] (bf16 shadow kept transposed)
//   wgrad    dW = dy^T x        A = dy^T [N',Mp], B = x^T [K',Mp] (explicit transposes, split-K)
//
// Structure (MI355X: 256 CUs x 4 SIMDs, 64-lane waves, 160 KiB LDS/CU, 8 XCDs with private L2):
//   * workgroup tile BM x BN x 64: 256x256 (8 waves, 128x64 per wave, 1 WG/CU), 256x128 (8 waves, 64x64 per wave) and
//     128x128 (4 waves, 2 WG/CU); MFMA 32x32x16 bf16, fp32 accumulators in registers.  The large tiles exist to cut LDS
//     traffic per MFMA (operand re-use across the wave tile and across waves).
//   * operands go global -> LDS with the 16-byte `global_load_lds` DMA (no VGPR round trip), double buffered.
//     The LDS image is lane-linear by construction of that instruction, so the bank swizzle is applied to the per-lane
//     SOURCE address: two 128-byte tile rows share one 256-byte LDS row whose sixteen 16-byte slots are XOR-permuted
//     by (lds_row & 15) -> every ds_read_b128 lane group hits 16 distinct slots (conflict free; SQ_LDS_BANK_CONFLICT
//     < 2 % of wave cycles in profiles/).
//   * main-loop schedules here: "lockstep" -- one barrier per K tile, all 8 waves read fragments and issue MFMAs together (any tile
//     shape; small and split-K problems, the patch embedding, the wgrad partials) -- and, for the 256x256 tile, split operand rings
//     (three A stages, two B stages, DMA pieces interleaved with the MFMAs), one tile per workgroup (schedule 7) or as a persistent tile
//     loop (gemm_persist_kernel, schedule 9).  The tower shapes run the streaming kernel of gemm_stream.hip (schedule 11); this file's
//     persistent kernel is its fallback for the epilogues it does not cover (exact GELU, shapes with N % 32 != 0).
//   * epilogue through LDS: each wave parks a 32 x 64 fp32 slab of its accumulators in a private LDS region and reads
//     it back row-wise, so global stores/loads (bias, residual, pos-embed) are 16-byte per lane and cover whole
//     256-byte row segments.
//   * workgroup ids are remapped so that each XCD owns a contiguous range of tiles, rasterised in groups of 8 M panels
//     (B tiles stay in that XCD's 4 MiB L2 while 8 A panels stream past).
#include "gemm_common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

constexpr int EP_LD = 68;                       // fp32 row stride of the epilogue slab (64 + 4 pad)
constexpr int EP_BYTES = 32 * EP_LD * 4;        // 32 rows per wave slab

// Fused SwiGLU epilogue.  x1 and x2 of a hidden unit sit in the same lane and register of accumulator column-tiles j = 0 / 1, so
// silu(x1 + b1) * (x2 + b2) is formed in registers; two vertically adjacent results (accumulator registers e, e+1 = rows r, r+1)
// are packed into one bf16x2 word and only that -- 8 ds_write_b32 and 2 ds_read_b128 per 32x32 block instead of 32 and 8 --
// goes through the wave-private slab [16 row pairs][32 columns] to become 8-byte row-contiguous global stores.
template <int FM, int BN>
__device__ __forceinline__ void epilogue_swiglu(const GemmArgs& p, const f32x16 (&acc)[FM][2], char* slab_bytes, char* rowst_bytes, int lane,
                                                int row0, int tn, int wn) {
    const int l31 = lane & 31, hf = lane >> 5;
    uint32_t* slab = (uint32_t*)slab_bytes;
    const int hbase = tn * (BN / 2) + wn * 32;
    const int hl = hbase + l31;
    float b1 = 0.f, b2 = 0.f;
    if (p.bias && hl < p.group) { b1 = p.bias[hl]; b2 = p.bias[p.group + hl]; }
    // LayerNorm folded into this GEMM (norm2 of a frozen tower): x1 = rstd*(acc1 - mean*c1) + b1, likewise x2; the rows' (mean, rstd)
    // of this wave are staged behind its slab
    const bool ln = p.ln_mean != nullptr;
    float c1 = 0.f, c2 = 0.f;
    float2* rowst = (float2*)rowst_bytes;
    if (ln) {
        if (hl < p.group) { c1 = p.ln_colsum[hl]; c2 = p.ln_colsum[p.group + hl]; }
        for (int r = lane; r < FM * 32; r += 64) {
            const int row = min(row0 + r, p.M - 1);
            const float rs = p.ln_rstd[row];
            rowst[r] = make_float2(-rs * p.ln_mean[row], rs);          // value = rs * acc + (-rs * mean) * colsum + bias: two FMAs
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const int c4 = lane & 7, hcol = hbase + c4 * 4;
    const bool colok = hcol < p.group;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            float h[2];
            float4 st4 = make_float4(0.f, 1.f, 0.f, 1.f);
            if (ln) st4 = *(const float4*)(rowst + i * 32 + mfma32_row(e, lane));                    // rows r, r+1 (r even)
#pragma unroll
            for (int t = 0; t < 2; ++t) {        // hardware exp2/rcp (1 ulp each; the result is rounded to bf16)
                float u = acc[i][0][e + t], v = acc[i][1][e + t];
                if (ln) {
                    const float2 st = t ? make_float2(st4.z, st4.w) : make_float2(st4.x, st4.y);
                    u = fmaf(st.y, u, fmaf(st.x, c1, b1));
                    v = fmaf(st.y, v, fmaf(st.x, c2, b2));
                } else {
                    u += b1;
                    v += b2;
                }
                h[t] = u * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * u)) * v;
            }
            union { bf16x2 v; uint32_t u; } pk;
            pk.v[0] = f2bf(h[0]);
            pk.v[1] = f2bf(h[1]);
            const int rp = ((e & 3) >> 1) + 4 * (e >> 2) + 2 * hf;        // rows 2rp, 2rp+1 of the block
            slab[rp * 32 + l31] = pk.u;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int row_base = row0 + i * 32;
        float ps[4] = {0, 0, 0, 0}, pq[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int rp = (lane >> 3) + 8 * k;
            const uint4 w = *(const uint4*)(slab + rp * 32 + c4 * 4);
            const uint2 lo = make_uint2((w.x & 0xffffu) | (w.y << 16), (w.z & 0xffffu) | (w.w << 16));
            const uint2 hi = make_uint2((w.x >> 16) | (w.y & 0xffff0000u), (w.z >> 16) | (w.w & 0xffff0000u));
            const int ra = row_base + 2 * rp;
            if (colok && ra < p.M) *(uint2*)((__bf16*)p.C + (size_t)ra * p.ldc + hcol) = lo;
            if (colok && ra + 1 < p.M) *(uint2*)((__bf16*)p.C + (size_t)(ra + 1) * p.ldc + hcol) = hi;
            if (p.stats_part && colok) {
                const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float a = __uint_as_float(ws[t] << 16), b = __uint_as_float(ws[t] & 0xffff0000u);
                    ps[2 * k] += a; pq[2 * k] += a * a;
                    ps[2 * k + 1] += b; pq[2 * k + 1] += b * b;
                }
            }
        }
        if (p.stats_part) {
            // LayerNorm statistics of the rounded outputs: a row's 32 hidden units of this slice sit in 8 adjacent lanes (DPP
            // butterfly, every lane ends with the sum); lane (lane & 7) == j keeps row-partial j, so the 32 rows of the block
            // leave in ONE 256-byte store.  cs_ln_stats_finalize() pools the slices: no sub-LN pass over the hidden matrix.
            const int sel = lane & 7;
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = sum_lanes8(ps[j]), b = sum_lanes8(pq[j]);
                if (sel == j) { sv = a; qv = b; }
            }
            const int row = row_base + 2 * ((lane >> 3) + 8 * (sel >> 1)) + (sel & 1);
            if (sel < 4 && row < p.M) {
                const size_t slice = (size_t)tn * (BN / 64) + wn;
                *(float2*)(p.stats_part + (slice * p.M + row) * 2) = make_float2(sv, qv);
            }
        }
        if (i + 1 < FM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab reads done before the next block rewrites it
    }
}

// bf16 output epilogue (acc + bias): like epilogue_swiglu, vertically adjacent results are packed into bf16x2 words in registers, so
// the wave-private slab [16 row pairs][64 columns] sees 16 ds_write_b32 + 4 ds_read_b128 per 32x64 block instead of 32 + 8.
template <int ACT>
__device__ __forceinline__ float activate(float v) {
    if (ACT == 1) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));                       // nn.GELU (open_clip/transformer.py:195,211)
    if (ACT == 2) return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v));   // QuickGELU (:31-34)
    return v;
}

template <int FM, int BN, int ACT = 0>
__device__ __forceinline__ void epilogue_bf16(const GemmArgs& p, const f32x16 (&acc)[FM][2], char* slab_bytes, char* rowst_bytes, int lane,
                                              int row0, int n0, int wn) {
    const int l31 = lane & 31, hf = lane >> 5;
    uint32_t* slab = (uint32_t*)slab_bytes;
    float bj[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = n0 + wn * 64 + j * 32 + l31;
        if (p.bias && c < p.N) bj[j] = p.bias[c];
    }
    const bool ln = p.ln_mean != nullptr;      // LayerNorm folded into this GEMM (norm1 of a frozen tower), see epilogue_swiglu
    float cj[2] = {0.f, 0.f};
    float2* rowst = (float2*)rowst_bytes;
    if (ln) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = n0 + wn * 64 + j * 32 + l31;
            if (c < p.N) cj[j] = p.ln_colsum[c];
        }
        for (int r = lane; r < FM * 32; r += 64) {
            const int row = min(row0 + r, p.M - 1);
            const float rs = p.ln_rstd[row];
            rowst[r] = make_float2(-rs * p.ln_mean[row], rs);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const int c4 = lane & 15, col = n0 + wn * 64 + c4 * 4;
    const bool colok = col < p.N;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                float v0 = acc[i][j][e], v1 = acc[i][j][e + 1];
                if (ln) {
                    const float4 st = *(const float4*)(rowst + i * 32 + mfma32_row(e, lane));     // rows r, r+1 (r even): one 16-byte read
                    v0 = fmaf(st.y, v0, fmaf(st.x, cj[j], bj[j]));
                    v1 = fmaf(st.w, v1, fmaf(st.z, cj[j], bj[j]));
                } else {
                    v0 += bj[j];
                    v1 += bj[j];
                }
                union { bf16x2 v; uint32_t u; } pk;
                pk.v[0] = f2bf(activate<ACT>(v0));
                pk.v[1] = f2bf(activate<ACT>(v1));
                const int rp = ((e & 3) >> 1) + 4 * (e >> 2) + 2 * hf;
                slab[rp * 64 + j * 32 + l31] = pk.u;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int row_base = row0 + i * 32;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int rp = (lane >> 4) + 4 * k;
            const uint4 w = *(const uint4*)(slab + rp * 64 + c4 * 4);
            const uint2 lo = make_uint2((w.x & 0xffffu) | (w.y << 16), (w.z & 0xffffu) | (w.w << 16));
            const uint2 hi = make_uint2((w.x >> 16) | (w.y & 0xffff0000u), (w.z >> 16) | (w.w & 0xffff0000u));
            const int ra = row_base + 2 * rp;
            if (colok && ra < p.M) *(uint2*)((__bf16*)p.C + (size_t)ra * p.ldc + col) = lo;
            if (colok && ra + 1 < p.M) *(uint2*)((__bf16*)p.C + (size_t)(ra + 1) * p.ldc + col) = hi;
        }
        if (i + 1 < FM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// Epilogue through a wave-private LDS slab [32][EP_LD] fp32.  The slab is wave-private and a wave's DS operations execute in
// order, so no workgroup barrier is needed inside (a __syncthreads() would also wait for every outstanding global store).
template <int EPI, int FM, int FN, int BN, int LD = EP_LD>
__device__ __forceinline__ void epilogue_at(const GemmArgs& p, const f32x16 (&acc)[FM][FN], char* slab_bytes, int lane, int row0,
                                            int n0, int tn, int wn);

template <int EPI, int FM, int FN, int BN>
__device__ __forceinline__ void epilogue(const GemmArgs& p, const f32x16 (&acc)[FM][FN], char* smem, int wave, int lane, int row0,
                                         int n0, int tn, int wn) {
    epilogue_at<EPI, FM, FN, BN>(p, acc, smem + wave * EP_BYTES, lane, row0, n0, tn, wn);
}

// LD = fp32 row stride of the slab: 68 in the one-tile-per-workgroup kernels, 64 in the persistent kernel (8 KiB slabs, four per free ring slot)
template <int EPI, int FM, int FN, int BN, int LD>
__device__ __forceinline__ void epilogue_at(const GemmArgs& p, const f32x16 (&acc)[FM][FN], char* slab_bytes, int lane, int row0,
                                            int n0, int tn, int wn) {
    static_assert(FN == 2, "epilogue assumes 64-column wave tiles");
    if (EPI == EPI_SWIGLU_BF16) {
        epilogue_swiglu<FM, BN>(p, acc, slab_bytes, slab_bytes + 4096, lane, row0, tn, wn);
        return;
    }
    if (epi_is_bf16(EPI)) {
        epilogue_bf16<FM, BN, epi_act(EPI)>(p, acc, slab_bytes, slab_bytes + 4096, lane, row0, n0, wn);
        return;
    }
    const int l31 = lane & 31;
    float* slab = (float*)slab_bytes;
    const int rrow = lane >> 4, rcol = (lane & 15) * 4;  // read-out: 16 lanes per row, 4 consecutive columns per lane
    const int col = n0 + wn * 64 + rcol;                 // non-SwiGLU epilogues
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int row_base = row0 + i * 32;
        // All residual / pos-embed reads of the 32-row block are issued before the accumulators go through the slab (clamped
        // rows instead of predicates): a load->add->store chain per row would serialise 8 memory round trips per block
        // (vmcnt also counts the previous store), and here their latency hides behind the LDS transposition.
        float4 xin[8];
        float lmean[8], lrstd[8];
        if (EPI == EPI_RESID_LN_F32 && col < p.N) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = min(row_base + rrow + it * 4, p.M - 1);
                lmean[it] = p.ln_mean[row];
                lrstd[it] = p.ln_rstd[row];
            }
        }
        if ((EPI == EPI_RESID_F32 || EPI == EPI_RESID_LN_F32 || EPI == EPI_PATCH_F32) && col < p.N) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = min(row_base + rrow + it * 4, p.M - 1);
                if (EPI == EPI_RESID_F32 || EPI == EPI_RESID_LN_F32) {
                    // flags bit 12 (A/B switch): non-temporal accesses for the streams of the residual epilogue (read once / written once)
                    if (CS_ABL(p, 1)) {
                        const f32x4 t4 = __builtin_nontemporal_load((const f32x4*)(p.extra + (size_t)row * p.ldc + col));
                        xin[it] = make_float4(t4[0], t4[1], t4[2], t4[3]);
                    } else {
                        xin[it] = *(const float4*)(p.extra + (size_t)row * p.ldc + col);
                    }
                } else {
                    const int img = row / p.group, t = row - img * p.group;
                    xin[it] = *(const float4*)(p.extra + (size_t)(t + 1) * p.ldc + col);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) slab[mfma32_row(e, lane) * LD + j * 32 + l31] = acc[i][j][e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {
            float ps[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pq[8] = {0, 0, 0, 0, 0, 0, 0, 0};    // per-row partial statistics (residual epilogues)
            if (col < p.N) {
                float bv[4] = {0, 0, 0, 0};
                if (p.bias) { const float4 t = *(const float4*)(p.bias + col); bv[0] = t.x; bv[1] = t.y; bv[2] = t.z; bv[3] = t.w; }
                float cs[4] = {0, 0, 0, 0};
                if (EPI == EPI_RESID_LN_F32) { const float4 t = *(const float4*)(p.ln_colsum + col); cs[0] = t.x; cs[1] = t.y; cs[2] = t.z; cs[3] = t.w; }
                const bool full = row_base + 32 <= p.M;              // wave-uniform
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int rl = rrow + it * 4, row = row_base + rl;
                    if (!full && row >= p.M) continue;
                    const float4 s = *(const float4*)(slab + rl * LD + rcol);
                    float v[4] = {s.x + bv[0], s.y + bv[1], s.z + bv[2], s.w + bv[3]};
                    if (EPI == EPI_RESID_LN_F32) {
                        const float mu = lmean[it], rs = lrstd[it];
                        v[0] = rs * (s.x - mu * cs[0]) + bv[0]; v[1] = rs * (s.y - mu * cs[1]) + bv[1];
                        v[2] = rs * (s.z - mu * cs[2]) + bv[2]; v[3] = rs * (s.w - mu * cs[3]) + bv[3];
                    }
                    if (EPI == EPI_F32) {
                        *(float4*)((float*)p.C + (size_t)blockIdx.y * p.split_stride + (size_t)row * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                    } else if (EPI == EPI_RESID_F32 || EPI == EPI_RESID_LN_F32) {
                        const float4 x = xin[it];
                        const float o[4] = {x.x + v[0], x.y + v[1], x.z + v[2], x.w + v[3]};
                        if (CS_ABL(p, 1)) {
                            __builtin_nontemporal_store(f32x4{o[0], o[1], o[2], o[3]}, (f32x4*)((float*)p.C + (size_t)row * p.ldc + col));
                        } else {
                            *(float4*)((float*)p.C + (size_t)row * p.ldc + col) = make_float4(o[0], o[1], o[2], o[3]);
                        }
                        if (p.xb_out) {                       // bf16 copy of the new residual stream: A operand of the next LN-folded GEMM
                            U64 ob;
#pragma unroll
                            for (int t = 0; t < 4; ++t) ob.e[t] = f2bf(o[t]);
                            *(uint2*)(p.xb_out + (size_t)row * p.ldxb + col) = ob.u;
                        }
                        if (p.stats_part) {
#pragma unroll
                            for (int t = 0; t < 4; ++t) { ps[it] += o[t]; pq[it] += o[t] * o[t]; }
                        }
                    } else if (EPI == EPI_ATOMIC_F32) {
                        float* d = (float*)p.C + (size_t)row * p.ldc + col;
#pragma unroll
                        for (int t = 0; t < 4; ++t) unsafeAtomicAdd(d + t, v[t]);
                    } else if (EPI == EPI_PATCH_F32) {
                        const int img = row / p.group;
                        const float4 x = xin[it];
                        *(float4*)((float*)p.C + (size_t)(row + img + 1) * p.ldc + col) = make_float4(x.x + v[0], x.y + v[1], x.z + v[2], x.w + v[3]);
                    }
                }
            }
            if ((EPI == EPI_RESID_F32 || EPI == EPI_RESID_LN_F32) && p.stats_part) {
                // LayerNorm statistics of the fp32 outputs for the NEXT norm: a row's 64 columns of this slice sit in 16 adjacent lanes
                // (DPP butterfly); lane (lane & 15) == it keeps iteration it's row, so the 32 rows leave in one 256-byte store.
                const int sel = lane & 15;
                float sv = 0.f, qv = 0.f;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const float a = sum_lanes16(ps[it]), b = sum_lanes16(pq[it]);
                    if (sel == it) { sv = a; qv = b; }
                }
                const int row = row_base + rrow + 4 * sel;
                if (sel < 8 && row < p.M && n0 + wn * 64 < p.N) {
                    const size_t slice = (size_t)tn * (BN / 64) + wn;
                    *(float2*)(p.stats_part + (slice * p.M + row) * 2) = make_float2(sv, qv);
                }
            }
        }
        if (i + 1 < FM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab reads done before the next block rewrites it
    }
}

// ------------------------------------------------------------------------------------------------ lockstep schedule
template <int EPI, int BM, int BN, int WM, int WN, bool GLDS, int NS>
__global__ __launch_bounds__(WM * WN * 64) void gemm_nt_kernel(GemmArgs p) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;           // wave tile
    constexpr int FM = TM / 32, FN = TN / 32;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int hf = lane >> 5, l31 = lane & 31;
    int tm, tn;
    tile_of_block(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    int arow[A_INSTR], achk[A_INSTR], brow[B_INSTR], bchk[B_INSTR];
    source_rows<EPI, BM, BN, NW, A_INSTR, B_INSTR>(p, wave, lane, m0, n0, tn, arow, achk, brow, bchk);

    const int kt_begin = blockIdx.y * p.ktiles_per_split;
    const int kt_end = min(kt_begin + p.ktiles_per_split, p.K / BK);

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // NS == 32: split rings -- three A buffers (activations stream from HBM: two tiles of prefetch) and two B buffers
    // (weights sit in L2: one tile of prefetch), 3*A_BYTES + 2*B_BYTES = the whole 160 KiB for the 256x256 tile.
    constexpr bool AB = (NS == 32);
    char* const b_ring = smem + 3 * A_BYTES;
    if (kt_begin < kt_end) {
        stage_tile<A_INSTR, GLDS>(p.A, p.lda, kt_begin * BK, smem, wave * A_INSTR, lane, arow, achk);
        stage_tile<B_INSTR, GLDS>(p.B, p.ldb, kt_begin * BK, AB ? b_ring : smem + A_BYTES, wave * B_INSTR, lane, brow, bchk);
    }
    if (AB && kt_begin + 1 < kt_end)
        stage_tile<A_INSTR, GLDS>(p.A, p.lda, (kt_begin + 1) * BK, smem + A_BYTES, wave * A_INSTR, lane, arow, achk);
    // fragment addressing: row = base + l31 with base a multiple of 32 -> (row>>1)&15 == l31>>1, row&1 == l31&1
    const int a_base = ((wm * TM + l31) >> 1) << 8;
    const int b_base = ((wn * TN + l31) >> 1) << 8;
    const int par8 = (l31 & 1) << 3, sw = l31 >> 1;
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        if (AB) {
            // In issue order this wave's pending DMA is A(kt), B(kt), A(kt+1): everything but the newest A_INSTR ops must
            // have landed.  Raw s_barrier (a __syncthreads() would drain the queue).
            if (kt + 1 < kt_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_INSTR) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();     // tile kt landed everywhere; tile kt-1's buffers (A slot (kt+2)%3, B slot (kt+1)&1) are free
            const int i = kt - kt_begin;
            if (!(A_INSTR == 4 && B_INSTR == 4) || (p.dbg & 8)) {     // burst issue (dbg bit 3 = A/B switch); default: interleaved below
                if (kt + 1 < kt_end)
                    stage_tile<B_INSTR, GLDS>(p.B, p.ldb, (kt + 1) * BK, b_ring + ((i + 1) & 1) * B_BYTES, wave * B_INSTR, lane, brow, bchk);
                if (kt + 2 < kt_end)
                    stage_tile<A_INSTR, GLDS>(p.A, p.lda, (kt + 2) * BK, smem + ((i + 2) % 3) * A_BYTES, wave * A_INSTR, lane, arow, achk);
            }
        } else {
            __syncthreads();        // tile kt landed (the barrier drains the LDS-DMA queue); buffer cur^1 is free
            if (kt + 1 < kt_end && !CS_ABL(p, 1)) {
                char* nxt = smem + (cur ^ 1) * STAGE;
                stage_tile<A_INSTR, GLDS>(p.A, p.lda, (kt + 1) * BK, nxt, wave * A_INSTR, lane, arow, achk);
                stage_tile<B_INSTR, GLDS>(p.B, p.ldb, (kt + 1) * BK, nxt + A_BYTES, wave * B_INSTR, lane, brow, bchk);
            }
        }
        const char* la = (AB ? smem + cur * A_BYTES : smem + cur * STAGE) + a_base;
        const char* lb = (AB ? b_ring + ((kt - kt_begin) & 1) * B_BYTES : smem + cur * STAGE + A_BYTES) + b_base;
        if (CS_ABL(p, 2)) { cur = AB ? (cur == 2 ? 0 : cur + 1) : (cur ^ 1); continue; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int off = ((par8 | (ks * 2 + hf)) ^ sw) << 4;
            bf16x8 a[FM], b[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) a[i] = *(const bf16x8*)(la + i * (16 * 256) + off);
#pragma unroll
            for (int j = 0; j < FN; ++j) b[j] = *(const bf16x8*)(lb + j * (16 * 256) + off);
            if (AB && A_INSTR == 4 && B_INSTR == 4 && !(p.dbg & 8)) {
                // the 8 DMA pieces of this iteration are issued two per k-step, in the shadow of the MFMAs, instead of in one burst
                // behind the barrier (+1..3 %; same issue order: B(kt+1) first, then A(kt+2))
                const int i0 = kt - kt_begin;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int x = (ks & 1) * 2 + h2;
                    if (ks < 2) {
                        if (kt + 1 < kt_end) {
                            const int r1[1] = {brow[x]}, c1[1] = {bchk[x]};
                            stage_tile<1, GLDS>(p.B, p.ldb, (kt + 1) * BK, b_ring + ((i0 + 1) & 1) * B_BYTES, wave * B_INSTR + x, lane, r1, c1);
                        }
                    } else if (kt + 2 < kt_end) {
                        const int r1[1] = {arow[x]}, c1[1] = {achk[x]};
                        stage_tile<1, GLDS>(p.A, p.lda, (kt + 2) * BK, smem + ((i0 + 2) % 3) * A_BYTES, wave * A_INSTR + x, lane, r1, c1);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        cur = AB ? (cur == 2 ? 0 : cur + 1) : (cur ^ 1);
    }
    __syncthreads();                                    // every wave is done reading the operand buffers
    if (CS_ABL(p, 4)) return;
    epilogue<EPI, FM, FN, BN>(p, acc, smem, wave, lane, m0 + wm * TM, n0, tn, wn);
}

// ------------------------------------------------------------------------------------------------ persistent split-ring schedule
// The default 256x256 split-ring kernel as a persistent loop over tiles (one workgroup per CU, tile ids strided by the grid size, which
// keeps the XCD affinity of tile_of_id): the first operand tiles of the NEXT output tile (A0, B0, A1) are put in flight before the
// epilogue of the current one, whose packed slabs live in A slot 2 and row statistics in B slot 1, so the prologue latency and the
// workgroup relaunch disappear behind the store phase.  bf16, SwiGLU and fp32 residual epilogues (their slabs fit beside the prefetch).
// RM = raster / cache-policy mode: 0 grouped raster (tile_of_id) | 1 B-stationary raster (tile_bstat) | 2 = 1 + non-temporal A loads |
// 3 = grouped raster + non-temporal B loads
template <int EPI, int RM = 0>
__global__ __launch_bounds__(512) void gemm_persist_kernel(GemmArgs p) {
    constexpr int AUX_A = RM == 2 ? 2 : 0, AUX_B = RM == 3 ? 2 : 0;
    constexpr bool BSTAT = RM == 1 || RM == 2;
    static_assert(epi_is_bf16(EPI) || EPI == EPI_SWIGLU_BF16 || EPI == EPI_RESID_F32 || EPI == EPI_RESID_LN_F32, "epilogues whose slabs fit");
    constexpr bool PACKED = epi_is_bf16(EPI) || EPI == EPI_SWIGLU_BF16;
    constexpr int BM = 256, BN = 256, WN = 4, NW = 8, TM = 128, TN = 64, FM = 4, FN = 2;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, A_INSTR = 4, B_INSTR = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int hf = lane >> 5, l31 = lane & 31;
    char* const b_ring = smem + 3 * A_BYTES;
    // epilogue scratch lives in A slot 2 and B slot 1 -- the prefetch of the next tile only touches A0, A1 and B0.  Packed epilogues:
    // 4 KiB slabs in A2, row statistics in B1; fp32 residual epilogues: 8 KiB slabs (row stride 64), four in A2 and four in B1.
    char* const slab = PACKED ? smem + 2 * A_BYTES + wave * 4096 : (wave < 4 ? smem + 2 * A_BYTES + wave * 8192 : b_ring + B_BYTES + (wave - 4) * 8192);
    char* const rowst = b_ring + B_BYTES + wave * 1024;
    const int ntiles = p.tiles_m * p.tiles_n, ktiles = p.K / BK;
    const int a_base = ((wm * TM + l31) >> 1) << 8, b_base = ((wn * TN + l31) >> 1) << 8;
    const int par8 = (l31 & 1) << 3, sw = l31 >> 1;

    int tile = BSTAT ? (int)(blockIdx.x >> 3) : (int)blockIdx.x, tm, tn;      // BSTAT: XCD-local index, XCD = blockIdx.x & 7
    const int tstep = BSTAT ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    if constexpr (BSTAT) {
        if (!tile_bstat(p, blockIdx.x & 7, tile, tm, tn)) return;
    } else {
        tile_of_id(p, tile, ntiles, tm, tn);
    }
    int arow[A_INSTR], achk[A_INSTR], brow[B_INSTR], bchk[B_INSTR];
    source_rows<EPI, BM, BN, NW, A_INSTR, B_INSTR>(p, wave, lane, tm * BM, tn * BN, tn, arow, achk, brow, bchk);
    auto prologue = [&]() {
        stage_tile<A_INSTR, true, AUX_A>(p.A, p.lda, 0, smem, wave * A_INSTR, lane, arow, achk);
        stage_tile<B_INSTR, true, AUX_B>(p.B, p.ldb, 0, b_ring, wave * B_INSTR, lane, brow, bchk);
        if (ktiles > 1) stage_tile<A_INSTR, true, AUX_A>(p.A, p.lda, BK, smem + A_BYTES, wave * A_INSTR, lane, arow, achk);
    };
    prologue();
    bool first = true;
    for (;;) {
        const int m0 = tm * BM, n0 = tn * BN, tn_cur = tn;
        f32x16 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        int cur = 0;
        for (int kt = 0; kt < ktiles; ++kt) {
            // pending DMA in issue order: A(kt), B(kt), A(kt+1).  On the first K tile of a later output tile the queue also holds the
            // previous epilogue's stores (loads and stores do not retire in order with each other): drain it.
            if ((kt == 0 && !first) || kt + 1 >= ktiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_INSTR) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const char* la = smem + cur * A_BYTES + a_base;
            const char* lb = b_ring + (kt & 1) * B_BYTES + b_base;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = ((par8 | (ks * 2 + hf)) ^ sw) << 4;
                bf16x8 a[FM], b[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i) a[i] = *(const bf16x8*)(la + i * (16 * 256) + off);
#pragma unroll
                for (int j = 0; j < FN; ++j) b[j] = *(const bf16x8*)(lb + j * (16 * 256) + off);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {            // two DMA pieces per k-step: B(kt+1) first, then A(kt+2)
                    const int x = (ks & 1) * 2 + h2;
                    if (ks < 2) {
                        if (kt + 1 < ktiles) {
                            const int r1[1] = {brow[x]}, c1[1] = {bchk[x]};
                            stage_tile<1, true, AUX_B>(p.B, p.ldb, (kt + 1) * BK, b_ring + ((kt + 1) & 1) * B_BYTES, wave * B_INSTR + x, lane, r1, c1);
                        }
                    } else if (kt + 2 < ktiles) {
                        const int r1[1] = {arow[x]}, c1[1] = {achk[x]};
                        stage_tile<1, true, AUX_A>(p.A, p.lda, (kt + 2) * BK, smem + ((kt + 2) % 3) * A_BYTES, wave * A_INSTR + x, lane, r1, c1);
                    }
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            cur = cur == 2 ? 0 : cur + 1;
        }
        __syncthreads();                                  // every wave is done reading the operand rings; no DMA pending
        const int next = tile + tstep;
        bool more;
        if constexpr (BSTAT) more = tile_bstat(p, blockIdx.x & 7, next, tm, tn);
        else {
            more = next < ntiles;
            if (more) tile_of_id(p, next, ntiles, tm, tn);
        }
        if (more) {                                       // next tile's first operands fly during this tile's epilogue
            source_rows<EPI, BM, BN, NW, A_INSTR, B_INSTR>(p, wave, lane, tm * BM, tn * BN, tn, arow, achk, brow, bchk);
            prologue();
        }
        if constexpr (EPI == EPI_SWIGLU_BF16) epilogue_swiglu<FM, BN>(p, acc, slab, rowst, lane, m0 + wm * TM, tn_cur, wn);
        else if constexpr (epi_is_bf16(EPI)) epilogue_bf16<FM, BN, epi_act(EPI)>(p, acc, slab, rowst, lane, m0 + wm * TM, n0, wn);
        else epilogue_at<EPI, FM, FN, BN, 64>(p, acc, slab, lane, m0 + wm * TM, n0, tn_cur, wn);
        if (!more) break;
        tile = next;
        first = false;
    }
}

// ------------------------------------------------------------------------------------------------ launch
template <int EPI, int BM, int BN, int WM, int WN, int NS>
int launch_cfg(GemmArgs a, int splits, int use_glds, hipStream_t stream) {
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (EPI == EPI_SWIGLU_BF16) ? (a.group + BN / 2 - 1) / (BN / 2) : (a.N + BN - 1) / BN;
    constexpr int NW = WM * WN;
    constexpr size_t stage = NS == 32 ? (size_t)(3 * BM + 2 * BN) * BK * 2 : (size_t)(BM + BN) * BK * 2 * NS;
    constexpr size_t lds = stage > (size_t)NW * EP_BYTES ? stage : (size_t)NW * EP_BYTES;
    dim3 grid(a.tiles_m * a.tiles_n, splits), block(NW * 64);
    if (use_glds) {
        static bool once = ((void)hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI, BM, BN, WM, WN, true, NS>,
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
        (void)once;
        hipLaunchKernelGGL((gemm_nt_kernel<EPI, BM, BN, WM, WN, true, NS>), grid, block, lds, stream, a);
    } else {
        static bool once = ((void)hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI, BM, BN, WM, WN, false, NS>,
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
        (void)once;
        hipLaunchKernelGGL((gemm_nt_kernel<EPI, BM, BN, WM, WN, false, NS>), grid, block, lds, stream, a);
    }
    CS_LAUNCH_CHECK();
    return 0;
}

template <int EPI, int RM>
int launch_persist_rm(const GemmArgs& a, unsigned grid, hipStream_t stream) {
    constexpr size_t lds = 160 * 1024;
    static bool once = ((void)hipFuncSetAttribute((const void*)gemm_persist_kernel<EPI, RM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL((gemm_persist_kernel<EPI, RM>), dim3(grid), dim3(512), lds, stream, a);
    CS_LAUNCH_CHECK();
    return 0;
}

template <int EPI>
int launch_persist(GemmArgs a, hipStream_t stream) {
    constexpr int BM = 256, BN = 256;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (EPI == EPI_SWIGLU_BF16) ? (a.group + BN / 2 - 1) / (BN / 2) : (a.N + BN - 1) / BN;
    const long ntiles = (long)a.tiles_m * a.tiles_n;
    const long cap = cs_persistent_cap(a.reserve);       // flags bits 20-27: compute units left free
    const unsigned grid = (unsigned)(ntiles < cap ? ntiles : cap);
    if constexpr (EPI == EPI_BF16 || EPI == EPI_SWIGLU_BF16) {           // the wide-N GEMMs of the towers (q|k|v, W1|W2)
        // the B-stationary raster gives every XCD gridDim / 8 workgroups: any full persistent grid that is a multiple of 8 will do -- also
        // the reduced grid of a launch that leaves compute units to RCCL or to the other tower (a.reserve)
        const bool parts_ok = (long)grid == cap && grid % 8 == 0 && a.tiles_n >= a.nsplit && a.tiles_m >= 8 / a.nsplit;
        if (a.rm == 1 && parts_ok) return launch_persist_rm<EPI, 1>(a, grid, stream);
        if (a.rm == 2 && parts_ok) return launch_persist_rm<EPI, 2>(a, grid, stream);
        if (a.rm == 3) return launch_persist_rm<EPI, 3>(a, grid, stream);
    }
    return launch_persist_rm<EPI, 0>(a, grid, stream);
}

template <int EPI>
int launch(GemmArgs a, int splits, int use_glds, int force_cfg, hipStream_t stream) {
    const long ncols = (EPI == EPI_SWIGLU_BF16) ? 2L * a.group : a.N;
    const int ktiles = a.K / BK;
    auto tiles = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((ncols + bn - 1) / bn); };
    // split-K (atomic epilogue only, splits <= 0 = automatic): per tile shape pick the slice count minimising a two-term
    // cost -- MFMA time (rounds of resident workgroups x K tiles per slice + a fixed pro/epilogue) + the fp32 atomic
    // traffic, which grows with the number of slices (every slice adds the whole tile to C).
    const double out_bytes = 4.0 * (double)a.M * (double)ncols;
    auto best_split = [&](int bm, int bn, int per_cu, double us_per_ktile, double& cost) {
        const long t = tiles(bm, bn);
        const long slots = (long)cs_num_cus() * per_cu;
        int best = 1;
        cost = 1e30;
        const int lo = splits > 0 ? splits : 1, hi = splits > 0 ? splits : (ktiles < 64 ? ktiles : 64);
        for (int s = lo; s <= hi; ++s) {
            const long rounds = (t * s + slots - 1) / slots;
            const int kper = (ktiles + s - 1) / s;
            const double c = rounds * (kper * us_per_ktile + 5.0) + (EPI == EPI_ATOMIC_F32 ? s * out_bytes / 0.33e6 : 0.0);   // measured: fp32 atomics sustain ~0.33 TB/s
            if (c < cost) { cost = c; best = s; }
        }
        return best;
    };
    double c1, c2, c3;
    // measured microseconds per K tile of one resident workgroup set: 256x256 1.4 (the streaming kernel: 1.3-1.45 on M = 18 464 / 12 608 rows,
    // round 3), 256x128 0.95, 2 x 128x128 1.3.  Round 1's figures (2.2 / 1.4 / 1.9) sent the N = 1024, K = 3072 / 5504 dgrads of the L/14
    // student to the 256x128 kernel: 297 / 145 us against 232 / 132 us on the streaming kernel.
    const int sp[4] = {0, best_split(128, 128, 2, 1.3, c1), best_split(256, 128, 1, 0.95, c2), best_split(256, 256, 1, 1.4, c3)};
    int cfg = force_cfg;
    if (cfg == 0) {
        if (a.M < 256 || ncols < 256) c3 = 1e30;
        if (a.M < 256 || ncols < 128) c2 = 1e30;
        cfg = (c3 <= c2 && c3 <= c1) ? 3 : (c2 <= c1 ? 2 : 1);
        // split rings: +1..3 % over the lockstep 2-stage ring on the tower shapes; as a persistent tile loop (bf16 / GELU / SwiGLU /
        // residual epilogues, case 9 falls back to 7 for the others): another -2.4 % (q|k|v) / -4.9 % (W1|W2) per launch
        // ... and with register-level epilogues on a continuous operand ring (gemm_stream.hip) for the bf16 / QuickGELU / SwiGLU outputs:
        // q|k|v 1600 -> 1480 us, W1|W2 2840 -> 2430 us per 2048-crop launch; the fp32 residual epilogues (HBM-bound, whole-line slab
        // form) proj 870 -> 800 us, w3 1510 -> 1445 us (profiles/r02_a_stream_gemm.md)
        if (cfg == 3 && use_glds) cfg = 11;
    }
    if (cfg != 1 && cfg != 2 && cfg != 3 && cfg != 7 && cfg != 9 && cfg != 11) {
        cs_set_error("cs_gemm_nt: schedule %d does not exist (flags bits 4-7: 0 heuristic, 1, 2, 3, 7, 9, 11)", cfg);
        return -1;
    }
    const int ns = sp[cfg >= 7 ? 3 : cfg];                                                  // 7, 9 and 11 are 256x256 schedules
    a.ktiles_per_split = (ktiles + ns - 1) / ns;
    if (cfg == 11) {                                                                        // streaming persistent kernel, register epilogues
        if (use_glds && ns == 1) {
            const int rc = cs_gemm_stream_launch(a, EPI, a.reserve, stream);
            if (rc <= 0) return rc;
        }
        cfg = 9;                                                                            // outside its coverage: slab-epilogue persistent kernel
    }
    switch (cfg) {
        case 9:                                                                                 // persistent split rings (packed / residual epilogues)
            if constexpr (epi_is_bf16(EPI) || EPI == EPI_SWIGLU_BF16 || EPI == EPI_RESID_F32 || EPI == EPI_RESID_LN_F32) {
                if (use_glds && ns == 1) return launch_persist<EPI>(a, stream);
            }
            return launch_cfg<EPI, 256, 256, 2, 4, 32>(a, ns, use_glds, stream);
        case 7: return launch_cfg<EPI, 256, 256, 2, 4, 32>(a, ns, use_glds, stream);       // 256x256, A ring 3 / B ring 2 (160 KiB)
        case 3: return launch_cfg<EPI, 256, 256, 2, 4, 2>(a, ns, use_glds, stream);
        case 2: return launch_cfg<EPI, 256, 128, 4, 2, 2>(a, ns, use_glds, stream);
        default: return launch_cfg<EPI, 128, 128, 2, 2, 2>(a, ns, use_glds, stream);
    }
}

}  // namespace

// C ABI ------------------------------------------------------------------------------------------
// epi: 0 bf16 out (+bias) | 1 f32 out (+bias) | 2 f32 out = extra(residual) + acc + bias |
//      3 fused SwiGLU (B = [W1;W2] stacked [2*group, K], bias [2*group], out bf16 [M, group]) |
//      4 f32 atomic accumulate (split-K, C pre-zeroed or accumulating) |
//      5 patch-embed: out row = row + row/group + 1, += extra[(row%group+1)*ldc + col] |
//      6 residual with a folded LayerNorm (cs_gemm_nt_ln) | 7 bf16 out = GELU(acc + bias) (exact, erf) | 8 bf16 out = QuickGELU(acc + bias)
// flags bit0: 0 = global_load_lds staging, 1 = register staging (debug/fallback A-B switch)
//       bits 4-7: force schedule (1 = 128x128, 2 = 256x128, 3 = 256x256 lockstep, 7 = 256x256 split rings A3/B2 (one tile per workgroup),
//                 9 = persistent split rings (bf16 / GELU / SwiGLU / residual epilogues through LDS slabs),
//                 11 = streaming persistent kernel with register epilogues (gemm_stream.hip: bf16 / QuickGELU / SwiGLU / residual;
//                      others fall back to 9); 0 = heuristic: the cheapest of 1 / 2 / 3 by a cost model, 3 running as 11).
//                 Rounds 1-2 also carried a 3-stage 256x128 ring, an L2 warm-up variant, two ping-pong schedules and a two-workgroups-
//                 per-CU K-32 ring; all measured slower on this path (profiles/r01_t_gemm_schedule_experiments.md) and were removed.
//       bits 8-11: raster group height override (0 = 8)
//       bit 12: streaming kernel: slab form of the bf16 / SwiGLU epilogues (exact A/B switch); bit 15: split-ring schedule issues its DMA in
//               one burst behind the barrier instead of interleaved with the MFMAs (exact A/B switch).  Builds with -DCS_ABLATION_SWITCHES
//               additionally read bits 12-14 as timing ablations with wrong results (gemm_common.h: CS_ABL); the shipped library does not.
//       bits 20-27: compute units the persistent kernels leave free (grid = compute units - n; multi-GPU runs keep room for RCCL's kernels)
//       bits 16-17 (persistent kernel, epilogues 0 and 3): 1 = B-stationary raster (each XCD keeps its share of B in L2; bits 8-11 = N parts,
//                 0 = automatic), 2 = the same with non-temporal A loads, 3 = grouped raster with non-temporal B loads
static int gemm_nt_impl(const void* A, const void* B, void* C, const float* bias, const float* extra, const float* ln_mean,
                        const float* ln_rstd, const float* ln_colsum, float* stats_part, void* xb_out, int ldxb, int M, int N, int K, int lda,
                        int ldb, int ldc, int epi, int splits, int group, int flags, hipStream_t stream) {
    CS_CHECK_ARG(M > 0 && N > 0 && K > 0, "cs_gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
    CS_CHECK_ARG(K % BK == 0, "cs_gemm_nt: K=%d must be a multiple of %d", K, BK);
    CS_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0, "cs_gemm_nt: lda/ldb must be multiples of 8 (16-byte rows)");
    CS_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "cs_gemm_nt: operands must be 16-byte aligned");
    CS_CHECK_ARG(splits == 1 || epi == EPI_ATOMIC_F32, "cs_gemm_nt: split-K (splits != 1; <= 0 = automatic) needs the atomic epilogue");
    CS_CHECK_ARG((epi == EPI_SWIGLU_BF16 ? group : N) % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C % 16) == 0,
                 "cs_gemm_nt: N, ldc must be multiples of 4 and C 16-byte aligned (vector epilogue)");
    CS_CHECK_ARG(bias == nullptr || ((uintptr_t)bias % 16) == 0, "cs_gemm_nt: bias must be 16-byte aligned");
    GemmArgs a;
    a.A = (const __bf16*)A; a.B = (const __bf16*)B; a.C = C; a.bias = bias; a.extra = extra;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.group = group;
    a.split_stride = 0;
    a.ln_mean = ln_mean; a.ln_rstd = ln_rstd; a.ln_colsum = ln_colsum; a.stats_part = stats_part;
    a.xb_out = (__bf16*)xb_out; a.ldxb = ldxb;
    a.tiles_m = a.tiles_n = 0;
    a.gm = ((flags >> 8) & 15) ? ((flags >> 8) & 15) : 8;
    a.rm = (flags >> 16) & 3;
    a.nsplit = 1;
    if (a.rm == 1 || a.rm == 2) {                          // B-stationary raster: the raster field carries the number of N parts
        const int f = (flags >> 8) & 15;
        CS_CHECK_ARG(f == 0 || f == 1 || f == 2 || f == 4 || f == 8, "cs_gemm_nt: B-stationary raster needs 1, 2, 4 or 8 N parts (got %d)", f);
        // automatic: the fewest parts whose share of B (bf16 [N/parts, K]) leaves room beside the streaming operands in a 4 MiB L2
        int parts = f;
        if (parts == 0) for (parts = 1; parts < 8 && (double)N * K * 2 / parts > 3.3e6; parts *= 2) {}
        a.nsplit = parts;
        a.gm = 8;
    }
    if (epi == EPI_SWIGLU_BF16) CS_CHECK_ARG(group > 0 && N == 2 * group, "cs_gemm_nt: swiglu epilogue needs N == 2*group");
    if (epi == EPI_RESID_LN_F32) CS_CHECK_ARG(ln_mean && ln_rstd && ln_colsum && ((uintptr_t)ln_colsum % 16) == 0, "cs_gemm_nt_ln: epilogue 6 needs mean, rstd and a 16-byte aligned column-sum vector");
    CS_CHECK_ARG(stats_part == nullptr || epi == EPI_SWIGLU_BF16 || epi == EPI_RESID_F32 || epi == EPI_RESID_LN_F32,
                 "cs_gemm_nt_ln: statistics output exists for the SwiGLU and residual epilogues");
    CS_CHECK_ARG(xb_out == nullptr || ((epi == EPI_RESID_F32 || epi == EPI_RESID_LN_F32) && ldxb % 4 == 0 && ((uintptr_t)xb_out % 8) == 0),
                 "cs_gemm_nt_ln: the bf16 copy exists for the residual epilogues (8-byte aligned, ldxb %% 4 == 0)");
    CS_CHECK_ARG((ln_mean == nullptr) == (ln_rstd == nullptr) && (ln_mean == nullptr || ln_colsum != nullptr),
                 "cs_gemm_nt_ln: mean, rstd and the column-sum vector come together");
    CS_CHECK_ARG(ln_mean == nullptr || epi_is_bf16(epi) || epi == EPI_SWIGLU_BF16 || epi == EPI_RESID_LN_F32,
                 "cs_gemm_nt_ln: a folded LayerNorm exists for epilogues 0, 3, 6, 7 and 8");
    if (epi == EPI_PATCH_F32 || epi == EPI_RESID_F32 || epi == EPI_RESID_LN_F32) CS_CHECK_ARG(extra != nullptr && ((uintptr_t)extra % 16) == 0, "cs_gemm_nt: epilogue %d needs 16-byte aligned extra", epi);
    if (epi == EPI_PATCH_F32) CS_CHECK_ARG(group > 0, "cs_gemm_nt: patch epilogue needs group");
    a.ktiles_per_split = K / BK;
    const int glds = (flags & 1) ? 0 : 1;
    const int force = (flags >> 4) & 15;
    a.dbg = (flags >> 12) & 15;
    a.reserve = (flags >> 20) & 255;
    switch (epi) {
        case EPI_BF16: return launch<EPI_BF16>(a, splits, glds, force, stream);
        case EPI_F32: return launch<EPI_F32>(a, splits, glds, force, stream);
        case EPI_RESID_F32: return launch<EPI_RESID_F32>(a, splits, glds, force, stream);
        case EPI_SWIGLU_BF16: return launch<EPI_SWIGLU_BF16>(a, splits, glds, force, stream);
        case EPI_ATOMIC_F32: return launch<EPI_ATOMIC_F32>(a, splits, glds, force, stream);
        case EPI_PATCH_F32: return launch<EPI_PATCH_F32>(a, splits, glds, force, stream);
        case EPI_RESID_LN_F32: return launch<EPI_RESID_LN_F32>(a, splits, glds, force, stream);
        case EPI_GELU_BF16: return launch<EPI_GELU_BF16>(a, splits, glds, force, stream);
        case EPI_QGELU_BF16: return launch<EPI_QGELU_BF16>(a, splits, glds, force, stream);
    }
    cs_set_error("cs_gemm_nt: unknown epilogue %d", epi);
    return -1;
}

// EPI_F32 with the K range cut into `splits` slices, slice y writing its [M,N] partial to C + y*M*N (cs_gemm_wgrad)
static int gemm_nt_split_f32(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int cfg, int splits, hipStream_t stream) {
    CS_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % BK == 0 && N % 4 == 0, "cs_gemm_wgrad: bad problem M=%d N=%d K=%d", M, N, K);
    CS_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "cs_gemm_wgrad: operands must be 16-byte aligned rows");
    GemmArgs a;
    a.A = (const __bf16*)A; a.B = (const __bf16*)B; a.C = C; a.bias = nullptr; a.extra = nullptr;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = N; a.group = 0;
    a.split_stride = (long)M * N;
    a.ln_mean = a.ln_rstd = a.ln_colsum = nullptr; a.stats_part = nullptr; a.xb_out = nullptr; a.ldxb = 0;
    a.tiles_m = a.tiles_n = 0; a.gm = 8; a.dbg = 0; a.rm = 0; a.nsplit = 1; a.reserve = 0;
    a.ktiles_per_split = K / BK;
    return launch<EPI_F32>(a, splits, 1, cfg, stream);
}

extern "C" int cs_gemm_nt(const void* A, const void* B, void* C, const float* bias, const float* extra, int M, int N, int K,
                          int lda, int ldb, int ldc, int epi, int splits, int group, int flags, hipStream_t stream) {
    CS_CHECK_ARG(epi != EPI_RESID_LN_F32, "cs_gemm_nt: epilogue 6 (folded LayerNorm) is reached through cs_gemm_nt_ln");
    return gemm_nt_impl(A, B, C, bias, extra, nullptr, nullptr, nullptr, nullptr, nullptr, 0, M, N, K, lda, ldb, ldc, epi, splits, group, flags, stream);
}

// cs_gemm_nt plus the folded-LayerNorm operands (frozen towers; see GemmArgs):
//   epi 6: C = extra + ln_rstd[m] * (A.B^T - ln_mean[m] * ln_colsum[n]) + bias[n]     (A un-normalised, B = gamma (.) W)
//   epi 3 with stats_part != null: also writes, per 32-hidden-unit slice s and row m, (sum, sum of squares) of the rounded
//          outputs to stats_part[(s*M + m)*2 ..]; slices = 4*ceil(group/128); combine with cs_ln_stats_finalize.
//   epi 0 / 3 with ln_mean != null: the LayerNorm in front of the GEMM is folded the same way (norm1 -> q|k|v, norm2 -> W1|W2):
//          A = bf16 copy of the un-normalised rows, value = ln_rstd[m] * (acc - ln_mean[m] * ln_colsum[n]) + bias[n] before bf16 / SiLU.
//   epi 2 / 6 with xb_out != null: also stores that bf16 copy of the fp32 output (row stride ldxb); with stats_part != null also the
//          per-64-column-slice (sum, sum of squares) of the fp32 outputs, stats_part[(s*M + m)*2 ..], slices = ceil(N/64).
extern "C" int cs_gemm_nt_ln(const void* A, const void* B, void* C, const float* bias, const float* extra, const float* ln_mean,
                             const float* ln_rstd, const float* ln_colsum, float* stats_part, void* xb_out, int ldxb, int M, int N, int K,
                             int lda, int ldb, int ldc, int epi, int splits, int group, int flags, hipStream_t stream) {
    return gemm_nt_impl(A, B, C, bias, extra, ln_mean, ln_rstd, ln_colsum, stats_part, xb_out, ldxb, M, N, K, lda, ldb, ldc, epi, splits, group,
                        flags, stream);
}


// ------------------------------------------------------------------------------------------------ wgrad: split-K through partials
// dW[M,N] += A[M,K] . B[N,K]^T with a long contraction (K = tokens) and a small output: the K range is cut into `splits` slices
// whose partial products go to a workspace with plain stores, and one streaming pass adds them into dW.  (fp32 atomics into dW
// cost ~3 us per MB and slice on this part -- 0.33 TB/s -- which made the atomic split-K form 2-4x slower; profiles/r01_j.)
namespace {

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, long stride, float* __restrict__ dst,
                                                            int M, int N, int ldc) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;          // one float4 of the [M, N] partials
    const int n4 = N >> 2;
    if (i >= (long)M * n4) return;
    const int row = (int)(i / n4), c = (int)(i - (long)row * n4) * 4;
    float4 acc = *(const float4*)(ws + (size_t)row * N + c);
    for (int s = 1; s < splits; ++s) {
        const float4 t = *(const float4*)(ws + (size_t)s * stride + (size_t)row * N + c);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    float4* d = (float4*)(dst + (size_t)row * ldc + c);
    const float4 o = *d;
    *d = make_float4(o.x + acc.x, o.y + acc.y, o.z + acc.z, o.w + acc.w);
}

// ------------------------------------------------------------------------------------------------ wgrad without transposes ("TN")
// dW[N, K] = sum_m dY[m, n] * X[m, k] with BOTH operands token-major as the forward / backward kernels left them (rows = tokens,
// contraction = the row index): the MFMA operand of lane (n, half) is 8 consecutive tokens of ONE column -- a strided access that
// ds_read_b64_tr_b16 performs in hardware (within a 16-lane group lane i hands in the address of four contiguous bf16 = row i>>2,
// segment i&3 of a [4 tokens][16 columns] block and receives column i of the four rows).  Tiles: 256 (n) x 256 (k) outputs, K tiles of
// 64 tokens; LDS image of an operand tile = [64 tokens][256 columns] bf16 (512-byte rows) staged by the 16-byte LDS-DMA, the 32-byte
// column groups of row r XOR-permuted by (r & 7) on the SOURCE side so that the four rows of a transposing read hit four different bank
// groups.  Same split rings (A x3, B x2), counted waits and split-K-through-partials epilogue as the NT kernel; replaces the explicit
// bf16 transposes of dY and X (2 x 4 per block and step).  Any token count, N and K multiples of 8 (ragged edges: see the kernel).
typedef __attribute__((ext_vector_type(4))) short s16x4;

// 32-byte column group g of token row t sits at group g ^ tn_swz(t): the two 16-lane groups of a half-wave read the same four tokens at
// adjacent column groups, so the four tokens must land on every second group (g ^ (t & 7) leaves a 2-way conflict: measured 2-5 % slower)
__device__ __forceinline__ int tn_swz(int tok) { return (tok & 3) << 1; }
// The transposing reads go out as inline asm (round 4).  Through the builtin, hipcc (ROCm 7.2) cannot tell the read's LDS slot from the
// slots the pending LDS-DMA pieces write and drains the DMA queue -- s_waitcnt vmcnt(0) -- in front of the first read that follows a DMA
// issue: the K loop then waited for the burst it had just issued (B(kt+1), A(kt+2)) before touching K tile kt, i.e. no operand prefetch
// at all.  An asm read is invisible to that bookkeeping; its completion is counted by hand (tr_wait: lgkmcnt(0) naming the fragment
// registers, so that no MFMA that consumes them is scheduled above the wait -- cdna_hip_programming.md 5.7 form (ii)).
typedef short s16x8 __attribute__((ext_vector_type(8)));
struct TrFrag { s16x4 lo, hi; };
__device__ __forceinline__ void tr_read(TrFrag& f, const char* tile, int tok, int colbyte) {
    // rows tok .. tok+3 and tok+4 .. tok+7 of the lane's column; colbyte = logical byte offset of the lane's 8-byte segment in the row
    const int a0 = tok * 512 + (colbyte ^ (tn_swz(tok) << 5)), a1 = (tok + 4) * 512 + (colbyte ^ (tn_swz(tok + 4) << 5));
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)tile;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(base + (unsigned)a0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(base + (unsigned)a1));
}
__device__ __forceinline__ bf16x8 tr_join(const TrFrag& f) {
    const s16x8 v = __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// RAGGED: the shape has partial tiles (tokens % 64, N % 256 or K % 256); the exact-tile variant carries none of that code (it costs the
// longest shape 7 % when compiled in: one more uniform branch in the DMA issue and 19 more registers).
template <bool RAGGED>
__global__ __launch_bounds__(512) void gemm_tn_kernel(GemmArgs p) {
    constexpr int BM = 256, BN = 256, WN = 4, TM = 128, TN = 64, FM = 4, FN = 2;
    constexpr int T_BYTES = 64 * 512;                     // one operand tile: 64 tokens x 256 columns
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int li = lane & 15, grp = lane >> 4, hf = grp >> 1, g1 = grp & 1;
    int tm, tn;
    tile_of_block(p, tm, tn);
    const int n0 = tm * BM, k0 = tn * BN;
    char* const b_ring = smem + 3 * T_BYTES;
    // DMA: instruction x of this wave fills LDS rows 2*(wave*4+x), +1; lane l lands at row + (l >> 5), physical 16-byte chunk l & 31
    const int drow = lane >> 5, pc = lane & 31;
    // Ragged edges: columns past the operand's width re-read its last 8 columns (they only reach output rows / columns the epilogue masks);
    // token rows past the end re-read the last token, and those rows of the dY image are zeroed in LDS before they are read (last K tile only).
    // Round 4 (as in gemm_stream.hip): the two waves of a SIMD take different DMA roles -- waves 0-3 issue the 8 pieces of the X (B) tile
    // right behind the barrier, waves 4-7 the 8 pieces of the dY (A) tile behind the last fragment reads of the K tile -- so that a wave
    // stalled in its DMA burst sits beside a partner that is issuing MFMAs.  One operand per wave: ooff / ocol describe ITS 8 pieces.
    const bool role_a = wave >= 4;
    const int piece0 = (wave & 3) * 8;
    unsigned ocol[8], ooff[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        const int row = 2 * (piece0 + x) + drow;
        const int lc = ((((pc >> 1) ^ tn_swz(row)) << 1) | (pc & 1));   // logical 16-byte chunk this LDS position holds
        const unsigned ca = (unsigned)(RAGGED ? min(n0 + lc * 8, p.M - 8) : n0 + lc * 8) * 2u;
        const unsigned cb = (unsigned)(RAGGED ? min(k0 + lc * 8, p.N - 8) : k0 + lc * 8) * 2u;
        ocol[x] = role_a ? ca : cb;
        ooff[x] = (unsigned)row * (unsigned)(role_a ? p.lda : p.ldb) * 2u + ocol[x];      // full tiles: one 32-bit offset per piece
    }
    const int ktiles_all = (p.K + 63) >> 6;
    const int kt_begin = blockIdx.y * p.ktiles_per_split, kt_end = min(kt_begin + p.ktiles_per_split, ktiles_all);
    // the wave's 8 pieces of ITS operand tile kt (dY for waves 4-7, X for waves 0-3) into the slot at dst
    auto issue = [&](int kt, char* dst) {
        const int ld = role_a ? p.lda : p.ldb;
        const char* src = (const char*)(role_a ? p.A : p.B) + (size_t)kt * 64 * ld * 2;
        const int rmax = p.K - 1 - kt * 64;                     // last valid token row of this tile (wave-uniform)
        if (!RAGGED || rmax >= 63) {
#pragma unroll
            for (int x = 0; x < 8; ++x)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ooff[x]),
                                                 (__attribute__((address_space(3))) void*)(dst + (piece0 + x) * 1024), 16, 0, 0);
        } else {                                                // ragged last tile
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int row = min(2 * (piece0 + x) + drow, rmax);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)row * ld * 2 + ocol[x]),
                                                 (__attribute__((address_space(3))) void*)(dst + (piece0 + x) * 1024), 16, 0, 0);
            }
        }
    };
    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // in flight at the top of K tile kt -- A loaders: dY(kt), dY(kt+1); B loaders: X(kt)
    if (role_a) {
        if (kt_begin < kt_end) issue(kt_begin, smem);
        if (kt_begin + 1 < kt_end) issue(kt_begin + 1, smem + T_BYTES);
    } else if (kt_begin < kt_end) {
        issue(kt_begin, b_ring);
    }
    // lane's 8-byte segment inside a 32-column block: column 16*g1 + 4*(li&3); tokens of its 16-lane group: 8*hf + (li>>2)
    const int segbyte = (16 * g1 + 4 * (li & 3)) * 2, trow = 8 * hf + (li >> 2);
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        if (role_a && kt + 1 < kt_end) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // dY(kt) landed, dY(kt+1) may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int i0 = kt - kt_begin;
        // B loaders: one burst behind the barrier.  (Spreading pieces over the k-steps makes hipcc drain the DMA queue -- vmcnt(0) -- in
        // front of every transposing read that follows a DMA issue: measured 115.9 -> 136.9 us on the W1|W2 shape; the A loaders' burst
        // therefore sits behind the LAST fragment reads of the K tile, below.)
        if (!role_a && kt + 1 < kt_end) issue(kt + 1, b_ring + ((i0 + 1) & 1) * T_BYTES);
        const char* ta = smem + cur * T_BYTES;
        const char* tb = b_ring + (i0 & 1) * T_BYTES;
        // ragged last token tile: the rows of the missing tokens hold copies of the last token -- zero them in the dY image (wave-uniform
        // branch taken once per workgroup at most; the full tiles pay nothing)
        const int tvalid = p.K - kt * 64;
        if (RAGGED && tvalid < 64) {
            for (int idx = tid; idx < (64 - tvalid) * 32; idx += 512)
                *(uint4*)(smem + cur * T_BYTES + (tvalid + (idx >> 5)) * 512 + (idx & 31) * 16) = make_uint4(0, 0, 0, 0);
            __syncthreads();
        }
        // fragments of k-step ts+1 are requested before the MFMAs of k-step ts (two register sets): the transposing reads otherwise sit
        // directly in front of the MFMAs that consume them
        TrFrag a[2][FM], b[2][FN];
        auto load_frags = [&](int set, int ts) {
#pragma unroll
            for (int i = 0; i < FM; ++i) tr_read(a[set][i], ta, 16 * ts + trow, (wm * TM + i * 32) * 2 + segbyte);
#pragma unroll
            for (int j = 0; j < FN; ++j) tr_read(b[set][j], tb, 16 * ts + trow, (wn * TN + j * 32) * 2 + segbyte);
        };
        // every asm read issued so far has landed; the fragment registers of `set` are operands of the wait, so their consumers stay below it
        // (the operand list names a[set][0..3] and b[set][0..1]; the asm reads carry no memory clobber -- their order against the LDS-DMA
        // writes of the ring is kept by the K tile's barrier and the hand-counted vmcnt in front of it, not by the compiler)
        static_assert(FM == 4 && FN == 2, "tr_wait names exactly FM = 4 and FN = 2 fragment pairs: extend its operand list with the tile");
        auto tr_wait = [&](int set) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(a[set][0].lo), "+v"(a[set][0].hi), "+v"(a[set][1].lo), "+v"(a[set][1].hi), "+v"(a[set][2].lo), "+v"(a[set][2].hi),
                           "+v"(a[set][3].lo), "+v"(a[set][3].hi), "+v"(b[set][0].lo), "+v"(b[set][0].hi), "+v"(b[set][1].lo), "+v"(b[set][1].hi));
        };
        load_frags(0, 0);
        tr_wait(0);
#pragma unroll
        for (int ts = 0; ts < 4; ++ts) {
            if (ts < 3) load_frags((ts + 1) & 1, ts + 1);
            // A loaders: dY(kt+2) goes to the slot of dY(kt-1), free since this K tile's barrier; no LDS read of this K tile follows
            if (ts == 3 && role_a && kt + 2 < kt_end) issue(kt + 2, smem + ((i0 + 2) % 3) * T_BYTES);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(a[ts & 1][i]), tr_join(b[ts & 1][j]), acc[i][j], 0, 0, 0);
            if (ts < 3) tr_wait((ts + 1) & 1);
        }
        cur = cur == 2 ? 0 : cur + 1;
    }
    __syncthreads();
    epilogue<EPI_F32, FM, FN, BN>(p, acc, smem, wave, lane, n0 + wm * TM, k0, tn, wn);
}

// tile schedule (1 = 128x128 two per CU, 2 = 256x128, 7 = 256x256) and slice count minimising MFMA rounds + partial traffic
void choose_wgrad(int M, int N, int K, int& cfg, int& splits) {
    const int ktiles = K / BK;
    const double out_mb = 4.0 * M * N / 1e6;
    const struct { int cfg, bm, bn, per_cu; double us; } cand[3] = {{1, 128, 128, 2, 1.9}, {2, 256, 128, 1, 1.4}, {7, 256, 256, 1, 2.2}};
    double best = 1e30;
    cfg = 7; splits = 1;
    for (const auto& c : cand) {
        if (M < c.bm || N < c.bn) continue;
        const long tiles = (long)((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
        for (int s = 1; s <= 64 && s <= ktiles; ++s) {
            const long rounds = (tiles * s + (long)cs_num_cus() * c.per_cu - 1) / ((long)cs_num_cus() * c.per_cu);
            const int kper = (ktiles + s - 1) / s;
            const double t = rounds * (kper * c.us + 6.0) + (s + 2) * out_mb / 5.0 + 6.0;
            if (t < best) { best = t; cfg = c.cfg; splits = s; }
        }
    }
}

}  // namespace

namespace {
int choose_tn_splits(int N, int K, int tokens) {
    const long tiles = (long)((N + 255) / 256) * ((K + 255) / 256);
    const int ktiles = (tokens + 63) / 64;
    double best = 1e30;
    int splits = 1;
    for (int sp = 1; sp <= 64 && sp <= ktiles; ++sp) {
        const long rounds = (tiles * sp + 255) / 256;
        const double t = rounds * (((ktiles + sp - 1) / sp) * 2.2 + 6.0) + (sp + 2) * (4.0 * N * K / 1e6) / 5.0;
        if (t < best) { best = t; splits = sp; }
    }
    return splits;
}
}  // namespace

// Token-major wgrad: dW[N,K] (f32, row stride ldc) += dY[tokens,N]^T . X[tokens,K], bf16 operands as the step's kernels left them (no
// transposed copies).  Any token count; N and K multiples of 8.  Returns 1 (nothing launched) outside that coverage, so that the caller
// can take the transposing path (cs_transpose_bf16 + cs_gemm_wgrad).  Workspace: cs_gemm_wgrad_tn_workspace bytes.
extern "C" size_t cs_gemm_wgrad_tn_workspace(int N, int K, int tokens) {
    if (tokens <= 0 || N < 8 || K < 8 || N % 8 != 0 || K % 8 != 0) return 0;
    return (size_t)choose_tn_splits(N, K, tokens) * N * K * sizeof(float);
}

extern "C" int cs_gemm_wgrad_tn(const void* dY, const void* X, float* dW, void* workspace, int N, int K, int tokens, int ldy, int ldx, int ldc,
                                hipStream_t stream) {
    if (tokens <= 0 || N < 8 || K < 8 || N % 8 != 0 || K % 8 != 0) return 1;
    CS_CHECK_ARG(workspace != nullptr && ((uintptr_t)workspace % 16) == 0 && dW != nullptr && ((uintptr_t)dW % 16) == 0 && ldc % 4 == 0,
                 "cs_gemm_wgrad_tn: workspace / dW must be 16-byte aligned buffers");
    CS_CHECK_ARG(ldy % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)dY % 16) == 0 && ((uintptr_t)X % 16) == 0 && ldy >= N && ldx >= K,
                 "cs_gemm_wgrad_tn: operand rows must be 16-byte aligned");
    CS_CHECK_ARG((long)64 * ldy * 2 < 0x7fffffffL && (long)64 * ldx * 2 < 0x7fffffffL, "cs_gemm_wgrad_tn: row stride too large");
    int splits = choose_tn_splits(N, K, tokens);
    GemmArgs a;
    a.A = (const __bf16*)dY; a.B = (const __bf16*)X; a.C = workspace; a.bias = nullptr; a.extra = nullptr;
    a.M = N; a.N = K; a.K = tokens; a.lda = ldy; a.ldb = ldx; a.ldc = K; a.group = 0;
    a.split_stride = (long)N * K;
    a.ln_mean = a.ln_rstd = a.ln_colsum = nullptr; a.stats_part = nullptr; a.xb_out = nullptr; a.ldxb = 0;
    a.tiles_m = (N + 255) / 256; a.tiles_n = (K + 255) / 256; a.gm = 8; a.dbg = 0; a.rm = 0; a.nsplit = 1; a.reserve = 0;
    const int ktiles = (tokens + 63) / 64;
    a.ktiles_per_split = (ktiles + splits - 1) / splits;
    constexpr size_t lds = 160 * 1024;
    static bool once = ((void)hipFuncSetAttribute((const void*)gemm_tn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    if (tokens % 64 == 0 && N % 256 == 0 && K % 256 == 0) hipLaunchKernelGGL(gemm_tn_kernel<false>, dim3(a.tiles_m * a.tiles_n, splits), dim3(512), lds, stream, a);
    else hipLaunchKernelGGL(gemm_tn_kernel<true>, dim3(a.tiles_m * a.tiles_n, splits), dim3(512), lds, stream, a);
    CS_LAUNCH_CHECK();
    const long n = (long)N * (K >> 2);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)workspace, splits,
                       (long)N * K, dW, N, K, ldc);
    CS_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t cs_gemm_wgrad_workspace(int M, int N, int K) {
    int cfg, splits;
    choose_wgrad(M, N, K, cfg, splits);
    return (size_t)splits * M * N * sizeof(float);
}

// dW[M,N] (f32, row stride ldc) += A[M,K] . B[N,K]^T, bf16 operands; workspace >= cs_gemm_wgrad_workspace(M,N,K) bytes, 16-byte aligned.
extern "C" int cs_gemm_wgrad(const void* A, const void* B, float* dW, void* workspace, int M, int N, int K, int lda, int ldb, int ldc,
                             hipStream_t stream) {
    CS_CHECK_ARG(workspace != nullptr && ((uintptr_t)workspace % 16) == 0 && dW != nullptr && ((uintptr_t)dW % 16) == 0 && ldc % 4 == 0,
                 "cs_gemm_wgrad: workspace / dW must be 16-byte aligned buffers");
    int cfg, splits;
    choose_wgrad(M, N, K, cfg, splits);
    const int rc = gemm_nt_split_f32(A, B, (float*)workspace, M, N, K, lda, ldb, cfg, splits, stream);
    if (rc) return rc;
    const long n = (long)M * (N >> 2);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)workspace, splits,
                       (long)M * N, dW, M, N, ldc);
    CS_LAUNCH_CHECK();
    return 0;
}
