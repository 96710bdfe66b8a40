// LayerNorm (fwd/bwd) and row L2-normalise (fwd/bwd) -- HBM-bound row kernels, one 64-lane wave per
// row, 16-byte (fp32) / 8-byte (bf16) vector accesses, wave-shuffle reductions, fp32 statistics.
//
// Reference semantics: LayerNorm(eps=1e-6, biased variance)
//   src/open_clip/eva_clip/transformer.py:52-58, eva_clip/model.py:123 (norm1, norm2, inner_attn_ln,
//   ffn_ln, final norm: eva_vit_model.py:306-307,218,102,616) and F.normalize(dim=-1, eps=1e-12)
//   (eva_vit_model.py:620).
#include "cs_common.h"

namespace {

constexpr int MAXC = 3072;  // vec4 groups per lane NG in {4,8,12} -> C <= 1024 / 2048 / 3072
constexpr int ROWS_PER_WG = 4;

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    typedef float4 raw;
    static __device__ __forceinline__ raw load_raw(const float* p) { return *(const float4*)p; }
    static __device__ __forceinline__ void cvt(const raw& t, float (&v)[4]) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) { cvt(load_raw(p), v); }
};
template <> struct Vec4<__bf16> {
    typedef uint2 raw;
    static __device__ __forceinline__ raw load_raw(const __bf16* p) { return *(const uint2*)p; }
    static __device__ __forceinline__ void cvt(const raw& r, float (&v)[4]) {
        U64 t; t.u = r;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = bf2f(t.e[i]);
    }
    static __device__ __forceinline__ void load(const __bf16* p, float (&v)[4]) { cvt(load_raw(p), v); }
};
__device__ __forceinline__ void store_bf16x4(__bf16* p, const float (&v)[4]) {
    U64 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t.e[i] = f2bf(v[i]);
    *(uint2*)p = t.u;
}
__device__ __forceinline__ void store_x4(__bf16* p, const float (&v)[4]) { store_bf16x4(p, v); }
__device__ __forceinline__ void store_x4(float* p, const float (&v)[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }

// ---------------------------------------------------------------------------------------------
// Q8 (bf16 output only): the row also leaves as e4m3 bytes with one fp32 scale -- the A operand of the fp8 GEMM that consumes this
// LayerNorm (cs_gemm_nt_f8), quantised from the ROUNDED bf16 values exactly as cs_quant_rows_fp8 would from y (same amax / 448 scale,
// same v_cvt_pk_fp8_f32), columns C .. Kp-1 zero: the quantiser's pass over y (read 2 B, write 1 B per element, one launch) disappears.
template <typename TX, int MAXG, typename TY = __bf16, bool Q8 = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TX* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TY* __restrict__ y, long ldy,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     int M, int C, float eps, unsigned char* __restrict__ q8 = nullptr, long ldq = 0,
                                                     float* __restrict__ q_scale = nullptr, int Kp = 0) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_WG + (threadIdx.x >> 6);
    if (row >= M) return;
    const TX* xr = x + (size_t)row * ldx;
    const int ng = (C + 255) >> 8;
    const int clast = (C - 1) & ~3;                  // last vector start inside the row
    // All loads of the row are issued before the first use, from branch-free code: with the load and its use inside one `c < C` block,
    // hipcc waits for every vector group separately (vmcnt(0) per group: C/256 serial round trips per row).  Lanes past the row end
    // re-read its last vector and are zeroed by a select.
    float v[MAXG][4];
    typename Vec4<TX>::raw xr_raw[MAXG];
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        const int c = (g * 64 + lane) * 4;
        xr_raw[g] = Vec4<TX>::load_raw(xr + (g < ng ? min(c, clast) : 0));          // groups past the row end re-read its first vector (L1 hit)
    }
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        const int c = (g * 64 + lane) * 4;
        Vec4<TX>::cvt(xr_raw[g], v[g]);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[g][i] = (g < ng && c < C) ? v[g][i] : 0.f;
        s += (v[g][0] + v[g][1]) + (v[g][2] + v[g][3]);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        const int c = (g * 64 + lane) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = v[g][i] - mean; q += (g < ng && c + i < C) ? d * d : 0.f; }   // C may end inside a vector (padded rows)
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    if (y != nullptr) {                              // y == null: statistics only
        TY* yr = y + (size_t)row * ldy;
        float4 ga[MAXG], be[MAXG];
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int c = g < ng ? min((g * 64 + lane) * 4, clast) : 0;
            ga[g] = *(const float4*)(gamma + c);
            be[g] = *(const float4*)(beta + c);
        }
        float amax = 0.f;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int c = (g * 64 + lane) * 4;
            if (g < ng && c < C) {
                float o[4] = {(v[g][0] - mean) * rstd * ga[g].x + be[g].x, (v[g][1] - mean) * rstd * ga[g].y + be[g].y,
                              (v[g][2] - mean) * rstd * ga[g].z + be[g].z, (v[g][3] - mean) * rstd * ga[g].w + be[g].w};
                store_x4(yr + c, o);
                if (Q8) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[g][i] = (c + i < C) ? bf2f(f2bf(o[i])) : 0.f;       // what y holds: the quantiser's input
                        amax = fmaxf(amax, fabsf(v[g][i]));
                    }
                }
            } else if (Q8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[g][i] = 0.f;
            }
        }
        if (Q8) {
            amax = wave_max(amax);
            const float inv = amax > 0.f ? 448.f / amax : 0.f;
            if (lane == 0) q_scale[row] = amax > 0.f ? amax / 448.f : 1.f;
#pragma unroll
            for (int g = 0; g < MAXG; ++g) {
                const int c = (g * 64 + lane) * 4;
                if (c < Kp) {                                                  // Kp % 128 == 0: whole 4-byte groups; columns >= C are zero bytes
                    unsigned w = 0;
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[g][0] * inv, v[g][1] * inv, w, false);
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[g][2] * inv, v[g][3] * inv, w, true);
                    *(unsigned*)(q8 + (size_t)row * ldq + c) = w;
                }
            }
        }
    }
    if (lane == 0 && mean_out) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// Combine per-slice (sum, sum of squares) partials -- written by the SwiGLU GEMM epilogue (32 columns per slice) or the attention
// forward (64 per head) -- into LayerNorm mean / rstd per row.  Slice p covers columns [p*npp, min((p+1)*npp, C)); the pooled
// variance = sum of the slices' own centred second moments + the spread of the slice means about the row mean (Chan et al.).
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ part, int P, int npp, int C, int M, float eps,
                                                                float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    // one row per thread, slices in order, eight 8-byte loads in flight per thread (a wave reads 512 contiguous bytes per slice): the kernel
    // is a pure stream of P * M * 8 bytes (up to 206 MB for the SwiGLU hidden's 64 slices), so what matters is bytes in flight
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= M) return;
    const float2* src = (const float2*)part + row;
    float s = 0.f, q = 0.f, b = 0.f;                  // sum, within-slice centred squares, sum of s_p^2 / n_p
    int p = 0;
    for (; p + 8 <= P; p += 8) {
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(p + j) * M];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = min(npp, C - (p + j) * npp);
            if (n > 0) {
                const float t = v[j].x * v[j].x / (float)n;
                s += v[j].x;
                q += fmaxf(v[j].y - t, 0.f);
                b += t;
            }
        }
    }
    for (; p < P; ++p) {
        const int n = min(npp, C - p * npp);
        if (n <= 0) break;
        const float2 sq = src[(size_t)p * M];
        const float t = sq.x * sq.x / (float)n;
        s += sq.x;
        q += fmaxf(sq.y - t, 0.f);
        b += t;
    }
    const float mean = s / (float)C;
    // pooled variance = within-slice part + between-slice part; only the latter (slice means against the row mean) can
    // cancel, and it is a small share of the total unless the row mean dwarfs the row's spread
    const float m2 = q + fmaxf(b - s * mean, 0.f);
    mean_out[row] = mean;
    rstd_out[row] = rsqrtf(m2 / (float)C + eps);
}

// Two adjacent rows per thread, 16-byte loads (round 4): the same per-row arithmetic in the same order (bit-identical mean / rstd), half the
// load instructions -- the 64-slice launch (SwiGLU hidden, 206 MB) is a pure stream.  Needs an even M (the 16-byte alignment of a slice's rows).
__global__ __launch_bounds__(256) void ln_stats_finalize2_kernel(const float* __restrict__ part, int P, int npp, int C, int M, float eps,
                                                                 float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int row = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (row >= M) return;
    const float4* src = (const float4*)(part + (size_t)row * 2);
    const size_t stride = (size_t)M / 2;               // float4 units between slices
    float s[2] = {0.f, 0.f}, q[2] = {0.f, 0.f}, b[2] = {0.f, 0.f};
    auto add = [&](const float4& v, int n) {
        const float t0 = v.x * v.x / (float)n, t1 = v.z * v.z / (float)n;
        s[0] += v.x; q[0] += fmaxf(v.y - t0, 0.f); b[0] += t0;
        s[1] += v.z; q[1] += fmaxf(v.w - t1, 0.f); b[1] += t1;
    };
    // sixteen slices in flight per thread, the ragged last batch included (slices past P re-read slice P - 1 and are not added): round 4's
    // form walked the last P % 8 slices one dependent load at a time -- for the 12-slice launches (three of a block's four) that was four
    // of its five memory round trips.  Same additions in the same order: bit-identical statistics.
    for (int p = 0; p < P; p += 16) {
        float4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = src[(size_t)min(p + j, P - 1) * stride];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int n = min(npp, C - (p + j) * npp);
            if (p + j < P && n > 0) add(v[j], n);
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float mean = s[r] / (float)C;
        const float m2 = q[r] + fmaxf(b[r] - s[r] * mean, 0.f);
        mean_out[row + r] = mean;
        rstd_out[row + r] = rsqrtf(m2 / (float)C + eps);
    }
}

// dx modes
enum { DX_BF16 = 0, DX_F32_ASSIGN = 1, DX_F32_ACCUM = 2 };

// Backward.  One wave per row, grid-stride over rows; per-lane partial dgamma/dbeta are reduced across the
// workgroup's waves in LDS and written as one partial row per workgroup; ln_param_reduce sums them.
// COPY (fp32 dx modes): the updated residual-gradient row also leaves as a bf16 copy (the operand of the next dgrad / wgrad GEMMs) and
// its column sums -- the bias gradient of the linear layer in front of this LayerNorm's residual branch -- ride along as a third
// partial row: the `cast_f32_bf16` + `colsum_bf16` passes over the stream that used to follow every such LayerNorm backward are gone,
// and the sums are combined in a fixed order (no atomics).
template <typename TX, int DXMODE, int MAXG, bool COPY>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const __bf16* __restrict__ dy, long lddy, const TX* __restrict__ x, long ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, void* __restrict__ dx, long lddx,
                                                     __bf16* __restrict__ dxb, long lddxb, float* __restrict__ part, int M, int C,
                                                     unsigned char* __restrict__ q8 = nullptr, long ldq = 0, float* __restrict__ q_scale = nullptr,
                                                     int Kp = 0) {
    // q8 (COPY only, nullable): the bf16 copy's row also leaves as e4m3 bytes + one fp32 scale, = cs_quant_rows_fp8 of the copy (the fp8
    // dgrad's A operand), from the registers that hold the rounded row
    static_assert(!COPY || DXMODE != DX_BF16, "the bf16 copy exists for the fp32 stream modes");
    constexpr int NR = COPY ? 3 : 2;               // partial rows: dgamma, dbeta[, column sums of the bf16 copy]
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = (float*)smem_raw;                 // [3 waves][NR][Cp]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ng = (C + 255) >> 8;
    const int Cp = (C + 3) & ~3;                   // partial rows are laid out [NR][Cp] so that float4 accesses stay aligned
    const int clast = (C - 1) & ~3;
    float dg[MAXG][4], db[MAXG][4], ga[MAXG][4], dc[COPY ? MAXG : 1][4];
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        const int c = (g * 64 + lane) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) { dg[g][i] = 0.f; db[g][i] = 0.f; ga[g][i] = 0.f; if (COPY) dc[g][i] = 0.f; }
        if (g < ng && c < C) { const float4 t = *(const float4*)(gamma + c); ga[g][0] = t.x; ga[g][1] = t.y; ga[g][2] = t.z; ga[g][3] = t.w; }
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        float xh[MAXG][4], gy[MAXG][4];
        float s1 = 0.f, s2 = 0.f;
        // loads of the whole row first, branch-free (see ln_fwd_kernel): lanes past the row end re-read the last vector and are masked
        float xv[MAXG][4], dv[MAXG][4];
        typename Vec4<TX>::raw x_raw[MAXG];
        typename Vec4<__bf16>::raw d_raw[MAXG];
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int c = g < ng ? min((g * 64 + lane) * 4, clast) : 0;
            x_raw[g] = Vec4<TX>::load_raw(x + (size_t)row * ldx + c);
            d_raw[g] = Vec4<__bf16>::load_raw(dy + (size_t)row * lddy + c);
        }
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int c = (g * 64 + lane) * 4;
            const bool ok = g < ng && c < C;
            Vec4<TX>::cvt(x_raw[g], xv[g]);
            Vec4<__bf16>::cvt(d_raw[g], dv[g]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = ok ? dv[g][i] : 0.f;
                xh[g][i] = ok ? (xv[g][i] - mean) * rstd : 0.f;
                gy[g][i] = d * ga[g][i];
                s1 += gy[g][i];
                s2 += gy[g][i] * xh[g][i];
                dg[g][i] += d * xh[g][i];
                db[g][i] += d;
            }
        }
        const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
        float rq[COPY ? MAXG : 1][4];              // the rounded copy, kept for the quantiser
        float amax = 0.f;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int c = (g * 64 + lane) * 4;
            if (COPY) { rq[g][0] = rq[g][1] = rq[g][2] = rq[g][3] = 0.f; }
            if (g < ng && c < C) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (c + i < C) ? rstd * (gy[g][i] - m1 - xh[g][i] * m2) : 0.f;
                if (DXMODE == DX_BF16) {
                    store_bf16x4((__bf16*)dx + (size_t)row * lddx + c, o);
                } else {
                    float* p = (float*)dx + (size_t)row * lddx + c;
                    if (DXMODE == DX_F32_ACCUM) {
                        const float4 t = *(const float4*)p;
                        o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
                    }
                    *(float4*)p = make_float4(o[0], o[1], o[2], o[3]);
                    if (COPY) {
                        U64 t;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            t.e[i] = f2bf(o[i]);
                            rq[g][i] = (c + i < C) ? bf2f(t.e[i]) : 0.f;
                            dc[g][i] += rq[g][i];                              // sums of the ROUNDED values: what a colsum over the copy gives
                            amax = fmaxf(amax, fabsf(rq[g][i]));
                        }
                        *(uint2*)(dxb + (size_t)row * lddxb + c) = t.u;
                    }
                }
            }
        }
        if (COPY && q8 != nullptr) {               // wave-uniform
            amax = wave_max(amax);
            const float inv = amax > 0.f ? 448.f / amax : 0.f;
            if (lane == 0) q_scale[row] = amax > 0.f ? amax / 448.f : 1.f;
#pragma unroll
            for (int g = 0; g < MAXG; ++g) {
                const int c = (g * 64 + lane) * 4;
                if (c < Kp) {                      // Kp % 128 == 0: whole 4-byte groups; columns >= C are zero bytes
                    unsigned w = 0;
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(rq[g][0] * inv, rq[g][1] * inv, w, false);
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(rq[g][2] * inv, rq[g][3] * inv, w, true);
                    *(unsigned*)(q8 + (size_t)row * ldq + c) = w;
                }
            }
        }
    }
    if (part == nullptr) return;
    // cross-wave reduction of dgamma/dbeta: waves 1..3 publish, wave 0 sums (fixed order -> deterministic)
    if (wave > 0) {
        float* mine = red + (size_t)(wave - 1) * NR * Cp;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int c = (g * 64 + lane) * 4;
            if (g < ng && c < C) {
                *(float4*)(mine + c) = make_float4(dg[g][0], dg[g][1], dg[g][2], dg[g][3]);
                *(float4*)(mine + Cp + c) = make_float4(db[g][0], db[g][1], db[g][2], db[g][3]);
                if (COPY) *(float4*)(mine + 2 * Cp + c) = make_float4(dc[g][0], dc[g][1], dc[g][2], dc[g][3]);
            }
        }
    }
    __syncthreads();
    if (wave == 0) {
        float* outp = part + (size_t)blockIdx.x * NR * Cp;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int c = (g * 64 + lane) * 4;
            if (g < ng && c < C) {
                float a[4] = {dg[g][0], dg[g][1], dg[g][2], dg[g][3]};
                float b[4] = {db[g][0], db[g][1], db[g][2], db[g][3]};
                float cc[4] = {0.f, 0.f, 0.f, 0.f};
                if (COPY) { cc[0] = dc[g][0]; cc[1] = dc[g][1]; cc[2] = dc[g][2]; cc[3] = dc[g][3]; }
                for (int w = 0; w < 3; ++w) {
                    const float4 t = *(const float4*)(red + (size_t)w * NR * Cp + c);
                    const float4 u = *(const float4*)(red + (size_t)w * NR * Cp + Cp + c);
                    a[0] += t.x; a[1] += t.y; a[2] += t.z; a[3] += t.w;
                    b[0] += u.x; b[1] += u.y; b[2] += u.z; b[3] += u.w;
                    if (COPY) {
                        const float4 v = *(const float4*)(red + (size_t)w * NR * Cp + 2 * Cp + c);
                        cc[0] += v.x; cc[1] += v.y; cc[2] += v.z; cc[3] += v.w;
                    }
                }
                *(float4*)(outp + c) = make_float4(a[0], a[1], a[2], a[3]);
                *(float4*)(outp + Cp + c) = make_float4(b[0], b[1], b[2], b[3]);
                if (COPY) *(float4*)(outp + 2 * Cp + c) = make_float4(cc[0], cc[1], cc[2], cc[3]);
            }
        }
    }
}

// part [nparts][NR][Cp] -> out0[C] (dgamma), out1[C] (dbeta)[, out2[C] (column sums)], each nullable; += when accumulate
// 16 columns per workgroup, 16 thread groups split the partial rows (4 loads in flight each), LDS combine in fixed order
// (deterministic).  96 workgroups for C = 768 instead of 24 with one dependent load chain of 128 per thread.
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ part, int nparts, int C, int NR, float* __restrict__ out0,
                                                              float* __restrict__ out1, float* __restrict__ out2, int accumulate) {
    __shared__ float red[16][16];
    const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int Cp = (C + 3) & ~3;
    const int c = blockIdx.x * 16 + cl;              // index into a [NR][Cp] partial row
    const int r = c / Cp, cc = c - r * Cp;
    float* const outs[3] = {out0, out1, out2};
    const bool live = r < NR && cc < C && outs[r < 3 ? r : 0] != nullptr;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const size_t stride = (size_t)NR * Cp;
    if (live) {
        int p = grp;
        for (; p + 48 < nparts; p += 64) {
            s0 += part[(size_t)p * stride + c];
            s1 += part[(size_t)(p + 16) * stride + c];
            s2 += part[(size_t)(p + 32) * stride + c];
            s3 += part[(size_t)(p + 48) * stride + c];
        }
        for (; p < nparts; p += 16) s0 += part[(size_t)p * stride + c];
    }
    red[grp][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && live) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += red[g][cl];
        float* dst = outs[r] + cc;
        *dst = accumulate ? *dst + s : s;
    }
}

// ---------------------------------------------------------------------------------------------
// y = x / max(||x||, eps) per row (fp32 in, fp32 out); inv_norm saved for the backward.
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ inv_out, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_WG + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * C;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) { const float4 t = *(const float4*)(xr + c); s += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w; }
    const float inv = 1.f / fmaxf(sqrtf(wave_sum(s)), eps);
    for (int c = lane * 4; c < C; c += 256) {
        const float4 t = *(const float4*)(xr + c);
        *(float4*)(y + (size_t)row * C + c) = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    }
    if (lane == 0 && inv_out) inv_out[row] = inv;
}

// dx = (dy - y * <y,dy>) * inv_norm  -> bf16 (GEMM operand for the head dgrad)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ inv, __bf16* __restrict__ dx, int M, int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_WG + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* yr = y + (size_t)row * C;
    const float* gr = dy + (size_t)row * C;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 a = *(const float4*)(yr + c), b = *(const float4*)(gr + c);
        s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    const float dot = wave_sum(s), iv = inv[row];
    for (int c = lane * 4; c < C; c += 256) {
        const float4 a = *(const float4*)(yr + c), b = *(const float4*)(gr + c);
        float o[4] = {(b.x - a.x * dot) * iv, (b.y - a.y * dot) * iv, (b.z - a.z * dot) * iv, (b.w - a.w * dot) * iv};
        store_bf16x4(dx + (size_t)row * C + c, o);
    }
}

}  // namespace

// C ABI ------------------------------------------------------------------------------------------
// x_dtype: 0 = f32, 1 = bf16.  mean/rstd may be null (teacher, no backward).
static int layernorm_fwd_impl(const void* x, int x_dtype, long ldx, const float* gamma, const float* beta, void* y, long ldy, float* mean,
                              float* rstd, int M, int C, float eps, void* q8, long ldq, float* q_scale, hipStream_t stream) {
    // C % 4 != 0 is allowed for zero-padded rows: x, y, gamma, beta must then be readable/writable up to the next multiple of 4
    // (padding zero in x/gamma/beta -> the padding of y comes out zero).
    CS_CHECK_ARG(C <= MAXC && (C % 4 == 0 || (ldx >= ((C + 3) & ~3) && ldy >= ((C + 3) & ~3))), "cs_layernorm_fwd: C=%d unsupported (ld too small for a padded row, or > %d)", C, MAXC);
    CS_CHECK_ARG(M > 0, "cs_layernorm_fwd: empty input");
    dim3 grid((M + ROWS_PER_WG - 1) / ROWS_PER_WG), block(256);
    if (q8 != nullptr) {
        const int Kp = (C + 127) / 128 * 128;
        CS_CHECK_ARG(y != nullptr && q_scale != nullptr && ldq >= Kp && ldq % 4 == 0 && ((uintptr_t)q8 % 4) == 0 && Kp <= 256 * 12,
                     "cs_layernorm_fwd_q8: needs y, the scale vector and 4-byte aligned e4m3 rows of >= %d bytes", Kp);
#define LNQ(TX, NG) hipLaunchKernelGGL((ln_fwd_kernel<TX, NG, __bf16, true>), grid, block, 0, stream, (const TX*)x, ldx, gamma, beta, (__bf16*)y, ldy, mean, rstd, M, C, eps, (unsigned char*)q8, ldq, q_scale, Kp)
        if (x_dtype == 0) { if (Kp <= 1024) LNQ(float, 4); else if (Kp <= 2048) LNQ(float, 8); else LNQ(float, 12); }
        else { if (Kp <= 1024) LNQ(__bf16, 4); else if (Kp <= 2048) LNQ(__bf16, 8); else LNQ(__bf16, 12); }
#undef LNQ
        CS_LAUNCH_CHECK();
        return 0;
    }
#define LNF(TX, NG) hipLaunchKernelGGL((ln_fwd_kernel<TX, NG>), grid, block, 0, stream, (const TX*)x, ldx, gamma, beta, (__bf16*)y, ldy, mean, rstd, M, C, eps)
    if (x_dtype == 0) { if (C <= 1024) LNF(float, 4); else if (C <= 2048) LNF(float, 8); else LNF(float, 12); }
    else { if (C <= 1024) LNF(__bf16, 4); else if (C <= 2048) LNF(__bf16, 8); else LNF(__bf16, 12); }
#undef LNF
    CS_LAUNCH_CHECK();
    return 0;
}
extern "C" int cs_layernorm_fwd(const void* x, int x_dtype, long ldx, const float* gamma, const float* beta, void* y, long ldy,
                                float* mean, float* rstd, int M, int C, float eps, hipStream_t stream) {
    return layernorm_fwd_impl(x, x_dtype, ldx, gamma, beta, y, ldy, mean, rstd, M, C, eps, nullptr, 0, nullptr, stream);
}
// cs_layernorm_fwd that also emits the e4m3 copy of y for the fp8 GEMM: q8 [M, ldq >= C rounded up to 128] bytes (padding zero) and
// q_scale [M] -- bit-identical to cs_quant_rows_fp8(y) (BASELINE configs[4], precision amp_fp8).
extern "C" int cs_layernorm_fwd_q8(const void* x, int x_dtype, long ldx, const float* gamma, const float* beta, void* y, long ldy,
                                   float* mean, float* rstd, void* q8, long ldq, float* q_scale, int M, int C, float eps, hipStream_t stream) {
    CS_CHECK_ARG(q8 != nullptr, "cs_layernorm_fwd_q8: q8 is required (use cs_layernorm_fwd otherwise)");
    return layernorm_fwd_impl(x, x_dtype, ldx, gamma, beta, y, ldy, mean, rstd, M, C, eps, q8, ldq, q_scale, stream);
}

// fp32 in -> fp32 out (y distinct from x): ln_pre of the OpenAI-CLIP ViT, whose output *is* the residual stream
// (open_clip/transformer.py:476-477; LayerNorm under autocast returns fp32).
extern "C" int cs_layernorm_fwd_f32(const float* x, long ldx, const float* gamma, const float* beta, float* y, long ldy, float* mean,
                                    float* rstd, int M, int C, float eps, hipStream_t stream) {
    CS_CHECK_ARG(x && y && gamma && beta && C <= MAXC && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && M > 0,
                 "cs_layernorm_fwd_f32: C=%d unsupported (multiple of 4, <= %d)", C, MAXC);
    CS_CHECK_ARG((mean == nullptr) == (rstd == nullptr), "cs_layernorm_fwd_f32: mean/rstd must both be given or both null");
    dim3 grid((M + ROWS_PER_WG - 1) / ROWS_PER_WG), block(256);
#define LNF32(NG) hipLaunchKernelGGL((ln_fwd_kernel<float, NG, float>), grid, block, 0, stream, x, ldx, gamma, beta, y, ldy, mean, rstd, M, C, eps)
    if (C <= 1024) LNF32(4); else if (C <= 2048) LNF32(8); else LNF32(12);
#undef LNF32
    CS_LAUNCH_CHECK();
    return 0;
}

// part [P][M][2] f32 (sum, sum of squares per slice of npp columns; slices past C are ignored) -> mean, rstd [M] of a C-wide LayerNorm.
extern "C" int cs_ln_stats_finalize(const float* part, int P, int npp, int C, int M, float eps, float* mean, float* rstd, hipStream_t stream) {
    CS_CHECK_ARG(part && mean && rstd && P > 0 && npp > 0 && C > 0 && M > 0 && (long)P * npp >= C,
                 "cs_ln_stats_finalize: bad arguments P=%d npp=%d C=%d M=%d", P, npp, C, M);
    if (M % 2 == 0 && ((uintptr_t)part % 16) == 0)
        hipLaunchKernelGGL(ln_stats_finalize2_kernel, dim3((M / 2 + 255) / 256), dim3(256), 0, stream, part, P, npp, C, M, eps, mean, rstd);
    else
        hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((M + 255) / 256), dim3(256), 0, stream, part, P, npp, C, M, eps, mean, rstd);
    CS_LAUNCH_CHECK();
    return 0;
}

// dx_mode: 0 = write bf16, 1 = write f32, 2 = accumulate into f32 (residual-gradient stream).
// dgamma/dbeta may be null (frozen LN); dx_copy (fp32 modes only, nullable): bf16 copy of the dx rows after the write / accumulate, row
// stride ldcopy, with copy_colsum[C] (nullable) += / = its column sums.  `workspace` must hold cs_layernorm_bwd_workspace(M,C) bytes
// whenever dgamma or copy_colsum is given.
extern "C" size_t cs_layernorm_bwd_workspace(int M, int C) {
    const int nwg = min(512, (M + 3) / 4);
    return (size_t)nwg * 3 * ((C + 3) & ~3) * sizeof(float);
}
static int layernorm_bwd_impl(const void* dy, long lddy, const void* x, int x_dtype, long ldx, const float* gamma, const float* mean,
                              const float* rstd, void* dx, int dx_mode, long lddx, float* dgamma, float* dbeta,
                              int accumulate_params, void* workspace, void* dx_copy, long ldcopy, float* copy_colsum, void* q8, long ldq,
                              float* q_scale, int M, int C, hipStream_t stream) {
    const int Kp = (C + 127) / 128 * 128;
    CS_CHECK_ARG(q8 == nullptr || (dx_copy != nullptr && q_scale != nullptr && ldq >= Kp && ldq % 4 == 0 && ((uintptr_t)q8 % 4) == 0 && Kp <= 256 * 12),
                 "cs_layernorm_bwd_q8: needs dx_copy, the scale vector and 4-byte aligned e4m3 rows of >= %d bytes", Kp);
    CS_CHECK_ARG(C <= MAXC && (C % 4 == 0 || (ldx >= ((C + 3) & ~3) && lddy >= ((C + 3) & ~3) && lddx >= ((C + 3) & ~3))), "cs_layernorm_bwd: C=%d unsupported", C);
    CS_CHECK_ARG(M > 0, "cs_layernorm_bwd: empty input");
    CS_CHECK_ARG(dx_mode >= 0 && dx_mode <= 2, "cs_layernorm_bwd: bad dx_mode");
    CS_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "cs_layernorm_bwd: dgamma/dbeta must both be given or both null");
    CS_CHECK_ARG(dx_copy == nullptr || (dx_mode != 0 && ldcopy >= ((C + 3) & ~3) && ldcopy % 4 == 0 && ((uintptr_t)dx_copy % 8) == 0),
                 "cs_layernorm_bwd: the bf16 copy exists for the fp32 dx modes (8-byte aligned rows, ldcopy >= C rounded up to 4)");
    CS_CHECK_ARG(copy_colsum == nullptr || dx_copy != nullptr, "cs_layernorm_bwd: copy_colsum are the column sums of dx_copy");
    const bool need_part = dgamma != nullptr || copy_colsum != nullptr;
    CS_CHECK_ARG(!need_part || workspace != nullptr, "cs_layernorm_bwd: workspace required for dgamma/dbeta / copy_colsum");
    const int nwg = min(512, (M + 3) / 4);
    float* part = need_part ? (float*)workspace : nullptr;
    const bool copy = dx_copy != nullptr;
    const int NR = copy ? 3 : 2;
    const size_t lds = (size_t)3 * NR * ((C + 3) & ~3) * sizeof(float);
    dim3 grid(nwg), block(256);
#define LNB4(TX, MODE, NG, CP)                                                                                              \
    do {                                                                                                                    \
        static bool once = (hipFuncSetAttribute((const void*)ln_bwd_kernel<TX, MODE, NG, CP>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * MAXC * 4), true); \
        (void)once;                                                                                                         \
        hipLaunchKernelGGL((ln_bwd_kernel<TX, MODE, NG, CP>), grid, block, lds, stream, (const __bf16*)dy, lddy, (const TX*)x, ldx, gamma, mean, rstd, dx, lddx, \
                           (__bf16*)dx_copy, ldcopy, part, M, C, (unsigned char*)q8, ldq, q_scale, Kp);                     \
    } while (0)
#define LNB3(TX, MODE, NG) do { if (copy) { if constexpr (MODE != DX_BF16) LNB4(TX, MODE, NG, true); } else LNB4(TX, MODE, NG, false); } while (0)
#define LNB(TX, MODE) do { if (C <= 1024) LNB3(TX, MODE, 4); else if (C <= 2048) LNB3(TX, MODE, 8); else LNB3(TX, MODE, 12); } while (0)
    if (x_dtype == 0) {
        if (dx_mode == 0) LNB(float, DX_BF16); else if (dx_mode == 1) LNB(float, DX_F32_ASSIGN); else LNB(float, DX_F32_ACCUM);
    } else {
        if (dx_mode == 0) LNB(__bf16, DX_BF16); else if (dx_mode == 1) LNB(__bf16, DX_F32_ASSIGN); else LNB(__bf16, DX_F32_ACCUM);
    }
#undef LNB
#undef LNB3
#undef LNB4
    CS_LAUNCH_CHECK();
    if (need_part) {
        hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((NR * ((C + 3) & ~3) + 15) / 16), dim3(256), 0, stream, part, nwg, C, NR, dgamma, dbeta,
                           copy_colsum, accumulate_params);
        CS_LAUNCH_CHECK();
    }
    return 0;
}
extern "C" int cs_layernorm_bwd(const void* dy, long lddy, const void* x, int x_dtype, long ldx, const float* gamma, const float* mean,
                                const float* rstd, void* dx, int dx_mode, long lddx, float* dgamma, float* dbeta,
                                int accumulate_params, void* workspace, void* dx_copy, long ldcopy, float* copy_colsum, int M, int C,
                                hipStream_t stream) {
    return layernorm_bwd_impl(dy, lddy, x, x_dtype, ldx, gamma, mean, rstd, dx, dx_mode, lddx, dgamma, dbeta, accumulate_params, workspace,
                              dx_copy, ldcopy, copy_colsum, nullptr, 0, nullptr, M, C, stream);
}
// cs_layernorm_bwd whose bf16 copy also leaves as e4m3 bytes q8 [M, ldq >= C rounded up to 128] + fp32 row scales q_scale [M]: the A operand of
// an fp8 dgrad (cs_gemm_nt_f8), bit-identical to cs_quant_rows_fp8(dx_copy)
extern "C" int cs_layernorm_bwd_q8(const void* dy, long lddy, const void* x, int x_dtype, long ldx, const float* gamma, const float* mean,
                                   const float* rstd, void* dx, int dx_mode, long lddx, float* dgamma, float* dbeta,
                                   int accumulate_params, void* workspace, void* dx_copy, long ldcopy, float* copy_colsum, void* q8, long ldq,
                                   float* q_scale, int M, int C, hipStream_t stream) {
    CS_CHECK_ARG(q8 != nullptr, "cs_layernorm_bwd_q8: q8 is required (cs_layernorm_bwd is the form without it)");
    return layernorm_bwd_impl(dy, lddy, x, x_dtype, ldx, gamma, mean, rstd, dx, dx_mode, lddx, dgamma, dbeta, accumulate_params, workspace,
                              dx_copy, ldcopy, copy_colsum, q8, ldq, q_scale, M, C, stream);
}

extern "C" int cs_l2norm_fwd(const float* x, float* y, float* inv_norm, int M, int C, float eps, hipStream_t stream) {
    CS_CHECK_ARG(C % 4 == 0 && M > 0, "cs_l2norm_fwd: C must be a multiple of 4");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((M + ROWS_PER_WG - 1) / ROWS_PER_WG), dim3(256), 0, stream, x, y, inv_norm, M, C, eps);
    CS_LAUNCH_CHECK();
    return 0;
}
extern "C" int cs_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, void* dx_bf16, int M, int C, hipStream_t stream) {
    CS_CHECK_ARG(C % 4 == 0 && M > 0, "cs_l2norm_bwd: C must be a multiple of 4");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((M + ROWS_PER_WG - 1) / ROWS_PER_WG), dim3(256), 0, stream, dy, y, inv_norm, (__bf16*)dx_bf16, M, C);
    CS_LAUNCH_CHECK();
    return 0;
}
