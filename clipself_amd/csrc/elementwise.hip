// HBM-bound helper kernels of the CLIPSelf step: SiLU*mul (fwd/bwd), f32->bf16 cast, padded bf16 transpose,
// column sums (bias gradients), patch im2row, CLS-row fill.  All vectorised to 16-byte accesses per lane.
#include "cs_common.h"

namespace {

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }

// h[m, j] = silu(x12[m, j]) * x12[m, Hd + j]        (reference: SwiGLU.forward, eva_vit_model.py:99-101)
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const __bf16* __restrict__ x12, long ldx, __bf16* __restrict__ h, long ldh,
                                                         int M, int Hd) {
    const int vec_per_row = Hd >> 3;
    const long total = (long)M * vec_per_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / vec_per_row), j = (int)(i - (long)m * vec_per_row) * 8;
        U128 a, b, o;
        a.u = *(const uint4*)(x12 + (size_t)m * ldx + j);
        b.u = *(const uint4*)(x12 + (size_t)m * ldx + Hd + j);
#pragma unroll
        for (int e = 0; e < 8; ++e) o.e[e] = f2bf(silu_f(bf2f(a.e[e])) * bf2f(b.e[e]));
        *(uint4*)(h + (size_t)m * ldh + j) = o.u;
    }
}

// dx1 = dh * x2 * (sig + x1*sig*(1-sig)) ; dx2 = dh * silu(x1)
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const __bf16* __restrict__ dh, long lddh, const __bf16* __restrict__ x12, long ldx,
                                                         __bf16* __restrict__ dx12, long lddx, int M, int Hd) {
    const int vec_per_row = Hd >> 3;
    const long total = (long)M * vec_per_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / vec_per_row), j = (int)(i - (long)m * vec_per_row) * 8;
        U128 a, b, g, o1, o2;
        a.u = *(const uint4*)(x12 + (size_t)m * ldx + j);
        b.u = *(const uint4*)(x12 + (size_t)m * ldx + Hd + j);
        g.u = *(const uint4*)(dh + (size_t)m * lddh + j);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x1 = bf2f(a.e[e]), x2 = bf2f(b.e[e]), d = bf2f(g.e[e]);
            const float sig = 1.f / (1.f + __expf(-x1));
            o1.e[e] = f2bf(d * x2 * (sig + x1 * sig * (1.f - sig)));
            o2.e[e] = f2bf(d * x1 * sig);
        }
        *(uint4*)(dx12 + (size_t)m * lddx + j) = o1.u;
        *(uint4*)(dx12 + (size_t)m * lddx + Hd + j) = o2.u;
    }
}

// swiglu_bwd + the column sums of its output (the bias gradients of w1 | w2) in one pass: the row-block structure of colsum_bf16_kernel -- a
// workgroup owns rows_per_block rows x 512 hidden units, four row phases (ty) x 64 lanes of 8 units -- so the partial row it writes is, bit
// for bit, the one colsum_bf16_kernel would compute from the stored dx12 (same bf16-rounded values, same rows per thread in the same order,
// same (r0 + r1) + (r2 + r3) combine); colsum_reduce_kernel finishes both halves.  Saves the 103 MB re-read of dx12 per block (round 5).
__global__ __launch_bounds__(256) void swiglu_bwd_colsum_kernel(const __bf16* __restrict__ dh, long lddh, const __bf16* __restrict__ x12, long ldx,
                                                                __bf16* __restrict__ dx12, long lddx, float* __restrict__ part, int M, int Hd,
                                                                int rows_per_block) {
    __shared__ float red[2][4][512];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = (blockIdx.x * 64 + tx) * 8;                           // first of the thread's 8 hidden units
    const int m_begin = blockIdx.y * rows_per_block, m_end = min(M, m_begin + rows_per_block);
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (j < Hd) {
        auto one = [&](const U128& a, const U128& b, const U128& g, int m) {
            U128 o1, o2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x1 = bf2f(a.e[e]), x2 = bf2f(b.e[e]), d = bf2f(g.e[e]);
                const float sig = 1.f / (1.f + __expf(-x1));
                o1.e[e] = f2bf(d * x2 * (sig + x1 * sig * (1.f - sig)));
                o2.e[e] = f2bf(d * x1 * sig);
                s1[e] += bf2f(o1.e[e]);
                s2[e] += bf2f(o2.e[e]);
            }
            *(uint4*)(dx12 + (size_t)m * lddx + j) = o1.u;
            *(uint4*)(dx12 + (size_t)m * lddx + Hd + j) = o2.u;
        };
        int m = m_begin + ty;
        for (; m + 4 < m_end; m += 8) {                                 // two rows = six 16-byte loads in flight per thread
            U128 a[2], b[2], g[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                a[u].u = *(const uint4*)(x12 + (size_t)(m + 4 * u) * ldx + j);
                b[u].u = *(const uint4*)(x12 + (size_t)(m + 4 * u) * ldx + Hd + j);
                g[u].u = *(const uint4*)(dh + (size_t)(m + 4 * u) * lddh + j);
            }
            one(a[0], b[0], g[0], m);
            one(a[1], b[1], g[1], m + 4);
        }
        for (; m < m_end; m += 4) {
            U128 a, b, g;
            a.u = *(const uint4*)(x12 + (size_t)m * ldx + j);
            b.u = *(const uint4*)(x12 + (size_t)m * ldx + Hd + j);
            g.u = *(const uint4*)(dh + (size_t)m * lddh + j);
            one(a, b, g, m);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][ty][tx * 8 + e] = s1[e]; red[1][ty][tx * 8 + e] = s2[e]; }
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int col = blockIdx.x * 512 + c;
        if (col < Hd) {
            part[(size_t)blockIdx.y * 2 * Hd + col] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
            part[(size_t)blockIdx.y * 2 * Hd + Hd + col] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
        }
    }
}

// The same with the e4m3 copy of the output row for an fp8 dgrad (cs_gemm_nt_f8): one wave per row; the rounded bf16 outputs stay in
// registers between the amax reduction and v_cvt_pk_fp8_f32, so q8 / q_scale are bit-identical to cs_quant_rows_fp8(dx12) without its
// pass over the 2*Hd-wide matrix.  q8 row = [dx1 codes | dx2 codes | zero bytes up to Kp].  IT x 512 >= Hd.
template <int IT>
__global__ __launch_bounds__(256) void swiglu_bwd_q8_kernel(const __bf16* __restrict__ dh, long lddh, const __bf16* __restrict__ x12, long ldx,
                                                            __bf16* __restrict__ dx12, long lddx, unsigned char* __restrict__ q8, long ldq,
                                                            float* __restrict__ q_scale, int M, int Hd, int Kp) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    U128 a[IT], b[IT], g[IT], o1[IT], o2[IT];
    const int jlast = Hd - 8;
#pragma unroll
    for (int it = 0; it < IT; ++it) {                       // branch-free loads: lanes past the row end re-read its last vector
        const int j = min((it * 64 + lane) * 8, jlast);
        a[it].u = *(const uint4*)(x12 + (size_t)m * ldx + j);
        b[it].u = *(const uint4*)(x12 + (size_t)m * ldx + Hd + j);
        g[it].u = *(const uint4*)(dh + (size_t)m * lddh + j);
    }
    float amax = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int j = (it * 64 + lane) * 8;
        const bool in = j < Hd;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x1 = bf2f(a[it].e[e]), x2 = bf2f(b[it].e[e]), d = bf2f(g[it].e[e]);
            const float sig = 1.f / (1.f + __expf(-x1));
            o1[it].e[e] = f2bf(d * x2 * (sig + x1 * sig * (1.f - sig)));
            o2[it].e[e] = f2bf(d * x1 * sig);
            if (in) amax = fmaxf(amax, fmaxf(fabsf(bf2f(o1[it].e[e])), fabsf(bf2f(o2[it].e[e]))));
        }
        if (in) {
            *(uint4*)(dx12 + (size_t)m * lddx + j) = o1[it].u;
            *(uint4*)(dx12 + (size_t)m * lddx + Hd + j) = o2[it].u;
        }
    }
    amax = wave_max(amax);
    const float inv = amax > 0.f ? 448.f / amax : 0.f;
    if (lane == 0) q_scale[m] = amax > 0.f ? amax / 448.f : 1.f;
    auto cvt8 = [&](const U128& v) {
        unsigned lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v.e[0]) * inv, bf2f(v.e[1]) * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v.e[2]) * inv, bf2f(v.e[3]) * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v.e[4]) * inv, bf2f(v.e[5]) * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v.e[6]) * inv, bf2f(v.e[7]) * inv, hi, true);
        return make_uint2(lo, hi);
    };
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int j = (it * 64 + lane) * 8;
        if (j < Hd) {
            *(uint2*)(q8 + m * ldq + j) = cvt8(o1[it]);
            *(uint2*)(q8 + m * ldq + Hd + j) = cvt8(o2[it]);
        }
    }
    for (int k = 2 * Hd + lane * 8; k < Kp; k += 512) *(uint2*)(q8 + m * ldq + k) = make_uint2(0u, 0u);      // 2*Hd % 8 == 0, Kp % 128 == 0
}

// MLP activation of the OpenAI-CLIP ViT blocks (open_clip/transformer.py:209-213): nn.GELU (exact, erf) or QuickGELU
// x*sigmoid(1.702x) (:31-34, forced for the `openai` weights).  Forward on the bf16 c_fc output kept for backward; fp32 inside.
template <bool QUICK>
__device__ __forceinline__ float gelu_f(float x) {
    if (QUICK) return x / (1.f + __expf(-1.702f * x));
    return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
}
template <bool QUICK>
__device__ __forceinline__ float gelu_grad_f(float x) {
    if (QUICK) {
        const float s = 1.f / (1.f + __expf(-1.702f * x));
        return s * (1.f + 1.702f * x * (1.f - s));
    }
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

template <bool QUICK>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const __bf16* __restrict__ x, long ldx, __bf16* __restrict__ y, long ldy, int M, int N) {
    const int vec_per_row = N >> 3;
    const long total = (long)M * vec_per_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / vec_per_row), j = (int)(i - (long)m * vec_per_row) * 8;
        U128 a, o;
        a.u = *(const uint4*)(x + (size_t)m * ldx + j);
#pragma unroll
        for (int e = 0; e < 8; ++e) o.e[e] = f2bf(gelu_f<QUICK>(bf2f(a.e[e])));
        *(uint4*)(y + (size_t)m * ldy + j) = o.u;
    }
}

// dx = dy * act'(x)
template <bool QUICK>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const __bf16* __restrict__ dy, long lddy, const __bf16* __restrict__ x, long ldx,
                                                       __bf16* __restrict__ dx, long lddx, int M, int N) {
    const int vec_per_row = N >> 3;
    const long total = (long)M * vec_per_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / vec_per_row), j = (int)(i - (long)m * vec_per_row) * 8;
        U128 a, g, o;
        a.u = *(const uint4*)(x + (size_t)m * ldx + j);
        g.u = *(const uint4*)(dy + (size_t)m * lddy + j);
#pragma unroll
        for (int e = 0; e < 8; ++e) o.e[e] = f2bf(bf2f(g.e[e]) * gelu_grad_f<QUICK>(bf2f(a.e[e])));
        *(uint4*)(dx + (size_t)m * lddx + j) = o.u;
    }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, long n8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const float4 a = *(const float4*)(x + i * 8), b = *(const float4*)(x + i * 8 + 4);
        U128 o;
        o.e[0] = f2bf(a.x); o.e[1] = f2bf(a.y); o.e[2] = f2bf(a.z); o.e[3] = f2bf(a.w);
        o.e[4] = f2bf(b.x); o.e[5] = f2bf(b.y); o.e[6] = f2bf(b.z); o.e[7] = f2bf(b.w);
        *(uint4*)(y + i * 8) = o.u;
    }
}

// out[c, r] = in[r, c] for r < R, 0 for R <= r < ld_out.  64x64 tiles through LDS (+1 dword pad), bf16.
// Used to build the contraction-major operands of the weight-gradient GEMMs and the transposed weight shadows.
__device__ __forceinline__ void transpose_tile(const __bf16* __restrict__ in, long ld_in, __bf16* __restrict__ out, long ld_out, int R, int Cc,
                                               int r0, int c0, uint16_t (*tile)[66]) {
    const uint16_t* src = (const uint16_t*)in;
    uint16_t* dst = (uint16_t*)out;
    const bool vec = ((ld_in | ld_out) & 7) == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0 && c0 + 64 <= Cc && r0 + 64 <= ld_out;
    if (vec) {
        // 16-byte global accesses both ways: 8 row-vectors in, 8 column-vectors out per thread pair of passes
#pragma unroll
        for (int pss = 0; pss < 2; ++pss) {
            const int v = threadIdx.x + pss * 256, r = v >> 3, cv = (v & 7) * 8;      // 64 rows x 8 vectors
            U128 t;
            if (r0 + r < R) t.u = *(const uint4*)(src + (size_t)(r0 + r) * ld_in + c0 + cv);
            else t.u = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[r][cv + e] = ((const uint16_t*)&t)[e];
        }
        __syncthreads();
#pragma unroll
        for (int pss = 0; pss < 2; ++pss) {
            const int v = threadIdx.x + pss * 256, c = v >> 3, rv = (v & 7) * 8;      // 64 out rows x 8 vectors
            U128 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) ((uint16_t*)&o)[e] = tile[rv + e][c];
            *(uint4*)(dst + (size_t)(c0 + c) * ld_out + r0 + rv) = o.u;
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;    // ty 0..3
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty * 16 + i, c = c0 + tx;
        tile[ty * 16 + i][tx] = (r < R && c < Cc) ? src[(size_t)r * ld_in + c] : (uint16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty * 16 + i, r = r0 + tx;
        if (c < Cc && r < ld_out) dst[(size_t)c * ld_out + r] = tile[tx][ty * 16 + i];
    }
}

__global__ __launch_bounds__(256) void transpose_bf16_kernel(const __bf16* __restrict__ in, long ld_in, __bf16* __restrict__ out,
                                                             long ld_out, int R, int Cc) {
    __shared__ uint16_t tile[64][66];
    transpose_tile(in, ld_in, out, ld_out, R, Cc, blockIdx.y * 64, blockIdx.x * 64, tile);
}

// Many independent transposes in ONE launch (the W^T shadows of every trainable block after an AdamW step: 49 matrices for B/16, a
// launch each took 5.5 us for ~1 us of traffic).  desc[i] = {in, out, ld_in, ld_out, R, Cc, first tile, tiles per row of tiles}; a
// workgroup finds its matrix by a linear scan of the (few dozen) first-tile entries.
struct TransposeDesc {
    const __bf16* in;
    __bf16* out;
    long ld_in, ld_out;
    int R, Cc, tile0, tiles_x;
};
__global__ __launch_bounds__(256) void transpose_bf16_batched_kernel(const TransposeDesc* __restrict__ desc, int count) {
    __shared__ uint16_t tile[64][66];
    const int t = blockIdx.x;
    int i = 0;
    while (i + 1 < count && desc[i + 1].tile0 <= t) ++i;           // wave-uniform (scalar loads)
    const TransposeDesc d = desc[i];
    const int local = t - d.tile0;
    transpose_tile(d.in, d.ld_in, d.out, d.ld_out, d.R, d.Cc, (local / d.tiles_x) * 64, (local % d.tiles_x) * 64, tile);
}

// out[n] += sum_m x[m, n]   (bias gradients; bf16 in, f32 out).  grid.x tiles columns, grid.y splits rows into blocks whose partial sums
// go to a workspace row each; colsum_reduce_kernel adds them up in a fixed order.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const __bf16* __restrict__ x, long ldx, float* __restrict__ out, int M, int N,
                                                          int rows_per_block) {
    // 64 column-vectors (8 bf16 = 16 bytes each) x 4 row phases per workgroup; LDS combine, then one atomic per column
    __shared__ float red[4][512];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = (blockIdx.x * 64 + tx) * 8;
    const int m_begin = blockIdx.y * rows_per_block, m_end = min(M, m_begin + rows_per_block);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool vec = (ldx & 7) == 0 && ((uintptr_t)x & 15) == 0;
    if (n + 8 <= N && vec) {
        int m = m_begin + ty;
        for (; m + 12 < m_end; m += 16) {                 // four independent 16-byte loads in flight per thread
            U128 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u].u = *(const uint4*)(x + (size_t)(m + 4 * u) * ldx + n);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += bf2f(v[u].e[e]);
        }
        for (; m < m_end; m += 4) {
            U128 v;
            v.u = *(const uint4*)(x + (size_t)m * ldx + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += bf2f(v.e[e]);
        }
    } else {
        for (int m = m_begin + ty; m < m_end; m += 4)
            for (int e = 0; e < 8; ++e)
                if (n + e < N) s[e] += bf2f(x[(size_t)m * ldx + n + e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = s[e];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int col = blockIdx.x * 512 + c;
        if (col < N) out[(size_t)blockIdx.y * N + col] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);      // partial row of this row block
    }
}

// out[n] += sum over the row blocks' partials in a fixed order (no atomics: the bias gradients are bit-reproducible):
// (s0 + s1) + (s2 + s3) with s_r = the partials of blocks r, r + 4, r + 8, ... added in ascending order.  64 columns x the 4 residues per
// workgroup, eight loads in flight per thread (round 4; the one-thread-per-column form walked ~200 dependent L2 round trips: 16.6 us for
// 3 MB of partials, now a few microseconds).  Deterministic, but NOT bit-identical to that older kernel when nblocks % 4 != 0: it added the
// leftover blocks to s0, here every block goes to its residue's sum.
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ part, int nblocks, int N, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (n < N) {
        int b = r;
        for (; b + 28 < nblocks; b += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + 4 * u) * N + n];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nblocks; b += 4) s += part[(size_t)b * N + n];
    }
    red[r][tx] = s;
    __syncthreads();
    if (r == 0 && n < N) out[n] += (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

// im2row for the patch-embed conv as a GEMM (reference: nn.Conv2d(3,C,p,stride=p), eva_vit_model.py:348,355).
// out[(b*g*g + py*g + px), (c*p + iy)*p + ix] = img[b, c, py*p + iy, px*p + ix]      ((c,iy,ix) = conv-weight order)
template <typename TI>
__global__ __launch_bounds__(256) void im2row_kernel(const TI* __restrict__ img, __bf16* __restrict__ out, int B, int S, int p, int g, int ldo) {
    const int segs = p / 8;                         // 8-pixel segments per patch row (16-byte bf16 stores)
    const long total = (long)B * g * g * 3 * p * segs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long t = i;
        const int sg = (int)(t % segs); t /= segs;
        const int iy = (int)(t % p); t /= p;
        const int c = (int)(t % 3); t /= 3;
        const int px = (int)(t % g); t /= g;
        const int py = (int)(t % g); t /= g;
        const int b = (int)t;
        const TI* s = img + (((size_t)b * 3 + c) * S + (py * p + iy)) * S + px * p + sg * 8;
        U128 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.e[e] = f2bf((float)s[e]);
        const size_t row = ((size_t)b * g + py) * g + px;
        *(uint4*)(out + row * ldo + (c * p + iy) * p + sg * 8) = o.u;
    }
}

// generic patch size (e.g. 14): one thread per output element, zero-fills the padding columns [3*p*p, ldo)
template <typename TI>
__global__ __launch_bounds__(256) void im2row_generic_kernel(const TI* __restrict__ img, __bf16* __restrict__ out, int B, int S, int p, int g, int ldo) {
    const long total = (long)B * g * g * ldo;
    const int kk = 3 * p * p;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % ldo);
        const long row = i / ldo;
        float v = 0.f;
        if (col < kk) {
            const int ix = col % p, iy = (col / p) % p, c = col / (p * p);
            const int px = (int)(row % g), py = (int)((row / g) % g), b = (int)(row / ((long)g * g));
            v = (float)img[(((size_t)b * 3 + c) * S + (py * p + iy)) * S + px * p + ix];
        }
        out[i] = f2bf(v);
    }
}

// x[b, 0, :] = cls + pos[0, :]   (eva_vit_model.py:540-543)
__global__ void cls_row_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos, int B, int Ntok, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * C) return;
    const int b = (int)(i / C), c = (int)(i - (long)b * C);
    x[(size_t)b * Ntok * C + c] = cls[c] + pos[c];
}

inline int grid_for(long total, int block = 256, int cap = 256 * 16) {
    long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

// C ABI ------------------------------------------------------------------------------------------
extern "C" int cs_swiglu_fwd(const void* x12, long ldx, void* h, long ldh, int M, int Hd, hipStream_t stream) {
    CS_CHECK_ARG(Hd % 8 == 0 && ldx % 8 == 0 && ldh % 8 == 0 && M > 0, "cs_swiglu_fwd: Hd/ld must be multiples of 8");
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for((long)M * (Hd / 8))), dim3(256), 0, stream, (const __bf16*)x12, ldx, (__bf16*)h, ldh, M, Hd);
    CS_LAUNCH_CHECK();
    return 0;
}
extern "C" int cs_swiglu_bwd(const void* dh, long lddh, const void* x12, long ldx, void* dx12, long lddx, int M, int Hd, hipStream_t stream) {
    CS_CHECK_ARG(Hd % 8 == 0 && ldx % 8 == 0 && lddh % 8 == 0 && lddx % 8 == 0 && M > 0, "cs_swiglu_bwd: Hd/ld must be multiples of 8");
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for((long)M * (Hd / 8))), dim3(256), 0, stream, (const __bf16*)dh, lddh, (const __bf16*)x12, ldx,
                       (__bf16*)dx12, lddx, M, Hd);
    CS_LAUNCH_CHECK();
    return 0;
}
constexpr int COLSUM_ROWS = 64;          // rows per block: 768 columns x 12608 rows -> 394 workgroups (100 were latency-bound at 0.7 TB/s)
// cs_swiglu_bwd + colsum[2*Hd] += the column sums of dx12 (bias gradients of w1 | w2), bit-identical to cs_swiglu_bwd followed by
// cs_colsum_bf16(dx12) (workspace: cs_colsum_workspace(M, 2*Hd) bytes).  Hd % 8 == 0.
extern "C" int cs_swiglu_bwd_colsum(const void* dh, long lddh, const void* x12, long ldx, void* dx12, long lddx, float* colsum, void* workspace,
                                    int M, int Hd, hipStream_t stream) {
    CS_CHECK_ARG(Hd % 8 == 0 && ldx % 8 == 0 && lddh % 8 == 0 && lddx % 8 == 0 && M > 0, "cs_swiglu_bwd_colsum: Hd/ld must be multiples of 8");
    CS_CHECK_ARG(colsum != nullptr && workspace != nullptr, "cs_swiglu_bwd_colsum: colsum and workspace (cs_colsum_workspace(M, 2*Hd) bytes) are required");
    const int nblocks = (M + COLSUM_ROWS - 1) / COLSUM_ROWS;
    hipLaunchKernelGGL(swiglu_bwd_colsum_kernel, dim3((Hd + 511) / 512, nblocks), dim3(256), 0, stream, (const __bf16*)dh, lddh, (const __bf16*)x12, ldx,
                       (__bf16*)dx12, lddx, (float*)workspace, M, Hd, COLSUM_ROWS);
    CS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((2 * Hd + 63) / 64), dim3(256), 0, stream, (const float*)workspace, nblocks, 2 * Hd, colsum);
    CS_LAUNCH_CHECK();
    return 0;
}
// cs_swiglu_bwd + the e4m3 copy of dx12 for an fp8 dgrad: q8 [M, ldq >= Kp = 2*Hd rounded up to 128] bytes, q_scale [M]; bit-identical to
// cs_quant_rows_fp8(dx12).  Hd <= 4096.
extern "C" int cs_swiglu_bwd_q8(const void* dh, long lddh, const void* x12, long ldx, void* dx12, long lddx, void* q8, long ldq, float* q_scale,
                                int M, int Hd, hipStream_t stream) {
    const int Kp = (2 * Hd + 127) / 128 * 128;
    CS_CHECK_ARG(Hd % 8 == 0 && Hd >= 8 && Hd <= 4096 && ldx % 8 == 0 && lddh % 8 == 0 && lddx % 8 == 0 && M > 0, "cs_swiglu_bwd_q8: Hd=%d (%% 8, <= 4096), ld %% 8", Hd);
    CS_CHECK_ARG(q8 && q_scale && ldq >= Kp && ldq % 8 == 0 && ((uintptr_t)q8 % 8) == 0, "cs_swiglu_bwd_q8: q8 rows of >= %d bytes, 8-byte aligned", Kp);
    const dim3 grid((M + 3) / 4), block(256);
#define SWQ(IT) hipLaunchKernelGGL(swiglu_bwd_q8_kernel<IT>, grid, block, 0, stream, (const __bf16*)dh, lddh, (const __bf16*)x12, ldx, (__bf16*)dx12, lddx, \
                                   (unsigned char*)q8, ldq, q_scale, M, Hd, Kp)
    if (Hd <= 2048) SWQ(4); else if (Hd <= 3072) SWQ(6); else SWQ(8);
#undef SWQ
    CS_LAUNCH_CHECK();
    return 0;
}
// y = act(x), dx = dy * act'(x) on bf16 [M, N] matrices; quick 0 = exact (erf) GELU, 1 = QuickGELU.  y may alias x only in the forward.
extern "C" int cs_gelu_fwd(const void* x, long ldx, void* y, long ldy, int M, int N, int quick, hipStream_t stream) {
    CS_CHECK_ARG(x && y && N % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && M > 0 && N > 0, "cs_gelu_fwd: N/ld must be multiples of 8");
    const dim3 grid(grid_for((long)M * (N / 8))), block(256);
    if (quick) hipLaunchKernelGGL(gelu_fwd_kernel<true>, grid, block, 0, stream, (const __bf16*)x, ldx, (__bf16*)y, ldy, M, N);
    else hipLaunchKernelGGL(gelu_fwd_kernel<false>, grid, block, 0, stream, (const __bf16*)x, ldx, (__bf16*)y, ldy, M, N);
    CS_LAUNCH_CHECK();
    return 0;
}
extern "C" int cs_gelu_bwd(const void* dy, long lddy, const void* x, long ldx, void* dx, long lddx, int M, int N, int quick, hipStream_t stream) {
    CS_CHECK_ARG(dy && x && dx && N % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && M > 0 && N > 0,
                 "cs_gelu_bwd: N/ld must be multiples of 8");
    const dim3 grid(grid_for((long)M * (N / 8))), block(256);
    if (quick) hipLaunchKernelGGL(gelu_bwd_kernel<true>, grid, block, 0, stream, (const __bf16*)dy, lddy, (const __bf16*)x, ldx, (__bf16*)dx, lddx, M, N);
    else hipLaunchKernelGGL(gelu_bwd_kernel<false>, grid, block, 0, stream, (const __bf16*)dy, lddy, (const __bf16*)x, ldx, (__bf16*)dx, lddx, M, N);
    CS_LAUNCH_CHECK();
    return 0;
}
extern "C" int cs_cast_f32_bf16(const float* x, void* y, long n, hipStream_t stream) {
    CS_CHECK_ARG(n > 0 && n % 8 == 0, "cs_cast_f32_bf16: n must be a positive multiple of 8");
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, x, (__bf16*)y, n / 8);
    CS_LAUNCH_CHECK();
    return 0;
}
extern "C" int cs_transpose_bf16(const void* in, long ld_in, void* out, long ld_out, int R, int Cc, hipStream_t stream) {
    CS_CHECK_ARG(R > 0 && Cc > 0 && ld_out >= R, "cs_transpose_bf16: bad shape R=%d C=%d ld_out=%ld", R, Cc, ld_out);
    dim3 grid((Cc + 63) / 64, (int)((ld_out + 63) / 64));
    hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, stream, (const __bf16*)in, ld_in, (__bf16*)out, ld_out, R, Cc);
    CS_LAUNCH_CHECK();
    return 0;
}
// desc: DEVICE array of `count` records {in, out, ld_in, ld_out, R, Cc, tile0, tiles_x} (8-byte pointers / longs, 4-byte ints: 48 bytes
// each, see TransposeDesc), tile0 ascending = running sum of ceil(Cc/64) * ceil(ld_out/64); total_tiles = the final sum.
extern "C" int cs_transpose_bf16_batched(const void* desc, int count, int total_tiles, hipStream_t stream) {
    CS_CHECK_ARG(desc != nullptr && count > 0 && total_tiles > 0, "cs_transpose_bf16_batched: empty batch");
    static_assert(sizeof(TransposeDesc) == 48, "descriptor layout is part of the C ABI");
    hipLaunchKernelGGL(transpose_bf16_batched_kernel, dim3(total_tiles), dim3(256), 0, stream, (const TransposeDesc*)desc, count);
    CS_LAUNCH_CHECK();
    return 0;
}
extern "C" size_t cs_colsum_workspace(int M, int N) {
    return (size_t)((M + COLSUM_ROWS - 1) / COLSUM_ROWS) * (size_t)(N > 0 ? N : 0) * sizeof(float);
}
extern "C" int cs_colsum_bf16(const void* x, long ldx, float* out, void* workspace, int M, int N, hipStream_t stream) {
    CS_CHECK_ARG(M > 0 && N > 0 && workspace != nullptr, "cs_colsum_bf16: empty input or no workspace (cs_colsum_workspace bytes)");
    const int nblocks = (M + COLSUM_ROWS - 1) / COLSUM_ROWS;
    dim3 grid((N + 511) / 512, nblocks);
    hipLaunchKernelGGL(colsum_bf16_kernel, grid, dim3(256), 0, stream, (const __bf16*)x, ldx, (float*)workspace, M, N, COLSUM_ROWS);
    CS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((N + 63) / 64), dim3(256), 0, stream, (const float*)workspace, nblocks, N, out);
    CS_LAUNCH_CHECK();
    return 0;
}
// img_dtype: 0 = f32, 1 = bf16
extern "C" int cs_im2row(const void* img, int img_dtype, void* out, int B, int S, int p, int ldo, hipStream_t stream) {
    CS_CHECK_ARG(S % p == 0 && B > 0 && ldo >= 3 * p * p, "cs_im2row: patch size must divide S and ldo >= 3*p*p (p=%d S=%d ldo=%d)", p, S, ldo);
    const int g = S / p;
    if (p % 8 == 0 && ldo == 3 * p * p) {
        const long total = (long)B * g * g * 3 * p * (p / 8);
        if (img_dtype == 0)
            hipLaunchKernelGGL((im2row_kernel<float>), dim3(grid_for(total)), dim3(256), 0, stream, (const float*)img, (__bf16*)out, B, S, p, g, ldo);
        else
            hipLaunchKernelGGL((im2row_kernel<__bf16>), dim3(grid_for(total)), dim3(256), 0, stream, (const __bf16*)img, (__bf16*)out, B, S, p, g, ldo);
    } else {
        const long total = (long)B * g * g * ldo;
        if (img_dtype == 0)
            hipLaunchKernelGGL((im2row_generic_kernel<float>), dim3(grid_for(total)), dim3(256), 0, stream, (const float*)img, (__bf16*)out, B, S, p, g, ldo);
        else
            hipLaunchKernelGGL((im2row_generic_kernel<__bf16>), dim3(grid_for(total)), dim3(256), 0, stream, (const __bf16*)img, (__bf16*)out, B, S, p, g, ldo);
    }
    CS_LAUNCH_CHECK();
    return 0;
}
extern "C" int cs_cls_row(float* x, const float* cls, const float* pos, int B, int Ntok, int C, hipStream_t stream) {
    CS_CHECK_ARG(B > 0, "cs_cls_row: empty batch");
    hipLaunchKernelGGL(cls_row_kernel, dim3((int)(((long)B * C + 255) / 256)), dim3(256), 0, stream, x, cls, pos, B, Ntok, C);
    CS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ fp8 operands (BASELINE configs[4])
// Row-wise e4m3 quantisation of a bf16 matrix for the fp8 MFMA GEMM (cs_gemm_nt_f8): q[m, k] = RNE_e4m3(x[m, k] * 448 / amax_m),
// scale[m] = amax_m / 448 (1 for an all-zero row), columns K .. Kp-1 of q zero (the GEMM contracts over Kp, a multiple of 128).
// One wave per row: the row stays in registers between the amax reduction and the conversion (K <= 8192), v_cvt_pk_fp8_f32 (OCP e4m3fn).
namespace {

template <int IT>                                           // IT x (64 lanes x 8 elements) columns: 8 -> 4096, 16 -> 8192
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const __bf16* __restrict__ x, long ldx, unsigned char* __restrict__ q, long ldq,
                                                             float* __restrict__ scale, int M, int K, int Kp) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    U128 v[IT];
    float amax = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int k = (it * 64 + lane) * 8;
        if (k < K) {                                       // K % 8 == 0
            v[it].u = *(const uint4*)(x + row * ldx + k);
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(bf2f(v[it].e[e])));
        } else {
            v[it].u = make_uint4(0, 0, 0, 0);
        }
    }
    amax = wave_max(amax);
    const float inv = amax > 0.f ? 448.f / amax : 0.f;
    if (lane == 0) scale[row] = amax > 0.f ? amax / 448.f : 1.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int k = (it * 64 + lane) * 8;
        if (k < Kp) {
            unsigned lo = 0, hi = 0;
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[it].e[0]) * inv, bf2f(v[it].e[1]) * inv, lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[it].e[2]) * inv, bf2f(v[it].e[3]) * inv, lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[it].e[4]) * inv, bf2f(v[it].e[5]) * inv, hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[it].e[6]) * inv, bf2f(v[it].e[7]) * inv, hi, true);
            *(uint2*)(q + row * ldq + k) = make_uint2(lo, hi);
        }
    }
}

}  // namespace

// x bf16 [M, K] (row stride ldx elements) -> q e4m3 [M, Kp] (row stride ldq bytes, Kp = K rounded up to 128, padding zeroed), scale f32 [M]
extern "C" int cs_quant_rows_fp8(const void* x, long ldx, void* q, long ldq, float* scale, int M, int K, hipStream_t stream) {
    const int Kp = (K + 127) / 128 * 128;
    CS_CHECK_ARG(M > 0 && K > 0 && K % 8 == 0 && K <= 8192, "cs_quant_rows_fp8: K=%d must be a multiple of 8, at most 8192", K);
    CS_CHECK_ARG(ldx % 8 == 0 && ldq >= Kp && ldq % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)q % 8) == 0,
                 "cs_quant_rows_fp8: rows must be 16-byte (input) / 8-byte (output) aligned and ldq >= %d", Kp);
    if (K <= 4096) hipLaunchKernelGGL(quant_rows_fp8_kernel<8>, dim3((M + 3) / 4), dim3(256), 0, stream, (const __bf16*)x, ldx, (unsigned char*)q, ldq, scale, M, K, Kp);
    else hipLaunchKernelGGL(quant_rows_fp8_kernel<16>, dim3((M + 3) / 4), dim3(256), 0, stream, (const __bf16*)x, ldx, (unsigned char*)q, ldq, scale, M, K, Kp);
    CS_LAUNCH_CHECK();
    return 0;
}
