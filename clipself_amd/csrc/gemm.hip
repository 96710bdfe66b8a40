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
// Tiling: 128x128x64 per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA 32x32x16
// tiles, 64 fp32 accumulators/lane).  Operands are staged global -> LDS with the direct
// `global_load_lds` 16-byte DMA (lane-linear LDS image; the XOR bank swizzle is applied to the
// per-lane *source* chunk and again on the ds_read_b128 side), double buffered, one barrier per
// K tile.  Workgroup ids are remapped so that each XCD (private L2) owns a contiguous range of
// M panels and walks the N tiles of a panel back to back (A panel re-use stays in that L2).
#include "cs_common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;   // 16 KiB per operand tile

enum Epi { EPI_BF16 = 0, EPI_F32 = 1, EPI_RESID_F32 = 2, EPI_SWIGLU_BF16 = 3, EPI_ATOMIC_F32 = 4, EPI_PATCH_F32 = 5 };

struct GemmArgs {
    const __bf16* A;
    const __bf16* B;
    void* C;
    const float* bias;    // [N] or null
    const float* extra;   // EPI_RESID: residual [M,ldc] f32 ; EPI_PATCH: pos table [group+1, ldc] f32
    int M, N, K;
    int lda, ldb, ldc;
    int tiles_m, tiles_n;
    int ktiles_per_split;
    int group;            // EPI_PATCH: tokens-1 per image ; EPI_SWIGLU: hidden width Hd
};

__device__ __forceinline__ void stage_tile(const __bf16* __restrict__ src, int ld, int k0, char* lds_tile,
                                           int wave, int lane, const int (&grow)[4], bool use_glds) {
    // tile = 128 rows x 8 chunks(16 B).  Wave w, step i covers rows (w*4+i)*8 .. +8; lane -> (row&7 = lane>>3, slot = lane&7)
    // and fetches source chunk slot ^ (row&7) so that LDS holds chunk c of row r at slot c ^ (r&7).
    const int chunk = (lane & 7) ^ (lane >> 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16* g = src + (size_t)grow[i] * ld + k0 + chunk * 8;
        char* dst = lds_tile + (wave * 4 + i) * 1024;     // wave-uniform
        if (use_glds) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        } else {
            *(uint4*)(dst + lane * 16) = *(const uint4*)g;
        }
    }
}

template <int EPI, bool GLDS>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // buffer b: A tile at smem + b*2*TILE_BYTES, B tile right after it

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int hf = lane >> 5, l31 = lane & 31;

    // ---- XCD-aware bijective remap of the 1-D workgroup id (block b runs on XCD b % 8)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tn = swz % p.tiles_n, tm = swz / p.tiles_n;
    const int m0 = tm * BM;
    const int n0 = tn * BN;          // for EPI_SWIGLU: tile-local packing, see below

    // ---- source rows for the 4 staging steps of this wave
    int arow[4], brow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tr = (wave * 4 + i) * 8 + (lane >> 3);       // tile-local row
        arow[i] = min(m0 + tr, p.M - 1);
        if (EPI == EPI_SWIGLU_BF16) {
            // tile rows [wn'*64 + jj*32 + t] <- weight row jj*Hd + (tn*64 + wn'*32 + t): x1 and x2 of the
            // same hidden unit land in the same lane/register of accumulator tiles j=0 / j=1.
            const int hidx = tn * 64 + (tr >> 6) * 32 + (tr & 31);
            brow[i] = ((tr >> 5) & 1) * p.group + min(hidx, p.group - 1);
        } else {
            brow[i] = min(n0 + tr, p.N - 1);
        }
    }

    const int kt_begin = blockIdx.y * p.ktiles_per_split;
    const int kt_end = min(kt_begin + p.ktiles_per_split, p.K / BK);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (kt_begin < kt_end) {
        stage_tile(p.A, p.lda, kt_begin * BK, smem, wave, lane, arow, GLDS);
        stage_tile(p.B, p.ldb, kt_begin * BK, smem + TILE_BYTES, wave, lane, brow, GLDS);
    }
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        __syncthreads();            // tile kt landed (the barrier drains the LDS-DMA queue); buffer cur^1 is free
        if (kt + 1 < kt_end) {
            stage_tile(p.A, p.lda, (kt + 1) * BK, smem + (cur ^ 1) * 2 * TILE_BYTES, wave, lane, arow, GLDS);
            stage_tile(p.B, p.ldb, (kt + 1) * BK, smem + (cur ^ 1) * 2 * TILE_BYTES + TILE_BYTES, wave, lane, brow, GLDS);
        }
        const char* la = smem + cur * 2 * TILE_BYTES + (wm * 64 + l31) * 128;
        const char* lb = smem + cur * 2 * TILE_BYTES + TILE_BYTES + (wn * 64 + l31) * 128;
        const int sw = lane & 7;    // (row & 7): the 32/64-row offsets are multiples of 8
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int off = (((ks * 2 + hf) ^ sw) << 4);
            bf16x8 a0 = *(const bf16x8*)(la + off);
            bf16x8 a1 = *(const bf16x8*)(la + 32 * 128 + off);
            bf16x8 b0 = *(const bf16x8*)(lb + off);
            bf16x8 b1 = *(const bf16x8*)(lb + 32 * 128 + off);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    // ---- epilogue: lane owns column (l31) of each 32x32 tile; register e <-> row mfma32_row(e, lane)
    if (EPI == EPI_SWIGLU_BF16) {
        const int hcol = tn * 64 + wn * 32 + l31;
        if (hcol < p.group) {
            const float b1 = p.bias ? p.bias[hcol] : 0.f;
            const float b2 = p.bias ? p.bias[p.group + hcol] : 0.f;
            __bf16* out = (__bf16*)p.C;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wm * 64 + i * 32 + mfma32_row(e, lane);
                    if (row < p.M) {
                        const float x1 = acc[i][0][e] + b1, x2 = acc[i][1][e] + b2;
                        out[(size_t)row * p.ldc + hcol] = f2bf(x1 / (1.f + __expf(-x1)) * x2);
                    }
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        if (col >= p.N) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + mfma32_row(e, lane);
                if (row >= p.M) continue;
                const float v = acc[i][j][e] + bv;
                if (EPI == EPI_BF16) {
                    ((__bf16*)p.C)[(size_t)row * p.ldc + col] = f2bf(v);
                } else if (EPI == EPI_F32) {
                    ((float*)p.C)[(size_t)row * p.ldc + col] = v;
                } else if (EPI == EPI_RESID_F32) {
                    const size_t o = (size_t)row * p.ldc + col;
                    ((float*)p.C)[o] = p.extra[o] + v;
                } else if (EPI == EPI_ATOMIC_F32) {
                    unsafeAtomicAdd(((float*)p.C) + (size_t)row * p.ldc + col, v);
                } else if (EPI == EPI_PATCH_F32) {
                    const int img = row / p.group, t = row - img * p.group;
                    ((float*)p.C)[(size_t)(row + img + 1) * p.ldc + col] = v + p.extra[(size_t)(t + 1) * p.ldc + col];
                }
            }
    }
}

template <int EPI>
int launch(const GemmArgs& a, int splits, int use_glds, hipStream_t stream) {
    dim3 grid(a.tiles_m * a.tiles_n, splits), block(256);
    const size_t lds = 4 * TILE_BYTES;
    if (use_glds) {
        static bool once = (hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI, true>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
        (void)once;
        hipLaunchKernelGGL((gemm_nt_kernel<EPI, true>), grid, block, lds, stream, a);
    } else {
        static bool once = (hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI, false>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
        (void)once;
        hipLaunchKernelGGL((gemm_nt_kernel<EPI, false>), grid, block, lds, stream, a);
    }
    CS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// C ABI ------------------------------------------------------------------------------------------
// epi: 0 bf16 out (+bias) | 1 f32 out (+bias) | 2 f32 out = extra(residual) + acc + bias |
//      3 fused SwiGLU (B = [W1;W2] stacked [2*group, K], bias [2*group], out bf16 [M, group]) |
//      4 f32 atomic accumulate (split-K, C pre-zeroed or accumulating) |
//      5 patch-embed: out row = row + row/group + 1, += extra[(row%group+1)*ldc + col]
// flags bit0: 0 = global_load_lds staging, 1 = register staging (debug/fallback A-B switch)
extern "C" int cs_gemm_nt(const void* A, const void* B, void* C, const float* bias, const float* extra,
                          int M, int N, int K, int lda, int ldb, int ldc, int epi, int splits, int group,
                          int flags, hipStream_t stream) {
    CS_CHECK_ARG(M > 0 && N > 0 && K > 0, "cs_gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
    CS_CHECK_ARG(K % BK == 0, "cs_gemm_nt: K=%d must be a multiple of %d", K, BK);
    CS_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0, "cs_gemm_nt: lda/ldb must be multiples of 8 (16-byte rows)");
    CS_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "cs_gemm_nt: operands must be 16-byte aligned");
    CS_CHECK_ARG(splits >= 1 && (splits == 1 || epi == EPI_ATOMIC_F32), "cs_gemm_nt: split-K needs the atomic epilogue");
    GemmArgs a;
    a.A = (const __bf16*)A; a.B = (const __bf16*)B; a.C = C; a.bias = bias; a.extra = extra;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.group = group;
    a.tiles_m = (M + BM - 1) / BM;
    if (epi == EPI_SWIGLU_BF16) {
        CS_CHECK_ARG(group > 0 && N == 2 * group, "cs_gemm_nt: swiglu epilogue needs N == 2*group");
        a.tiles_n = (group + 63) / 64;
    } else {
        a.tiles_n = (N + BN - 1) / BN;
    }
    if (epi == EPI_PATCH_F32 || epi == EPI_RESID_F32) CS_CHECK_ARG(extra != nullptr, "cs_gemm_nt: epilogue %d needs extra", epi);
    if (epi == EPI_PATCH_F32) CS_CHECK_ARG(group > 0, "cs_gemm_nt: patch epilogue needs group");
    const int ktiles = K / BK;
    a.ktiles_per_split = (ktiles + splits - 1) / splits;
    const int glds = (flags & 1) ? 0 : 1;
    switch (epi) {
        case EPI_BF16: return launch<EPI_BF16>(a, splits, glds, stream);
        case EPI_F32: return launch<EPI_F32>(a, splits, glds, stream);
        case EPI_RESID_F32: return launch<EPI_RESID_F32>(a, splits, glds, stream);
        case EPI_SWIGLU_BF16: return launch<EPI_SWIGLU_BF16>(a, splits, glds, stream);
        case EPI_ATOMIC_F32: return launch<EPI_ATOMIC_F32>(a, splits, glds, stream);
        case EPI_PATCH_F32: return launch<EPI_PATCH_F32>(a, splits, glds, stream);
    }
    cs_set_error("cs_gemm_nt: unknown epilogue %d", epi);
    return -1;
}
