// Shared pieces of the bf16 MFMA GEMM kernels (gemm.hip, gemm_stream.hip): argument block, LDS-DMA staging with the source-side
// bank swizzle, XCD-aware tile rasters.  gfx950 only.
#pragma once
#include "cs_common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

// Timing-ablation switches (cs_gemm_nt flags bits 12-14: skip the operand DMA / the MFMA loop / the epilogue, mask every store) give wrong
// results by construction.  They exist only in builds with -DCS_ABLATION_SWITCHES (CS_EXTRA_FLAGS of build.sh; tools/gemm_bench.py ablation
// columns, tools/barrier_cost.py); the shipped library compiles them out, so no flag value reachable through the C ABI changes a result.
#ifdef CS_ABLATION_SWITCHES
#define CS_ABL(p, bit) ((p).dbg & (bit))
#else
#define CS_ABL(p, bit) false
#endif

constexpr int BK = 64;
enum Epi { EPI_BF16 = 0, EPI_F32 = 1, EPI_RESID_F32 = 2, EPI_SWIGLU_BF16 = 3, EPI_ATOMIC_F32 = 4, EPI_PATCH_F32 = 5, EPI_RESID_LN_F32 = 6,
           EPI_GELU_BF16 = 7, EPI_QGELU_BF16 = 8 };
// bf16 output of acc + bias, optionally through the MLP activation of the OpenAI-CLIP ViT (0 none, 1 exact GELU, 2 QuickGELU)
constexpr bool epi_is_bf16(int e) { return e == EPI_BF16 || e == EPI_GELU_BF16 || e == EPI_QGELU_BF16; }
constexpr int epi_act(int e) { return e == EPI_GELU_BF16 ? 1 : (e == EPI_QGELU_BF16 ? 2 : 0); }

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
    long split_stride;    // EPI_F32 with split-K: slice y writes its partial product to C + y*split_stride (cs_gemm_wgrad)
    int group;            // EPI_PATCH: tokens-1 per image ; EPI_SWIGLU: hidden width Hd
    int gm;               // M panels per raster group
    int rm;               // persistent kernel: raster / cache-policy mode (template parameter RM), host-side selector
    int nsplit;           // B-stationary raster (persistent kernel, RM 1/2): N parts the XCDs are divided over (1, 2, 4 or 8)
    // LayerNorm folded into the GEMM (frozen towers): A holds the *un-normalised* rows, B = gamma (.) W, and the epilogue applies
    // out = extra + rstd[m] * (acc - mean[m] * ln_colsum[n]) + bias[n]   with ln_colsum[n] = sum_k B[n,k], bias[n] = beta.W[n] + b[n].
    const float* ln_mean;
    const float* ln_rstd;
    const float* ln_colsum;
    float* stats_part;    // optional per-(column slice, row) partial (sum, sum of squares) of the outputs: EPI_SWIGLU 32-column slices of
                          // the rounded bf16 values, residual epilogues 64-column slices of the fp32 values
    __bf16* xb_out;       // residual epilogues: optional bf16 copy of the fp32 output (operand of the next LN-folded GEMM)
    int ldxb;
    // "split stream" (cs_gemm_nt_ln_split): the fp32 residual stream kept as two 16-bit planes of the word y = bits(x) + 0x8000 --
    // xb_out holds y >> 16 (= x rounded to bf16, halves away from zero: the next folded GEMM's operand), lo holds y & 0xffff -- so that a
    // residual GEMM moves 8 bytes per element (read 4, write 4) instead of 10 (fp32 in, fp32 out, bf16 copy) and loses nothing.
    // split bit 0: the stream comes in as (xb_out, lo) instead of extra; bit 1: it leaves as (xb_out, lo) instead of C.  Row stride ldxb.
    unsigned short* lo = nullptr;
    int split = 0;
    int reserve;          // persistent kernels: compute units to leave free (grid = 256 - reserve), e.g. for RCCL kernels running beside the step
    int dbg;              // flags bits 12-15: bit 0 = slab form of the streaming kernel's bf16 / SwiGLU epilogues, bit 3 = burst DMA issue (both exact
                          // A/B switches); with -DCS_ABLATION_SWITCHES also the wrong-result timing ablations (CS_ABL)
};

// Compute units of the current device (multiProcessorCount, read once per process): the persistent kernels launch one workgroup per CU.
inline int cs_num_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}
// persistent grid: one workgroup per compute unit, minus the ones the caller keeps free (cs_gemm_nt flags bits 20-27: for RCCL's kernels in
// data-parallel runs, and for the OTHER tower's kernels when the frozen teacher's pass and the student's step share the chip -- the student's
// GEMMs then keep as few as 8 workgroups).  A reserve that would leave fewer than 8 is ignored.
inline long cs_persistent_cap(int reserve) {
    const int n = cs_num_cus();
    return n - (reserve > 0 && reserve <= n - 8 ? reserve : 0);
}

// gemm_stream.hip: streaming persistent kernel with register-level epilogues.  Returns 1 when the problem is outside what it covers
// (the caller falls back to gemm_persist_kernel), 0 on launch, < 0 on error.
int cs_gemm_stream_launch(GemmArgs a, int epi, int reserve, hipStream_t stream);
int cs_gemm_stream_launch_f8(GemmArgs a, int epi, int reserve, hipStream_t stream);

namespace {

// One wave instruction fills 1 KiB = 8 tile rows (row group rg).  Lane l lands at rg*1024 + l*16, i.e. LDS row
// rg*4 + (l>>4), slot l&15; element (tile row r, 16-byte chunk c) lives at LDS row r>>1, slot ((r&1)*8 | c) ^ ((r>>1)&15),
// so the lane must fetch the chunk that this map sends to its slot.
__device__ __forceinline__ void lane_source(int rg, int lane, int& tile_row, int& chunk) {
    const int lrow = rg * 4 + (lane >> 4);
    const int c16 = (lane & 15) ^ (lrow & 15);
    tile_row = lrow * 2 + (c16 >> 3);
    chunk = c16 & 7;
}

// AUX = cache-policy bits of the DMA load (gfx950: 1 = sc0, 2 = nt "streaming, evict first", 16 = sc1)
template <int NINSTR, bool GLDS, int AUX = 0>
__device__ __forceinline__ void stage_tile(const __bf16* __restrict__ src, int ld, int k0, char* lds_tile, int rg0, int lane,
                                           const int (&grow)[NINSTR], const int (&gchunk)[NINSTR]) {
#pragma unroll
    for (int i = 0; i < NINSTR; ++i) {
        const __bf16* g = src + (size_t)grow[i] * ld + k0 + gchunk[i] * 8;
        char* dst = lds_tile + (rg0 + i) * 1024;          // wave-uniform
        if (GLDS) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, AUX);
        } else {
            *(uint4*)(dst + lane * 16) = *(const uint4*)g;
        }
    }
}

// XCD-aware, grouped-raster tile assignment (block b runs on XCD b % 8; bijective for any grid size)
__device__ __forceinline__ void tile_of_id(const GemmArgs& p, int bid, int nwg, int& tm, int& tn);
__device__ __forceinline__ void tile_of_block(const GemmArgs& p, int& tm, int& tn) { tile_of_id(p, blockIdx.x, gridDim.x, tm, tn); }
__device__ __forceinline__ void tile_of_id(const GemmArgs& p, int bid, int nwg, int& tm, int& tn) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int per_group = p.gm * p.tiles_n;
    const int grp = swz / per_group, rem = swz - grp * per_group;
    const int rows = min(p.gm, p.tiles_m - grp * p.gm);
    tn = rem / rows;
    tm = grp * p.gm + (rem - tn * rows);
}

// B-stationary raster of the persistent kernel: XCD x owns N part x % NP and M part x / NP (NP = p.nsplit, 8 / NP M parts) and walks
// its tiles N-fastest, so the CUs of an XCD keep re-reading the same <= tiles_n / NP weight panels (they stay in that XCD's 4 MiB L2
// for the whole launch) while the activation panels stream through once per N part.  lt = XCD-local tile index; false past the end.
__device__ __forceinline__ bool tile_bstat(const GemmArgs& p, int xcd, int lt, int& tm, int& tn) {
    const int NP = p.nsplit, MP = 8 / NP;
    const int np = xcd % NP, mp = xcd / NP;
    const int n_lo = np * p.tiles_n / NP, nn = (np + 1) * p.tiles_n / NP - n_lo;
    const int m_lo = mp * p.tiles_m / MP, mm = (mp + 1) * p.tiles_m / MP - m_lo;
    if (lt >= mm * nn) return false;
    const int q = lt / nn;
    tm = m_lo + q;
    tn = n_lo + (lt - q * nn);
    return true;
}

template <int EPI, int BM, int BN, int NW, int A_INSTR, int B_INSTR>
__device__ __forceinline__ void source_rows(const GemmArgs& p, int wave, int lane, int m0, int n0, int tn, int (&arow)[A_INSTR],
                                            int (&achk)[A_INSTR], int (&brow)[B_INSTR], int (&bchk)[B_INSTR]) {
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        int tr;
        lane_source(wave * A_INSTR + i, lane, tr, achk[i]);
        arow[i] = min(m0 + tr, p.M - 1);
    }
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        int tr;
        lane_source(wave * B_INSTR + i, lane, tr, bchk[i]);
        if (EPI == EPI_SWIGLU_BF16) {
            // tile rows [w*64 + jj*32 + t] <- weight row jj*Hd + (tn*(BN/2) + w*32 + t): x1 and x2 of one hidden unit land in
            // the same lane/register of accumulator column-tiles j=0 / j=1 of wave column w.
            const int hidx = tn * (BN / 2) + (tr >> 6) * 32 + (tr & 31);
            brow[i] = ((tr >> 5) & 1) * p.group + min(hidx, p.group - 1);
        } else {
            brow[i] = min(n0 + tr, p.N - 1);
        }
    }
}

}  // namespace
