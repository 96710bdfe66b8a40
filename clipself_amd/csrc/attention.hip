// Multi-head self-attention (head dim 64, no mask, no dropout) with the EVA02 2-D RoPE applied on load, for the
// CLIPSelf teacher/student towers -- forward and the student's backward.  gfx950, MFMA 32x32x16 bf16.
//
// Reference semantics (src/open_clip/eva_clip/eva_vit_model.py:181-243, rope.py:25-29,148-164):
//   q,k,v = split heads of [B*N, 3C];  q,k tokens 1.. rotated:  y = t*cos + rotate_half(t)*sin  (CLS token passed through)
//   o = softmax(q k^T * d^-0.5) v,  written back token-major [B*N, C].
//
// Layout/algorithm (MI355X-first):
//   * one 512-thread workgroup (8 waves) per (image, head, 256-query group); each wave owns 32 queries.
//   * "swapped" products: S^T[key,query] = K . Q^T, so every lane owns ONE query column and the softmax row
//     reductions are in-register (+ one cross-half shuffle); P^T accumulator registers feed the next MFMA as the
//     B operand with no cross-lane movement because the contraction slots are *defined* by the accumulator layout
//     (slot (half,j) <-> key (j&3) + 8*(j>>2) + 4*half) and the A operand (V^T / K^T rows) is read in that order.
//   * K (roped) is staged once per workgroup into LDS as [key][64] with a 16-byte-chunk XOR swizzle
//     (conflict-free ds_read_b128), V as V^T [64][keys+4] (8-byte reads, odd dword stride -> conflict free).
//   * keys are consumed in chunks of CH*32 with an online softmax, so any sequence length works; for the
//     14x14(+CLS) grid a single chunk of 224 covers all 197 keys and no rescale is ever taken.
#include <type_traits>
#include <vector>
#include "cs_common.h"
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int HD = 64;
constexpr float LOG2E = 1.4426950408889634f;

// (Round 6 measured hand-packed fp32 math -- v_pk_fma_f32 / v_pk_add_f32 on float2 values for the softmax's scale-and-subtract, the row sums
// and the backward's exp2(fma) * fma -- against hipcc's own choice: 4-25 % SLOWER in the forward kernels, a tie in the backward; the packed
// forms need aligned register pairs, i.e. more VGPRs and copies, and return no issue slots here.  profiles/r06_d_attn_ab.txt.)

#define Z16 (f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f})
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    return z;
}

// rotate 8 consecutive head-dim values (4 interleaved pairs) by the table rows at cs/sn (fp32, 8 entries each)
__device__ __forceinline__ void rope8(U128& v, const float* __restrict__ cs, const float* __restrict__ sn) {
    const float4 c0 = *(const float4*)cs, c1 = *(const float4*)(cs + 4);
    const float4 s0 = *(const float4*)sn, s1 = *(const float4*)(sn + 4);
    const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float s[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    // The contraction is written out (one product rounded to fp32, then one FMA): left to hipcc's -ffp-contract=fast, WHICH of the two products is
    // fused depends on the surrounding code, and the same source then rounds differently in two kernels (round 6: one dQ element in 4 M differed
    // between the prepass kernel's rotation and the staging loop's).
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x0 = bf2f(v.e[2 * j]), x1 = bf2f(v.e[2 * j + 1]);
        v.e[2 * j] = f2bf(__builtin_fmaf(x0, c[2 * j], -(x1 * s[2 * j])));
        v.e[2 * j + 1] = f2bf(__builtin_fmaf(x1, c[2 * j + 1], x0 * s[2 * j + 1]));
    }
}

// The reference builds its tables separably (rope.py:118-142: row-angle part for dims [0,32), column-angle part for [32,64), the
// same for every token of a grid row / column), so a workgroup needs only 4 x [g][32] floats of the [g*g][64] tables.  rt holds
// cos_row | sin_row | cos_col | sin_col; dims chunk c8 (8 dims) of token tok >= 1 is rotated from LDS instead of re-reading
// 64 bytes of table per 16 bytes of q/k from L2 (that re-read was two thirds of the forward kernel's vector-memory traffic).
__device__ __forceinline__ void rope8_lds(U128& v, const float* rt, int g, float inv_g, int tok, int c8) {
    const int t = tok - 1;
    const int r = (int)(((float)t + 0.5f) * inv_g), c = t - r * g;
    const float* cs = rt + ((c8 < 4 ? r : 2 * g + c) << 5) + (c8 & 3) * 8;
    rope8(v, cs, cs + (g << 5));
}

// Round 6: the tables ONCE per frequency.  rope.py:118-142 builds the row part and the column part from the same `freqs` tensor and repeats
// every frequency for the two dims of a pair: cos_row[r][2j] == cos_row[r][2j+1] == cos_col[r][2j].  rt = cos [g][16] | sin [g][16] (a quarter of
// the [4][g][32] tables above: 8 KB instead of 32 KB at the recipe's 64 x 64 grid); same products, same bits as rope8_lds.  A documented
// precondition of the C ABI (include/clipself_hip.h), verified by HipOps once per table tensor.
__device__ __forceinline__ void rope8c(U128& v, const float* rt, int g, float inv_g, int tok, int c8) {
    const int t = tok - 1;
    const int r = (int)(((float)t + 0.5f) * inv_g), c = t - r * g;
    const float* cs = rt + ((c8 < 4 ? r : c) << 4) + (c8 & 3) * 4;
    const float4 c4 = *(const float4*)cs, s4 = *(const float4*)(cs + (g << 4));
    const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {              // the written-out contraction of rope8
        const float x0 = bf2f(v.e[2 * j]), x1 = bf2f(v.e[2 * j + 1]);
        v.e[2 * j] = f2bf(__builtin_fmaf(x0, cc[j], -(x1 * ss[j])));
        v.e[2 * j + 1] = f2bf(__builtin_fmaf(x1, cc[j], x0 * ss[j]));
    }
}

template <int NT>
__device__ __forceinline__ void load_rope_tables_c(float* rt, const float* __restrict__ cos_t, const float* __restrict__ sin_t, int g, int tid) {
    for (int i = tid; i < g * 16; i += NT) {               // the column part of grid row 0: token i >> 4, dims 32 + 2 (i & 15)
        const int r = i >> 4, j = i & 15;
        rt[i] = cos_t[(size_t)r * HD + 32 + 2 * j];
        rt[(g << 4) + i] = sin_t[(size_t)r * HD + 32 + 2 * j];
    }
}

template <int NT>
__device__ __forceinline__ void load_rope_tables(float* rt, const float* __restrict__ cos_t, const float* __restrict__ sin_t, int g, int tid) {
    for (int i = tid; i < g * 32; i += NT) {
        const int r = i >> 5, d = i & 31;
        rt[i] = cos_t[(size_t)(r * g) * HD + d];
        rt[(g << 5) + i] = sin_t[(size_t)(r * g) * HD + d];
        rt[(2 * g << 5) + i] = cos_t[(size_t)r * HD + 32 + d];
        rt[(3 * g << 5) + i] = sin_t[(size_t)r * HD + 32 + d];
    }
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int c2) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = f2bf(a[c2 * 8 + j]);
    return r;
}

// A operand rows from a transposed [64][ld] bf16 image: slots (half,j) <-> column base + (j&3) + 8*(j>>2) + 4*half
__device__ __forceinline__ bf16x8 load_t_frag(const __bf16* t, int ld, int d, int base, int hf) {
    U64 lo, hi;
    lo.u = *(const uint2*)(t + d * ld + base + 4 * hf);
    hi.u = *(const uint2*)(t + d * ld + base + 8 + 4 * hf);
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = lo.e[j]; r[4 + j] = hi.e[j]; }
    return r;
}

#ifdef CS_ABLATION_SWITCHES
#define ATT_ABL(p, bit) ((p).dbg & (bit))
#else
#define ATT_ABL(p, bit) false
#endif
struct AttnArgs {
    const __bf16* qkv;     // [B*N, ldqkv]: q | k | v, each C = H*64 wide
    const __bf16* dout;    // bwd: dO [B*N, ldo]
    const float* cos_t;    // [N-1, 64]
    const float* sin_t;
    const float* lse_in;   // bwd: [B*H, N]
    const float* dsum;     // bwd: rowsum(dO*O) [B*H, N]
    int dbg;               // timing ablations (results wrong; env CS_ATTN_DBG, read only in builds with -DCS_ABLATION_SWITCHES): 1 = no MFMA/softmax
                           // phase, 2 = no RoPE, 4 = no output stores
    int grid;              // fwd: token grid side g (Ntok = g*g + 1)
    float inv_grid;
    float* stats_part;     // fwd, optional: [H][B*N][2] per-head (sum, sum of squares) of the output rows
    long Mtot;             // B*N
    __bf16* out;           // fwd: O [B*N, ldo] ; bwd: dqkv [B*N, ldqkv]
    float* lse_out;        // fwd (nullable)
    int Ntok, H, ldqkv, ldo;
    float scale;
    // round 6, block schedule of the long-sequence kernels (set_schedule / map_block below); sch_on = 0: blockIdx.y = (image, head), blockIdx.x = row block
    int sch_on, sch_f0, sch_r, sch_bh, sch_fb, sch_rem, sch_absorb;
    const __bf16* qk;      // PRE kernels: rotated q | k, [B*N, ldqk] (rope_qk_kernel)
    int ldqk;
#ifdef CS_ABLATION_SWITCHES
    unsigned long long* trace;   // env CS_ATTN_TRACE=<file>: per workgroup 8 x u64 (HW_ID, XCC_ID, 100 MHz clock at entry / tables / images / attended / end)
#endif
};
#ifdef CS_ABLATION_SWITCHES
#define ATT_TRACE(slot) do { if (p.trace && threadIdx.x == 0) p.trace[(size_t)blockIdx.y * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ATT_TRACE(slot) do { } while (0)
#endif

// Round 6: the tail of a launch is split.  The long-sequence kernels run ONE 8-wave workgroup per CU (216-247 VGPRs), a workgroup = 8 row tiles of 32
// (256 queries resp. keys) of one (image, head) against the whole sequence.  At the recipe's 2 x 12 x 4097 that is 16 x 24 = 384 equal workgroups
// on 256 CUs: two rounds, the second half empty (+ 24 workgroups for token 4096 alone, each walking all keys with one active wave).  With F full
// blocks, G CUs and R = F mod G blocks left for the last round, 2 R <= G: those R blocks are issued as 2 R half blocks of 4 tiles -- the last round
// takes ~0.6 of a full one instead of 1 -- and a sequence's last tiles (<= 4) ride on the upper half of its last full block (a fifth ... eighth wave)
// instead of walking the sequence on their own.  Ids are block-major over the (image, head) pairs, so the halves are the last blocks of every
// pair and with B x H a multiple of 8 a pair's blocks land on one XCD, whose L2 then holds its K / V.  Every row is computed by the same code
// on the same operands in the same order: bit-identical results (tests/test_gpu_ops.py).
struct RowBlock { int bh, tile0, nt; };
__device__ __forceinline__ RowBlock map_block(const AttnArgs& p, int waves) {
    RowBlock r;
    if (!p.sch_on) {
        r.bh = blockIdx.y; r.tile0 = blockIdx.x * waves; r.nt = waves;
        return r;
    }
    const int id = blockIdx.x;
    if (id < p.sch_f0) {
        const int qb = id / p.sch_bh;
        r.bh = id - qb * p.sch_bh; r.tile0 = qb * 8; r.nt = 8;
    } else if (id < p.sch_f0 + 2 * p.sch_r) {
        const int j = id - p.sch_f0, k = p.sch_f0 + (j >> 1), half = j & 1, qb = k / p.sch_bh;
        r.bh = k - qb * p.sch_bh; r.tile0 = qb * 8 + half * 4;
        r.nt = 4 + ((p.sch_absorb && half && qb == p.sch_fb - 1) ? p.sch_rem : 0);
    } else {
        r.bh = id - p.sch_f0 - 2 * p.sch_r; r.tile0 = p.sch_fb * 8; r.nt = p.sch_rem;
    }
    return r;
}

// stage `rows` token rows (tok0..) of one head column block into a swizzled [rows][64] LDS tile (+ optional transposed copy)
template <int CHK, bool ROPE, bool WITH_T>
__device__ __forceinline__ void stage_rows(const __bf16* __restrict__ src, size_t rowbase, int ld, int coloff, int tok0, int Ntok,
                                           const float* cos_t, const float* sin_t, char* tile, __bf16* tile_t, int ldt, int tid) {
    for (int idx = tid; idx < CHK * 8; idx += 512) {
        const int r = idx >> 3, c = idx & 7, tok = tok0 + r;
        U128 v;
        if (tok < Ntok) {
            v.u = *(const uint4*)(src + (rowbase + tok) * ld + coloff + c * 8);
            if (ROPE && tok > 0) rope8(v, cos_t + (size_t)(tok - 1) * HD + c * 8, sin_t + (size_t)(tok - 1) * HD + c * 8);
        } else {
            v.u = make_uint4(0, 0, 0, 0);
        }
        if (tile) *(uint4*)(tile + r * 128 + ((c ^ (r & 7)) << 4)) = v.u;
        if (WITH_T) {
#pragma unroll
            for (int e = 0; e < 8; ++e) tile_t[(c * 8 + e) * ldt + r] = v.e[e];
        }
    }
}

__device__ __forceinline__ bf16x8 lds_frag(const char* tile, int row, int ks, int hf) {
    return *(const bf16x8*)(tile + row * 128 + ((((ks << 1) + hf) ^ (row & 7)) << 4));
}

// ------------------------------------------------------------------------------------------------ forward
// LDS images of the forward kernel:
//   K  : [keys][64] bf16, two 128-byte key rows per 256-byte LDS row, sixteen 16-byte slots XOR-permuted by
//        (lds_row & 15)  -> conflict-free ds_write_b128 (staging) and ds_read_b128 (MFMA A fragments)
//   V^T: [64 d][VT_LD keys] bf16 built by an in-register 8x8 transpose (8 keys x 8 dims per thread, eight
//        ds_write_b128); key blocks of 8 are XOR-permuted by (d>>3)&7 so the 8 writers of one key block hit 8 slots.
constexpr int VT_LD = 264;            // 33 blocks of 8 keys: odd block stride -> ds_read_b128 rows spread over all slots

__device__ __forceinline__ int k_off(int r, int c) { return ((r >> 1) << 8) + (((((r & 1) << 3) | c) ^ ((r >> 1) & 15)) << 4); }

template <int CHK, int NT>
__device__ __forceinline__ void stage_k(const __bf16* __restrict__ src, size_t rowbase, int ld, int coloff, int tok0, int Ntok,
                                        const float* rt, int g, float inv_g, char* tile, int tid) {
    constexpr int ITEMS = (CHK * 8 + NT - 1) / NT;
    U128 v[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {                      // all loads in flight before the first use
        const int idx = min(tid + it * NT, CHK * 8 - 1), tok = min(tok0 + (idx >> 3), Ntok - 1);   // branch-free: padding rows repeat the last row
        v[it].u = *(const uint4*)(src + (rowbase + tok) * ld + coloff + (idx & 7) * 8);
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int idx = tid + it * NT, r = idx >> 3, c = idx & 7, tok = tok0 + r;
        if (idx < CHK * 8) {
            if (tok > 0 && tok < Ntok) rope8_lds(v[it], rt, g, inv_g, tok, c);
            *(uint4*)(tile + k_off(r, c)) = v[it].u;
        }
    }
}

template <int CHK, int NT>
__device__ __forceinline__ void stage_vt(const __bf16* __restrict__ src, size_t rowbase, int ld, int coloff, int tok0, int Ntok,
                                         __bf16* vt, int tid) {
    for (int idx = tid; idx < CHK; idx += NT) {              // CHK/8 key blocks x 8 dim chunks
        const int kb = idx >> 3, c = idx & 7;
        U128 in[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int tok = min(tok0 + kb * 8 + i, Ntok - 1);        // branch-free (see attn_fwd_kernel): padding keys carry p = 0
            in[i].u = *(const uint4*)(src + (rowbase + tok) * ld + coloff + c * 8);
        }
        const int pos = (kb ^ c) * 8;                        // (d>>3)&7 == c for d = c*8 + j
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            U128 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.e[i] = in[i].e[j];
            *(uint4*)(vt + (c * 8 + j) * VT_LD + pos) = o.u;
        }
    }
}

// ---- round 5: row-major images + transposing reads (attention backward, long-sequence forward) -------------------------------------
typedef short s16x4v __attribute__((ext_vector_type(4)));

// lane-constant part of a transposing fragment read from a row-major [rows][64] image in the k_off layout: the lane's column is
// d0 + (lane & 31); its 16-lane group addresses rows rr .. rr + 3 (rr = first row of the 4-row block, relative to a 32-row tile).  The
// tile index only adds t * 4096: (r >> 1) & 15 does not see multiples of 32 rows.
__device__ __forceinline__ int tr_lane_off(int rr, int d0, int lane) {
    const int li = lane & 15, g1 = (lane >> 4) & 1;
    const int col = d0 + 16 * g1 + 4 * (li & 3);
    return k_off(rr + (li >> 2), col >> 3) + ((col >> 2) & 1) * 8;
}
__device__ __forceinline__ bf16x8 tr_join2(s16x4v lo, s16x4v hi) {
    typedef short s16x8v __attribute__((ext_vector_type(8)));
    const s16x8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ s16x4v tr_read4(const char* addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)addr);
}

// rows tok0 .. tok0 + CHK - 1 of one head's 64 columns into registers (rows past the sequence repeat its last row: finite values that
// only meet p = 0), and from registers into a row-major LDS image, rotated on the way (ROPE) from the LDS tables
template <int CHK, int NT = 512>
struct RowRegs {
    static constexpr int ITEMS = (CHK * 8 + NT - 1) / NT;
    uint4 v[ITEMS];                 // plain vectors: a union that lives across the chunk loop is kept on the stack
    __device__ __forceinline__ void load(const __bf16* __restrict__ src, size_t rowbase, int ld, int coloff, int tok0, int Ntok, int tid) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int idx = min(tid + it * NT, CHK * 8 - 1), tok = min(tok0 + (idx >> 3), Ntok - 1);
            v[it] = *(const uint4*)(src + (rowbase + tok) * ld + coloff + (idx & 7) * 8);
        }
    }
    // ZPAD: rows past the sequence are stored as zeros (the dQ kernel: a zero K row contributes nothing to dQ^T += K^T . dS^T whatever its
    // dS column holds, so the hot loop masks nothing)
    // CT: rt holds the compact tables (one entry per frequency, rope8c)
    template <bool ROPE, bool ZPAD = false, bool CT = false>
    __device__ __forceinline__ void store(char* tile, const float* rt, int g, float inv_g, int tok0, int Ntok, int tid) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int idx = tid + it * NT, r = idx >> 3, c = idx & 7, tok = tok0 + r;
            if (idx < CHK * 8) {
                U128 t;
                t.u = v[it];
                if (ROPE && tok > 0 && tok < Ntok) {
                    if constexpr (CT) rope8c(t, rt, g, inv_g, tok, c);
                    else rope8_lds(t, rt, g, inv_g, tok, c);
                }
                if (ZPAD && tok >= Ntok) t.u = make_uint4(0, 0, 0, 0);
                *(uint4*)(tile + k_off(r, c)) = t.u;
            }
        }
    }
};

// P^T accumulator registers [8*c2, 8*c2+8) of both wave halves -> B fragment in the conventional slot order
// (half h supplies keys 8h..8h+7 of the 16-key step): pack to bf16 pairs, then exchange half 0's keys 8-11 with
// half 1's keys 4-7 (v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second).
__device__ __forceinline__ bf16x8 pack8_swapped(const f32x16& a, int c2) {
    union { bf16x2 h; unsigned u; } x0, x1, y0, y1;
    x0.h[0] = f2bf(a[c2 * 8 + 0]); x0.h[1] = f2bf(a[c2 * 8 + 1]);
    x1.h[0] = f2bf(a[c2 * 8 + 2]); x1.h[1] = f2bf(a[c2 * 8 + 3]);
    y0.h[0] = f2bf(a[c2 * 8 + 4]); y0.h[1] = f2bf(a[c2 * 8 + 5]);
    y1.h[0] = f2bf(a[c2 * 8 + 6]); y1.h[1] = f2bf(a[c2 * 8 + 7]);
    const auto r0 = __builtin_amdgcn_permlane32_swap(x0.u, y0.u, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(x1.u, y1.u, false, false);
    union { unsigned u[4]; bf16x8 v; } out;
    out.u[0] = r0[0]; out.u[1] = r1[0]; out.u[2] = r0[1]; out.u[3] = r1[1];
    return out.v;
}

template <bool CT = false>
__device__ __forceinline__ void load_q_frags(const AttnArgs& p, const float* rt, size_t rowbase, int qc, int h, int hf, bf16x8 (&qf)[4]) {
    U128 t[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) t[ks].u = *(const uint4*)(p.qkv + (rowbase + qc) * p.ldqkv + h * HD + ks * 16 + hf * 8);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (qc > 0) {
            if constexpr (CT) rope8c(t[ks], rt, p.grid, p.inv_grid, qc, ks * 2 + hf);
            else rope8_lds(t[ks], rt, p.grid, p.inv_grid, qc, ks * 2 + hf);
        }
        qf[ks] = t[ks].h;
    }
}

// one key chunk against one 32-query tile: S^T = K Q^T, online-softmax update of (m, l), O^T += V^T P^T
// t0: first key tile of the chunk inside the LDS images (0 when the images hold only this chunk; the V^T key-block swizzle is a function of
// the absolute block index, so a chunk of a whole-sequence image cannot be addressed through an offset pointer)
// VROW: Vt points at a ROW-MAJOR [keys][64] image in the k_off layout instead of the transposed one; the V^T fragments are then read with
// ds_read_b64_tr_b16 (half hf supplies keys 8 hf .. 8 hf + 7 of the 16-key step, the conventional order pack8_swapped produces).
// VM, the V image (round 6):
//   0  V^T [64][264], key blocks XOR-permuted (rounds 1-5).  Under ds_read_b128's lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31} per wave
//      half: MI355X_MICROARCH.md "LDS") its fragment reads are 2-WAY bank-conflicted -- 8 instead of 4 LDS cycles each, and V^T reads are two
//      thirds of the attend phase's LDS reads (SQ_LDS_BANK_CONFLICT = 33 % of SQ_LDS_IDX_ACTIVE in profiles/r05_e_pmc_attention_lds.md)
//   1  row-major [keys][64] in the k_off layout, fragments through ds_read_b64_tr_b16 (VROW)
//   3  V^T [64][200] for sequences of <= 200 keys (attn_fwd4_kernel): key block kb of dim row d sits at block (kb + (d >> 5)) mod 25 -- a ROTATION
//      instead of the XOR: conflict-free fragment reads, 4-way conflicts on the 8 transposing writes per thread and unit (tools/lds_bank_model.py)
//   (2 was a [64][224] image rotated by the dim chunk, (kb + (d >> 3 & 7)) mod 28: conflict-free reads AND writes in attn_fwd8_kernel -- built,
//   bit-identical, `SQ_LDS_BANK_CONFLICT` 19.9 M -> 12 M per launch, and 1 % SLOWER (425-427 vs 419-422 us per 1024-crop launch: the wrap of the
//   rotation costs a compare + subtract per fragment address and the LDS array was 34 % busy to begin with); removed.  profiles/r06_c_attention_pipes.md)
// The same values reach the same MFMAs in every mode: bit-identical outputs.
constexpr int VT4_LD = 200;

template <int CH, bool TAIL, int VM = 0>
__device__ __forceinline__ void attend_chunk(const char* Kl, const __bf16* Vt, const bf16x8 (&qf)[4], int key0, int Ntok, float sl2,
                                             int lane, bool first, float& m, float& l, f32x16 (&o)[2], int t0 = 0) {
    constexpr bool VROW = VM == 1;
    const int hf = lane >> 5, l31 = lane & 31;
    // fragment addressing (row = t*32 + l31): LDS row (row>>1), slot ((row&1)*8 | chunk) ^ ((row>>1)&15)
    const int k_base = (l31 >> 1) << 8, par8 = (l31 & 1) << 3, sw = l31 >> 1;
    // Key tile t holds keys key0 + 32t ..; only the last tile of a sequence is ragged.  TAIL: the caller guarantees that tiles 0..CH-2
    // are full (single-chunk launches with Ntok > 32(CH-1)), so the ragged-tile code exists once.  For the 14x14(+CLS) grid the last tile
    // holds 5 keys: rows 0..7 of a 32x32 accumulator tile are registers 0..3 of the two wave halves, so a tile with <= 8 keys needs 4 of
    // its 16 registers in the softmax (the other 12 are p = 0 by construction) and one of its two P.V steps.
    f32x16 s[CH];
#pragma unroll
    for (int t = 0; t < CH; ++t) {
        s[t] = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 kfrag = *(const bf16x8*)(Kl + (t0 + t) * (16 * 256) + k_base + (((par8 | (ks * 2 + hf)) ^ sw) << 4));
            s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag, qf[ks], s[t], 0, 0, 0);
        }
    }
    const int rem_last = Ntok - key0 - (CH - 1) * 32;          // keys in the last tile (TAIL: 1..32)
    const bool short_tail = TAIL && rem_last <= 8;
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < CH; ++t) {
        if (TAIL && t == CH - 1 && short_tail) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e + 4 * hf >= rem_last) s[t][e] = -INFINITY;
                mx = fmaxf(mx, s[t][e]);
            }
            continue;
        }
        if ((!TAIL || t == CH - 1) && key0 + t * 32 + 32 > Ntok) {   // wave-uniform: only a ragged key tile needs masking
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (key0 + t * 32 + mfma32_row(e, lane) >= Ntok) s[t][e] = -INFINITY;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[t][e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);
    const float alpha = exp2f((m - m_new) * sl2);
    const float msc = m_new * sl2;
    float rs = 0.f;
#pragma unroll
    for (int t = 0; t < CH; ++t) {
        if (TAIL && t == CH - 1 && short_tail) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = __builtin_amdgcn_exp2f(s[t][e] * sl2 - msc);
                s[t][e] = pv;
                rs += pv;
            }
#pragma unroll
            for (int e = 4; e < 8; ++e) s[t][e] = 0.f;          // registers 4..7 complete the one P.V step taken below
            continue;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float pv = __builtin_amdgcn_exp2f(s[t][e] * sl2 - msc);     // raw v_exp_f32: argument <= 0, result in [0,1]
            s[t][e] = pv;
            rs += pv;
        }
    }
    rs += __shfl_xor(rs, 32, 64);
    l = l * alpha + rs;
    m = m_new;
    if (!first) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { o[0][e] *= alpha; o[1][e] *= alpha; }
    }
#pragma unroll
    for (int t = 0; t < CH; ++t)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            if (TAIL && t == CH - 1 && c2 == 1 && rem_last <= 16) continue;     // all 16 keys of this step are padding (p = 0)
            const bf16x8 pb = pack8_swapped(s[t], c2);
            const int kb = (t0 + t) * 4 + c2 * 2 + hf;       // key block (8 keys) this half supplies
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                bf16x8 vfrag;
                if constexpr (VROW) {
                    const char* Vl = (const char*)Vt + (t0 + t) * (16 * 256);
                    vfrag = tr_join2(tr_read4(Vl + tr_lane_off(c2 * 16 + 8 * hf, dt * 32, lane)), tr_read4(Vl + tr_lane_off(c2 * 16 + 8 * hf + 4, dt * 32, lane)));
                } else if constexpr (VM == 3) {
                    const int d = dt * 32 + l31;
                    int pos = kb + dt;
                    if ((t0 + t) * 4 + c2 * 2 + 2 >= 25) pos = pos >= 25 ? pos - 25 : pos;        // (compile-time condition: the last key tiles) blocks 25..27 = keys >= 200, p = 0: they wrap onto real, finite data
                    vfrag = *(const bf16x8*)(Vt + d * VT4_LD + (pos << 3));
                } else {
                    const int d = dt * 32 + l31;
                    vfrag = *(const bf16x8*)(Vt + d * VT_LD + ((kb ^ ((d >> 3) & 7)) << 3));
                }
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag, pb, o[dt], 0, 0, 0);
            }
        }
}

__device__ __forceinline__ void store_o(const AttnArgs& p, size_t rowbase, int q, int h, int bh, int hf, float m, float l, const f32x16 (&o)[2]) {
    const float inv = 1.f / l;
    __bf16* orow = p.out + (rowbase + q) * p.ldo + h * HD;
    float ps = 0.f, pq = 0.f;
    // A lane holds 4 consecutive head dims (8 bytes) per register group, its partner in the other wave half the next 4: one
    // v_permlane32_swap per dword pairs them up, so that each lane stores 16 contiguous bytes (groups 2a | 2a+1 go to half 0 | 1) -- half
    // the store instructions and write requests of the 8-byte form.
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        U64 t[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                t[g4].e[i] = f2bf(o[dt][g4 * 4 + i] * inv);
                const float r = bf2f(t[g4].e[i]);
                ps += r;
                pq += r * r;
            }
#pragma unroll
        for (int a = 0; a < 4; a += 2) {
            const auto r0 = __builtin_amdgcn_permlane32_swap(t[a].u.x, t[a + 1].u.x, false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(t[a].u.y, t[a + 1].u.y, false, false);
            if (!ATT_ABL(p, 4)) *(uint4*)(orow + dt * 32 + (a + hf) * 8) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
    }
    if (p.lse_out && hf == 0) p.lse_out[(size_t)bh * p.Ntok + q] = m * p.scale + logf(l);
    if (p.stats_part) {
        // (sum, sum of squares) of this row's 64 rounded outputs of head h: the two half-wave lanes of a query hold 32 each.
        // cs_ln_stats_finalize() pools the H heads, so inner_attn_ln needs no pass of its own over the attention output.
        ps += __shfl_xor(ps, 32);
        pq += __shfl_xor(pq, 32);
        if (hf == 0) *(float2*)(p.stats_part + ((size_t)h * p.Mtot + rowbase + q) * 2) = make_float2(ps, pq);
    }
}

// Sequences longer than one LDS image (Ntok > 224): eight waves, one 32-query tile per wave, 256 queries per workgroup; the keys are staged
// chunk by chunk (CH*32 keys) and consumed with the online softmax.
template <int CH>
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel(AttnArgs p) {
    constexpr int CHK = CH * 32, NT = 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kl = smem;
    __bf16* Vt = (__bf16*)(smem + CHK * 128);
    float* rt = (float*)(smem + CHK * 128 + HD * VT_LD * 2);      // compact RoPE tables [4][g][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int C = p.H * HD;
    const size_t rowbase = (size_t)b * p.Ntok;
    const float sl2 = p.scale * LOG2E;
    const int q0 = blockIdx.x * 256 + wave * 32, q = q0 + l31, qc = min(q, p.Ntok - 1);
    const bool active = q0 < p.Ntok;
    load_rope_tables<NT>(rt, p.cos_t, p.sin_t, p.grid, tid);
    __syncthreads();
    bf16x8 qf[4];
    load_q_frags(p, rt, rowbase, qc, h, hf, qf);
    float m = -INFINITY, l = 0.f;
    f32x16 o[2] = {zero16(), zero16()};
    for (int key0 = 0; key0 < p.Ntok; key0 += CHK) {
        __syncthreads();
        stage_k<CHK, NT>(p.qkv, rowbase, p.ldqkv, C + h * HD, key0, p.Ntok, rt, p.grid, p.inv_grid, Kl, tid);
        stage_vt<CHK, NT>(p.qkv, rowbase, p.ldqkv, 2 * C + h * HD, key0, p.Ntok, Vt, tid);
        __syncthreads();
        if (active) attend_chunk<CH, false>(Kl, Vt, qf, key0, p.Ntok, sl2, lane, key0 == 0, m, l, o);
    }
    if (active && q < p.Ntok) store_o(p, rowbase, q, h, bh, hf, m, l, o);
}

// Round 5, sequences longer than one LDS image.  attn_fwd_kernel above stages a chunk (K rotated, V transposed in registers: 8 rows per
// thread on 224 of the 512 threads) and only then attends it -- two barriers and a full memory round trip per chunk with nothing
// overlapping them: 234 us per block at the recipe's 4097 tokens (0.18 of the MFMA peak).  Here the NEXT chunk's K and V rows are
// requested before the current chunk is attended (register prefetch, 4 + 4 x 16 bytes per thread), V stays row-major in LDS and its
// transposed fragments come from ds_read_b64_tr_b16, and a staged chunk of CH tiles is attended as two online-softmax steps of CH1 and
// CH - CH1 tiles so that the score registers (16 per tile) leave room for the prefetch.
template <int CH, int CH1>
__global__ __launch_bounds__(512, 2) void attn_fwd2_kernel(AttnArgs p) {
    constexpr int CHK = CH * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kl = smem;
    char* Vl = smem + CHK * 128;
    float* rt = (float*)(smem + 2 * CHK * 128);          // compact RoPE tables [4][g][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l31 = lane & 31;
    const RowBlock rb = map_block(p, 8);
    const int bh = rb.bh, b = bh / p.H, h = bh - b * p.H;
    const int C = p.H * HD;
    const size_t rowbase = (size_t)b * p.Ntok;
    const float sl2 = p.scale * LOG2E;
    const int q0 = (rb.tile0 + wave) * 32, q = q0 + l31, qc = min(q, p.Ntok - 1);
    const bool active = wave < rb.nt && q0 < p.Ntok;
    RowRegs<CHK> kr, vr;
    kr.load(p.qkv, rowbase, p.ldqkv, C + h * HD, 0, p.Ntok, tid);
    vr.load(p.qkv, rowbase, p.ldqkv, 2 * C + h * HD, 0, p.Ntok, tid);
    load_rope_tables<512>(rt, p.cos_t, p.sin_t, p.grid, tid);
    __syncthreads();
    bf16x8 qf[4];
    load_q_frags(p, rt, rowbase, qc, h, hf, qf);
    float m = -INFINITY, l = 0.f;
    f32x16 o[2] = {zero16(), zero16()};
    for (int key0 = 0; key0 < p.Ntok; key0 += CHK) {
        if (key0) __syncthreads();
        kr.template store<true>(Kl, rt, p.grid, p.inv_grid, key0, p.Ntok, tid);
        vr.template store<false>(Vl, rt, p.grid, p.inv_grid, key0, p.Ntok, tid);
        __syncthreads();
        if (key0 + CHK < p.Ntok) {
            kr.load(p.qkv, rowbase, p.ldqkv, C + h * HD, key0 + CHK, p.Ntok, tid);
            vr.load(p.qkv, rowbase, p.ldqkv, 2 * C + h * HD, key0 + CHK, p.Ntok, tid);
        }
        if (!active) continue;
        attend_chunk<CH1, false, 1>(Kl, (const __bf16*)Vl, qf, key0, p.Ntok, sl2, lane, key0 == 0, m, l, o, 0);
        if (key0 + CH1 * 32 < p.Ntok)                     // (wave-uniform) the second step holds at least one real key
            attend_chunk<CH - CH1, false, 1>(Kl, (const __bf16*)Vl, qf, key0 + CH1 * 32, p.Ntok, sl2, lane, false, m, l, o, CH1);
    }
    if (active && q < p.Ntok) store_o(p, rowbase, q, h, bh, hf, m, l, o);
}

// (Round 6 also built the opposite shape -- MANY SMALL workgroups: four waves = 128 queries (keys), 128-row chunks, compact RoPE tables, 40 KB of LDS,
// three or four workgroups per CU so that their softmax, MFMA and load phases overlap and 768 workgroups spread evenly over one round.  Measured
// (profiles/r06_e_long_sequence_ab.txt): forward 227-241 vs 200 us at 4097 tokens (60 vs 65 at 577), backward 985 vs 520 us -- every workgroup stages
// and rotates ALL keys, so halving the queries per workgroup doubles that work, and these kernels are bound by instruction throughput at the
// power-limited clock, not by occupancy or by the half-empty second round: removed.)
// Eight waves, ONE 32-query tile per wave, whole sequence (<= 224 keys) staged once; the keys are consumed in chunks of three tiles with
// the online softmax, so a wave needs 48 score registers instead of 112 and fits 128 VGPRs: two workgroups per CU are 16 waves = FOUR per
// SIMD (the 4-wave / 2-tile form above: two per SIMD at 206 VGPRs).  The forward is bound by dependent latencies (S = K.Q^T -> max ->
// exp -> P.V per tile; global loads -> LDS images -> barrier per unit), not by MFMA or VALU throughput: twice the waves per SIMD is
// what hides them.  Same arithmetic per element as the single-chunk form except that the running maximum of the first chunks rescales
// the output accumulators (exact when the maximum does not change, one fp32 rounding of alpha * o otherwise).
// VROW (round 5, CS_ATTN_FWD8_VROW=1; NOT the default -- measured 2 % slower): V stays row-major in LDS (staged like K, four 16-byte items per
// thread on all 512 threads) and its transposed fragments come from ds_read_b64_tr_b16 -- no 8 x 8 in-register transposes on 224 of the 512
// threads, 16 instead of 32 staging registers; same values into the same MFMAs: bit-identical outputs.
// VM: the V image, see attend_chunk -- 0: the XOR-permuted V^T [64][264]; 1: row-major + transposing reads.
template <bool TAIL, int VM>
__global__ __launch_bounds__(512, 4) void attn_fwd8_kernel(AttnArgs p) {
    constexpr int CH = 7, CHK = CH * 32, NT = 512;
    constexpr bool VROW = VM == 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kl = smem;
    __bf16* Vt = (__bf16*)(smem + CHK * 128);
    float* rt = (float*)(smem + CHK * 128 + (VROW ? CHK * 128 : HD * VT_LD * 2));      // compact RoPE tables [4][g][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int C = p.H * HD;
    const size_t rowbase = (size_t)b * p.Ntok;
    const float sl2 = p.scale * LOG2E;
    const int last = p.Ntok - 1;
#ifdef CS_ABLATION_SWITCHES
    {   // timeline experiments (profiles/r04_t_attention_timeline.md): a subset of the first residents starts (dbg >> 8) x ~4.3 us late
        // (64: odd ids, 128: every second CU, else ids 256..511); CS_ATTN_TRACE: where and when every workgroup ran
        const int n = (p.dbg >> 8) & 0xff, id = blockIdx.y;
        const bool late = (p.dbg & 64) ? (id & 1) && id < 512 : (p.dbg & 128) ? ((id >> 3) & 1) && id < 512 : id >= 256 && id < 512;
        if (n && late) for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
        if (p.trace && threadIdx.x == 0) {
            p.trace[(size_t)id * 8 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
            p.trace[(size_t)id * 8 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
        }
    }
#endif
    ATT_TRACE(2);
    // every global load of the workgroup ahead of the first dependent instruction: tables (oldest: vmcnt retires in order), K, V, Q
    float tab[4];
    {
        const int i = min(tid, p.grid * 32 - 1), r = i >> 5, d = i & 31;       // g * 32 <= 448 entries per table
        tab[0] = p.cos_t[(size_t)(r * p.grid) * HD + d];
        tab[1] = p.sin_t[(size_t)(r * p.grid) * HD + d];
        tab[2] = p.cos_t[(size_t)r * HD + 32 + d];
        tab[3] = p.sin_t[(size_t)r * HD + 32 + d];
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int KI = (CHK * 8 + NT - 1) / NT;
    U128 kr[VROW ? 1 : KI], vin[VROW ? 1 : 8], qraw[4];
    RowRegs<CHK> krow, vrow;               // VROW: K and V rows as plain vectors, staged by the same routine as the backward's images
    const __bf16* kbase = p.qkv + C + h * HD;
    const __bf16* vbase = p.qkv + 2 * C + h * HD;
    const int vkb = min(tid, CHK - 1) >> 3, vc = tid & 7;      // one (key block, dim chunk) item per thread, threads >= 224 repeat the last
    if constexpr (VROW) {
        krow.load(p.qkv, rowbase, p.ldqkv, C + h * HD, 0, p.Ntok, tid);
        vrow.load(p.qkv, rowbase, p.ldqkv, 2 * C + h * HD, 0, p.Ntok, tid);
    } else {
#pragma unroll
        for (int it = 0; it < KI; ++it) {      // branch-free: rows past the sequence re-read its last row (masked to p = 0 exactly)
            const int idx = min(tid + it * NT, CHK * 8 - 1), tok = min(idx >> 3, last);
            kr[it].u = *(const uint4*)(kbase + (rowbase + tok) * p.ldqkv + (idx & 7) * 8);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int tok = min(vkb * 8 + i, last);
            vin[i].u = *(const uint4*)(vbase + (rowbase + tok) * p.ldqkv + vc * 8);
        }
    }
    const int q0 = wave * 32, q = q0 + l31, qc = min(q, last);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qraw[ks].u = *(const uint4*)(p.qkv + (rowbase + qc) * p.ldqkv + h * HD + ks * 16 + hf * 8);
    if (tid < p.grid * 32) {
        rt[tid] = tab[0];
        rt[(p.grid << 5) + tid] = tab[1];
        rt[(2 * p.grid << 5) + tid] = tab[2];
        rt[(3 * p.grid << 5) + tid] = tab[3];
    }
    __syncthreads();
    ATT_TRACE(3);
    if constexpr (VROW) {
        krow.template store<true>(Kl, rt, p.grid, p.inv_grid, 0, p.Ntok, tid);
        vrow.template store<false>((char*)Vt, rt, p.grid, p.inv_grid, 0, p.Ntok, tid);
    } else {
#pragma unroll
        for (int it = 0; it < KI; ++it) {      // K: rotate + swizzled LDS image
            const int idx = tid + it * NT, r = idx >> 3, c = idx & 7;
            if (idx < CHK * 8) {
                if (r > 0 && r < p.Ntok && !ATT_ABL(p, 2)) rope8_lds(kr[it], rt, p.grid, p.inv_grid, r, c);
                *(uint4*)(Kl + k_off(r, c)) = kr[it].u;
            }
        }
        if (tid < CHK) {                   // V: 8x8 in-register transpose -> V^T image
            const int pos = (vkb ^ vc) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                U128 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o.e[i] = vin[i].e[j];
                *(uint4*)(Vt + (vc * 8 + j) * VT_LD + pos) = o.u;
            }
        }
    }
    __syncthreads();
    ATT_TRACE(4);
    if (q0 >= p.Ntok) return;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (qc > 0 && !ATT_ABL(p, 2)) rope8_lds(qraw[ks], rt, p.grid, p.inv_grid, qc, ks * 2 + hf);
        qf[ks] = qraw[ks].h;
    }
    float m = -INFINITY, l = 0.f;
    f32x16 o[2] = {zero16(), zero16()};
    if (ATT_ABL(p, 1)) { l = 1.f; m = 0.f; o[0][0] = bf2f(qf[0][0]); }
    else if (TAIL) {                       // 192 < Ntok <= 224 (the 14x14 + CLS grid): two full chunks and the ragged last tile
        attend_chunk<3, false, VM>(Kl, Vt, qf, 0, p.Ntok, sl2, lane, true, m, l, o, 0);
        attend_chunk<3, false, VM>(Kl, Vt, qf, 96, p.Ntok, sl2, lane, false, m, l, o, 3);
        attend_chunk<1, true, VM>(Kl, Vt, qf, 192, p.Ntok, sl2, lane, false, m, l, o, 6);
    } else {
        attend_chunk<3, false, VM>(Kl, Vt, qf, 0, p.Ntok, sl2, lane, true, m, l, o, 0);
        if (p.Ntok > 96) attend_chunk<3, false, VM>(Kl, Vt, qf, 96, p.Ntok, sl2, lane, false, m, l, o, 3);
        if (p.Ntok > 192) attend_chunk<1, false, VM>(Kl, Vt, qf, 192, p.Ntok, sl2, lane, false, m, l, o, 6);
    }
    ATT_TRACE(5);
    if (q < p.Ntok) store_o(p, rowbase, q, h, bh, hf, m, l, o);
    ATT_TRACE(6);
}

// Round 6: THREE (crop, head) units per CU -- built on VERDICT r5's lead, bit-identical, and a TIE with attn_fwd8_kernel (424-426 vs 419-422 us per
// 1024-crop launch): the forward is no longer bound by units in flight, its VALU is busy 65-73 % of a launch (profiles/r06_c_attention_pipes.md).
// NOT the default; CS_ATTN_FWD4=1 selects it (read per launch).  What it is:  attn_fwd8_kernel is bound by units in flight per CU x latency, not by a pipe: two units of 69 KB
// alternate a 5.8 us load phase with a 4.5 us attend phase and for 29 % of a launch no unit of a CU is attending
// (profiles/r04_t_attention_timeline.md).  A third resident unit needs <= 54.6 KB of LDS and, with eight waves per unit, <= 80 VGPRs -- the
// attend phase holds 32 output + 16 query + 48 score registers.  So the unit shrinks instead: FOUR waves, each attending its query tiles one
// after the other (tiles w and w + 4 of the seven; same attend_chunk, same 3 + 3 + 1 key chunks, hence the same bits), 12 waves per CU = three per
// SIMD at <= 168 VGPRs, and LDS images sized for the sequence: K [200][64] (25.0 KB), V^T [64][200] (25.0 KB, layout 3 of attend_chunk) and
// the RoPE tables stored ONCE: rope.py:118-142 builds the row part and the column part from the same `freqs` tensor and repeats every
// frequency for the two dims of a pair, so cos_row[r][2j] == cos_row[r][2j+1] == cos_col[r][2j]: one [g][16] table each for cos and sin
// (1.75 KB at g = 14 instead of 7 KB) -- a documented precondition of the C ABI (include/clipself_hip.h), checked by HipOps once per table.
// 51.75 KB per unit.  For square grids up to 14 x 14 (Ntok <= 197).
constexpr int K4_ROWS = 200;

template <int NW, bool TAIL>
__global__ __launch_bounds__(NW * 64, 3) void attn_fwd4_kernel(AttnArgs p) {     // 3 units per CU: 12 waves = 3 per SIMD (five waves per unit at 128 VGPRs spilled 31-46 dwords: 699 us)
    constexpr int NT = NW * 64, KI = (K4_ROWS * 8 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kl = smem;
    __bf16* Vt = (__bf16*)(smem + K4_ROWS * 128);
    float* rt = (float*)(smem + K4_ROWS * 128 + HD * VT4_LD * 2);      // cos [g][16] | sin [g][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int C = p.H * HD;
    const size_t rowbase = (size_t)b * p.Ntok;
    const float sl2 = p.scale * LOG2E;
    const int last = p.Ntok - 1;
    // every global load of the workgroup ahead of the first dependent instruction: tables (oldest: vmcnt retires in order), K, V, Q of both tiles
    float tab[2];
    {
        const int i = min(tid, p.grid * 16 - 1), r = i >> 4, j = i & 15;       // the column part of grid row 0: token r, dims 32 + 2j
        tab[0] = p.cos_t[(size_t)r * HD + 32 + 2 * j];
        tab[1] = p.sin_t[(size_t)r * HD + 32 + 2 * j];
    }
    __builtin_amdgcn_sched_barrier(0);
    U128 kr[KI], vin[8], qraw[2][4];
    const __bf16* kbase = p.qkv + C + h * HD;
    const __bf16* vbase = p.qkv + 2 * C + h * HD;
    const int vkb = min(tid, K4_ROWS - 1) >> 3, vc = tid & 7;      // one (key block, dim chunk) item per thread, threads >= 200 repeat the last
#pragma unroll
    for (int it = 0; it < KI; ++it) {          // branch-free: rows past the sequence re-read its last row (masked to p = 0 exactly)
        const int idx = min(tid + it * NT, K4_ROWS * 8 - 1), tok = min(idx >> 3, last);
        kr[it].u = *(const uint4*)(kbase + (rowbase + tok) * p.ldqkv + (idx & 7) * 8);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int tok = min(vkb * 8 + i, last);
        vin[i].u = *(const uint4*)(vbase + (rowbase + tok) * p.ldqkv + vc * 8);
    }
    // the wave that owns a single query tile (seven tiles on NW waves) rotates with the unit index
    int wq = wave + bh % NW;
    wq = wq >= NW ? wq - NW : wq;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int qc = min((wq + ps * NW) * 32 + l31, last);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qraw[ps][ks].u = *(const uint4*)(p.qkv + (rowbase + qc) * p.ldqkv + h * HD + ks * 16 + hf * 8);
    }
    if (tid < p.grid * 16) {
        rt[tid] = tab[0];
        rt[(p.grid << 4) + tid] = tab[1];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < KI; ++it) {          // K: rotate + swizzled LDS image
        const int idx = tid + it * NT, r = idx >> 3, c = idx & 7;
        if (idx < K4_ROWS * 8) {
            if (r > 0 && r < p.Ntok) rope8c(kr[it], rt, p.grid, p.inv_grid, r, c);
            *(uint4*)(Kl + k_off(r, c)) = kr[it].u;
        }
    }
    if (tid < K4_ROWS) {                       // V: 8x8 in-register transpose -> V^T image, key block rotated by the dim chunk's upper bit
        const int rot = vkb + (vc >> 2), pos = (rot >= 25 ? rot - 25 : rot) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            U128 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.e[i] = vin[i].e[j];
            *(uint4*)(Vt + (vc * 8 + j) * VT4_LD + pos) = o.u;
        }
    }
    __syncthreads();
#pragma nounroll                               // one copy of the attend code: the second tile's rows are selected into the first's registers
    for (int ps = 0; ps < 2; ++ps) {
        const int q0 = (wq + ps * NW) * 32;
        if (q0 >= p.Ntok) break;               // wave-uniform
        // an opaque copy of the lane id: without it every lane mask and LDS address of the attend code is loop-invariant, gets hoisted in
        // front of the loop and is spilled there (344 SGPRs in the first build)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int hf = ln >> 5, l31 = ln & 31;
        const int q = q0 + l31, qc = min(q, last);
        bf16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            U128 t;
            t.u = ps ? qraw[1][ks].u : qraw[0][ks].u;
            if (qc > 0) rope8c(t, rt, p.grid, p.inv_grid, qc, ks * 2 + hf);
            qf[ks] = t.h;
        }
        float m = -INFINITY, l = 0.f;
        f32x16 o[2] = {zero16(), zero16()};
        if (TAIL) {                            // 192 < Ntok <= 200 (the 14x14 + CLS grid): two full chunks and the ragged last tile
            attend_chunk<3, false, 3>(Kl, Vt, qf, 0, p.Ntok, sl2, ln, true, m, l, o, 0);
            attend_chunk<3, false, 3>(Kl, Vt, qf, 96, p.Ntok, sl2, ln, false, m, l, o, 3);
            attend_chunk<1, true, 3>(Kl, Vt, qf, 192, p.Ntok, sl2, ln, false, m, l, o, 6);
        } else {
            attend_chunk<3, false, 3>(Kl, Vt, qf, 0, p.Ntok, sl2, ln, true, m, l, o, 0);
            if (p.Ntok > 96) attend_chunk<3, false, 3>(Kl, Vt, qf, 96, p.Ntok, sl2, ln, false, m, l, o, 3);
            if (p.Ntok > 192) attend_chunk<1, false, 3>(Kl, Vt, qf, 192, p.Ntok, sl2, ln, false, m, l, o, 6);
        }
        if (q < p.Ntok) store_o(p, rowbase, q, h, bh, hf, m, l, o);
    }
}

// ------------------------------------------------------------------------------------------------ backward
// dsum[b,h,n] = sum_d dO[b,n,h,d] * O[b,n,h,d]
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const __bf16* __restrict__ o, const __bf16* __restrict__ dout, float* __restrict__ dsum,
                                                            int B, int Ntok, int H, int ldo) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (row, head, chunk of 8)
    const long total = (long)B * Ntok * H * 8;
    float s = 0.f;
    long row = 0; int h = 0;
    const bool ok = idx < total;
    if (ok) {
        const int c = (int)(idx & 7);
        h = (int)((idx >> 3) % H);
        row = (idx >> 3) / H;
        U128 a, g;
        a.u = *(const uint4*)(o + row * ldo + h * HD + c * 8);
        g.u = *(const uint4*)(dout + row * ldo + h * HD + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += bf2f(a.e[e]) * bf2f(g.e[e]);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (ok && (idx & 7) == 0) {
        const long b = row / Ntok, n = row - b * Ntok;
        dsum[((size_t)b * H + h) * Ntok + n] = s;
    }
}

// q | k of every (token, head), rotated once: out [B*N, 2C] bf16.  One thread per 8 head dims; token 0 (CLS) passes through.  The same rope8 on the
// same table entries as the kernels' own staging (rope8_lds reads copies of these entries): the same bits.
__global__ __launch_bounds__(256) void rope_qk_kernel(const __bf16* __restrict__ qkv, const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                      __bf16* __restrict__ out, long rows, int Ntok, int C, int ldqkv) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (row, chunk of 8 over the 2C columns of q | k)
    const int cpr = C >> 2;                                            // 2C / 8 chunks per row
    if (idx >= rows * cpr) return;
    const long row = idx / cpr;
    const int c = (int)(idx - row * cpr), col = c * 8, tok = (int)(row % Ntok);
    U128 v;
    v.u = *(const uint4*)(qkv + row * ldqkv + col);
    if (tok > 0) rope8(v, cos_t + (size_t)(tok - 1) * HD + (col & 63), sin_t + (size_t)(tok - 1) * HD + (col & 63));
    *(uint4*)(out + row * (2 * C) + col) = v.u;
}

// inverse rotation of an accumulator tile pair (lane = token, registers = head-dim) + bf16 store, 8 bytes per 4 dims
__device__ __forceinline__ void store_grad_tile(const f32x16 (&g)[2], __bf16* dst, bool rope, const float* cs, const float* sn, int hf) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = dt * 32 + g4 * 8 + hf * 4;
            float v[4] = {g[dt][g4 * 4], g[dt][g4 * 4 + 1], g[dt][g4 * 4 + 2], g[dt][g4 * 4 + 3]};
            if (rope) {
                const float4 c = *(const float4*)(cs + d), s = *(const float4*)(sn + d);
                // (contraction written out, as in rope8: the same bits in every kernel that inlines this)
                const float a0 = __builtin_fmaf(v[0], c.x, v[1] * s.y), a1 = __builtin_fmaf(v[1], c.y, -(v[0] * s.x));
                const float a2 = __builtin_fmaf(v[2], c.z, v[3] * s.w), a3 = __builtin_fmaf(v[3], c.w, -(v[2] * s.z));
                v[0] = a0; v[1] = a1; v[2] = a2; v[3] = a3;
            }
            U64 t;
#pragma unroll
            for (int i = 0; i < 4; ++i) t.e[i] = f2bf(v[i]);
            *(uint2*)(dst + d) = t.u;
        }
}

// dQ: wave = 32 queries, loops over key chunks.  dS^T = P^T o (dP^T - D) * scale ; dQ^T += K^T . dS^T
template <int CH>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(AttnArgs p) {
    constexpr int CHK = CH * 32, VLD = CHK + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kl = smem;
    char* Vl = smem + CHK * 128;
    __bf16* Kt = (__bf16*)(smem + 2 * CHK * 128);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int C = p.H * HD;
    const size_t rowbase = (size_t)b * p.Ntok;
    const int q = blockIdx.x * 256 + wave * 32 + l31;
    const int qc = min(q, p.Ntok - 1);
    const bool wave_active = (blockIdx.x * 256 + wave * 32) < p.Ntok;
    const float sl2 = p.scale * LOG2E;

    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int d0 = ks * 16 + hf * 8;
        U128 t;
        t.u = *(const uint4*)(p.qkv + (rowbase + qc) * p.ldqkv + h * HD + d0);
        if (qc > 0) rope8(t, p.cos_t + (size_t)(qc - 1) * HD + d0, p.sin_t + (size_t)(qc - 1) * HD + d0);
        qf[ks] = t.h;
        t.u = *(const uint4*)(p.dout + (rowbase + qc) * p.ldo + h * HD + d0);
        dof[ks] = t.h;
    }
    const float lse2 = p.lse_in[(size_t)bh * p.Ntok + qc] * LOG2E;
    const float dq_sum = p.dsum[(size_t)bh * p.Ntok + qc];
    f32x16 dq[2] = {zero16(), zero16()};

    for (int key0 = 0; key0 < p.Ntok; key0 += CHK) {
        __syncthreads();
        stage_rows<CHK, true, true>(p.qkv, rowbase, p.ldqkv, C + h * HD, key0, p.Ntok, p.cos_t, p.sin_t, Kl, Kt, VLD, tid);
        stage_rows<CHK, false, false>(p.qkv, rowbase, p.ldqkv, 2 * C + h * HD, key0, p.Ntok, nullptr, nullptr, Vl, nullptr, 0, tid);
        __syncthreads();
        if (!wave_active) continue;
#pragma nounroll
        for (int t = 0; t < CH; ++t) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(Kl, t * 32 + l31, ks, hf), qf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(Vl, t * 32 + l31, ks, hf), dof[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = key0 + t * 32 + mfma32_row(e, lane);
                const float pv = key < p.Ntok ? exp2f(s[e] * sl2 - lse2) : 0.f;
                s[e] = pv * (dp[e] - dq_sum) * p.scale;
            }
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const bf16x8 db = pack8(s, c2);
                const int base = t * 32 + c2 * 16;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(load_t_frag(Kt, VLD, dt * 32 + l31, base, hf), db, dq[dt], 0, 0, 0);
            }
        }
    }
    if (wave_active && q < p.Ntok) {
        const size_t pos = (size_t)(q > 0 ? q - 1 : 0) * HD;
        store_grad_tile(dq, p.out + (rowbase + q) * p.ldqkv + h * HD, q > 0, p.cos_t + pos, p.sin_t + pos, hf);
    }
}

// dK, dV: wave = 32 keys, loops over query chunks.  S = Q K^T (lane = key, registers = queries)
//   dV^T += dO^T . P ;  dK^T += Q^T . dS
template <int CH>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(AttnArgs p) {
    constexpr int CHQ = CH * 32, VLD = CHQ + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ql = smem;
    char* Gl = smem + CHQ * 128;
    __bf16* Qt = (__bf16*)(smem + 2 * CHQ * 128);
    __bf16* Gt = Qt + HD * VLD;
    float* lse_s = (float*)(Gt + HD * VLD);
    float* dsum_s = lse_s + CHQ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int C = p.H * HD;
    const size_t rowbase = (size_t)b * p.Ntok;
    const int key = blockIdx.x * 256 + wave * 32 + l31;
    const int kc = min(key, p.Ntok - 1);
    const bool wave_active = (blockIdx.x * 256 + wave * 32) < p.Ntok;
    const float sl2 = p.scale * LOG2E;

    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int d0 = ks * 16 + hf * 8;
        U128 t;
        t.u = *(const uint4*)(p.qkv + (rowbase + kc) * p.ldqkv + C + h * HD + d0);
        if (kc > 0) rope8(t, p.cos_t + (size_t)(kc - 1) * HD + d0, p.sin_t + (size_t)(kc - 1) * HD + d0);
        kf[ks] = t.h;
        t.u = *(const uint4*)(p.qkv + (rowbase + kc) * p.ldqkv + 2 * C + h * HD + d0);
        vf[ks] = t.h;
    }
    f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};

    for (int q0 = 0; q0 < p.Ntok; q0 += CHQ) {
        __syncthreads();
        stage_rows<CHQ, true, true>(p.qkv, rowbase, p.ldqkv, h * HD, q0, p.Ntok, p.cos_t, p.sin_t, Ql, Qt, VLD, tid);
        stage_rows<CHQ, false, true>(p.dout, rowbase, p.ldo, h * HD, q0, p.Ntok, nullptr, nullptr, Gl, Gt, VLD, tid);
        for (int i = tid; i < CHQ; i += 512) {
            const int qi = q0 + i;
            lse_s[i] = qi < p.Ntok ? p.lse_in[(size_t)bh * p.Ntok + qi] * LOG2E : INFINITY;   // +inf -> P = 0 for padded queries
            dsum_s[i] = qi < p.Ntok ? p.dsum[(size_t)bh * p.Ntok + qi] : 0.f;
        }
        __syncthreads();
        if (!wave_active) continue;
#pragma nounroll
        for (int t = 0; t < CH; ++t) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(Ql, t * 32 + l31, ks, hf), kf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(Gl, t * 32 + l31, ks, hf), vf[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int qi = t * 32 + mfma32_row(e, lane);
                const float pv = key < p.Ntok ? exp2f(s[e] * sl2 - lse_s[qi]) : 0.f;
                s[e] = pv;
                dp[e] = pv * (dp[e] - dsum_s[qi]) * p.scale;
            }
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const bf16x8 pb = pack8(s, c2), db = pack8(dp, c2);
                const int base = t * 32 + c2 * 16;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(load_t_frag(Gt, VLD, dt * 32 + l31, base, hf), pb, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(load_t_frag(Qt, VLD, dt * 32 + l31, base, hf), db, dk[dt], 0, 0, 0);
                }
            }
        }
    }
    if (wave_active && key < p.Ntok) {
        const size_t pos = (size_t)(key > 0 ? key - 1 : 0) * HD;
        __bf16* row = p.out + (rowbase + key) * p.ldqkv + h * HD;
        store_grad_tile(dk, row + C, key > 0, p.cos_t + pos, p.sin_t + pos, hf);
        store_grad_tile(dv, row + 2 * C, false, nullptr, nullptr, hf);
    }
}

// ---- round 5: the backward kernels re-staged ----------------------------------------------------------------------------------------
// The round-1 kernels above stage a chunk with a loop of (global load -> RoPE from the global tables -> LDS store): 3.5 serial memory round
// trips per operand and chunk, 64 bytes of table per 16 bytes of q / k, a second, transposed image written with 2-byte LDS stores, and no
// overlap with the MFMA phase -- at the reference's recipe shape (4097 tokens) a chunk took ~11.8 us for ~3.2 us of MFMA work and the two
// kernels were 43 % of the step (profiles/r05_c_recipe_shape.md).  Here: every row of a chunk is requested up front, the NEXT chunk's rows
// are requested before the current chunk is consumed (register prefetch), RoPE comes from the compact LDS tables of the forward kernel, and
// the transposed operands (K^T, Q^T, dO^T) are read straight from the row-major images with ds_read_b64_tr_b16 -- no transposed images.
// Same products, same accumulation order, same rotated values: bit-identical gradients (tests/test_gpu_ops.py).
// dQ: wave = 32 queries, loops over key chunks.  dS^T = P^T o (dP^T - D) * scale ; dQ^T += K^T . dS^T
// SINGLE: the whole sequence is one chunk (Ntok <= CH * 32: the 14x14 grid) -- no chunk loop, no prefetch registers, 128 VGPRs: two
// workgroups per CU
// PRE (round 6): q and k come ROTATED from p.qk ([B*N, ldqk] bf16 = q | k, written once per launch by rope_qk_kernel): a workgroup of these
// kernels walks the whole sequence, so the K rows (dQ kernel) and the Q rows (dK/dV kernel) of an (image, head) pair were rotated again by
// every one of its 17 row blocks at the recipe's 4097 tokens -- a fifth of the kernels' VALU instructions.  Same rotation arithmetic, same bits.
template <int CH, bool SINGLE, bool PRE = false>
__global__ __launch_bounds__(512, SINGLE ? 4 : 2) void attn_bwd_dq2_kernel(AttnArgs p) {
    constexpr int CHK = CH * 32, NT = 512, NW = 8;
    constexpr bool CT = false, PF = true;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kl = smem;
    char* Vl = smem + CHK * 128;
    float* rt = (float*)(smem + 2 * CHK * 128);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l31 = lane & 31;
    const RowBlock rb = map_block(p, NW);
    const int bh = rb.bh, b = bh / p.H, h = bh - b * p.H;
    const int C = p.H * HD;
    const size_t rowbase = (size_t)b * p.Ntok;
    const int q = (rb.tile0 + wave) * 32 + l31;
    const int qc = min(q, p.Ntok - 1);
    const bool wave_active = wave < rb.nt && (rb.tile0 + wave) * 32 < p.Ntok;
    const float sl2 = p.scale * LOG2E;

    const __bf16* ksrc = PRE ? p.qk : p.qkv;
    const int kld = PRE ? p.ldqk : p.ldqkv;
    RowRegs<CHK, NT> kr, vr;
    kr.load(ksrc, rowbase, kld, C + h * HD, 0, p.Ntok, tid);
    vr.load(p.qkv, rowbase, p.ldqkv, 2 * C + h * HD, 0, p.Ntok, tid);
    bf16x8 qf[4], dof[4];
    if constexpr (PRE) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(p.qk + (rowbase + qc) * p.ldqk + h * HD + ks * 16 + hf * 8);
    } else {
        load_rope_tables<NT>(rt, p.cos_t, p.sin_t, p.grid, tid);
        __syncthreads();
        load_q_frags<CT>(p, rt, rowbase, qc, h, hf, qf);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) dof[ks] = *(const bf16x8*)(p.dout + (rowbase + qc) * p.ldo + h * HD + ks * 16 + hf * 8);
    const float lse2 = p.lse_in[(size_t)bh * p.Ntok + qc] * LOG2E;
    const float dq_sum_s = p.dsum[(size_t)bh * p.Ntok + qc] * p.scale;
    f32x16 dq[2] = {zero16(), zero16()};
    // row fragments (lane = key row): LDS row (row >> 1), slot ((row & 1) * 8 | chunk) ^ ((row >> 1) & 15), as attend_chunk
    const int k_base = (l31 >> 1) << 8, par8 = (l31 & 1) << 3, sw = l31 >> 1;
    // transposed fragments (lane = head dim): contraction slots in the accumulator order -- half hf supplies keys base + 4 hf + {0..3, 8..11}
    int toff[2][2][2];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            toff[c2][dt][0] = tr_lane_off(c2 * 16 + 4 * hf, dt * 32, lane);
            toff[c2][dt][1] = tr_lane_off(c2 * 16 + 8 + 4 * hf, dt * 32, lane);
        }

    for (int key0 = 0; key0 < (SINGLE ? 1 : p.Ntok); key0 += CHK) {
        if (key0) __syncthreads();                     // every wave is done with the previous chunk's images
        kr.template store<!PRE, true, CT>(Kl, rt, p.grid, p.inv_grid, key0, p.Ntok, tid);
        vr.template store<false>(Vl, rt, p.grid, p.inv_grid, key0, p.Ntok, tid);
        __syncthreads();
        if (!SINGLE && PF && key0 + CHK < p.Ntok) {    // the next chunk's rows travel while this one is consumed
            kr.load(ksrc, rowbase, kld, C + h * HD, key0 + CHK, p.Ntok, tid);
            vr.load(p.qkv, rowbase, p.ldqkv, 2 * C + h * HD, key0 + CHK, p.Ntok, tid);
        }
        // one key tile: RAGGED = the tile may hold padding keys (only in the sequence's last chunk)
        auto tile = [&](int t, auto ragged) {          // (t is a constant after unrolling)
            constexpr bool RAGGED = decltype(ragged)::value;
            f32x16 s, dp;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = t * (16 * 256) + k_base + (((par8 | (ks * 2 + hf)) ^ sw) << 4);       // t * 4096 becomes the immediate offset of every LDS read
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Kl + off), qf[ks], ks ? s : Z16, 0, 0, 0);       // first step onto the inline 0
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Vl + off), dof[ks], ks ? dp : Z16, 0, 0, 0);
            }
            // the softmax recomputation is the VALU half of this kernel (16 elements per lane and tile against 12 MFMAs): one FMA + the raw
            // v_exp_f32 (argument <= ~0, p in [0, 1]) + one FMA + one multiply per element.  Padding keys need no mask: their K rows are
            // zeros in the image (ZPAD), so their dS columns meet a zero K^T column.
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = __builtin_amdgcn_exp2f(fmaf(s[e], sl2, -lse2)) * fmaf(dp[e], p.scale, -dq_sum_s);
            // ... but 0 * inf is NaN: a padding key's score is 0, so its "probability" is exp2(-lse2), which overflows for a query whose
            // scaled scores are all below ~ -88 (lse < -88) and would poison the whole dQ row.  Only the sequence's ragged last tile holds
            // padding keys (wave-uniform branch, as in attend_chunk): their dS is set to 0 there.
            if (RAGGED && key0 + t * 32 + 32 > p.Ntok) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (key0 + t * 32 + mfma32_row(e, lane) >= p.Ntok) s[e] = 0.f;
            }
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const bf16x8 db = pack8(s, c2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16x8 kt = tr_join2(tr_read4(Kl + t * (16 * 256) + toff[c2][dt][0]), tr_read4(Kl + t * (16 * 256) + toff[c2][dt][1]));
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt, db, dq[dt], 0, 0, 0);
                }
            }
        };
        if (wave_active) {
            // (Round 6 tried a branch-free path for the chunks that lie inside the sequence -- CH tiles as ONE basic block, no per-tile exit, no
            // padding mask -- so that hipcc's scheduler could put tile t + 1's S / dP MFMAs beside tile t's softmax arithmetic: it interleaves
            // until the registers run out (dK/dV: 35-42 dwords spilled at 256 VGPRs) and the pair of kernels got 8.6 % SLOWER at 4097 tokens
            // (522 vs 480 us; the dQ kernel alone, which does not spill: 219 vs 213 us -- profiles/r06_g_full_chunk_path.txt).  The per-tile exits stay.)
#pragma unroll
            for (int t = 0; t < CH; ++t) {
                if (key0 + t * 32 >= p.Ntok) break;        // wave-uniform: a tile of padding keys only (p = 0 everywhere)
                tile(t, std::true_type{});
            }
        }
    }
    if (wave_active && q < p.Ntok) {
        const size_t pos = (size_t)(q > 0 ? q - 1 : 0) * HD;
        store_grad_tile(dq, p.out + (rowbase + q) * p.ldqkv + h * HD, q > 0, p.cos_t + pos, p.sin_t + pos, hf);
    }
}

// dK, dV: wave = 32 keys, loops over query chunks.  S = Q K^T (lane = key, registers = queries);  dV^T += dO^T . P ;  dK^T += Q^T . dS
template <int CH, bool SINGLE, bool PRE = false>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv2_kernel(AttnArgs p) {       // 64 + 32 + 32 accumulator / operand registers: no 128-register form
    constexpr int CHQ = CH * 32, NT = 512, NW = 8;
    constexpr bool CT = false, PF = true;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ql = smem;
    char* Gl = smem + CHQ * 128;
    float* lse_s = (float*)(smem + 2 * CHQ * 128);
    float* dsum_s = lse_s + CHQ;
    float* rt = dsum_s + CHQ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l31 = lane & 31;
    const RowBlock rb = map_block(p, NW);
    const int bh = rb.bh, b = bh / p.H, h = bh - b * p.H;
    const int C = p.H * HD;
    const size_t rowbase = (size_t)b * p.Ntok;
    const int key = (rb.tile0 + wave) * 32 + l31;
    const int kc = min(key, p.Ntok - 1);
    const bool wave_active = wave < rb.nt && (rb.tile0 + wave) * 32 < p.Ntok;
    const float sl2 = p.scale * LOG2E;

    const __bf16* qsrc = PRE ? p.qk : p.qkv;
    const int qld = PRE ? p.ldqk : p.ldqkv;
    RowRegs<CHQ, NT> qr, gr;
    qr.load(qsrc, rowbase, qld, h * HD, 0, p.Ntok, tid);
    gr.load(p.dout, rowbase, p.ldo, h * HD, 0, p.Ntok, tid);
    float lse_r = 0.f, dsum_r = 0.f;                   // thread i < CHQ carries query q0 + i's statistics
    auto load_stats = [&](int q0) {
        if (tid < CHQ) {
            const int qi = q0 + tid;
            lse_r = qi < p.Ntok ? p.lse_in[(size_t)bh * p.Ntok + qi] * LOG2E : INFINITY;      // +inf -> P = 0 for padded queries
            dsum_r = qi < p.Ntok ? p.dsum[(size_t)bh * p.Ntok + qi] * p.scale : 0.f;    // D * scale: dS = P (dP * scale - D * scale)
        }
    };
    load_stats(0);
    if constexpr (!PRE) {
        load_rope_tables<NT>(rt, p.cos_t, p.sin_t, p.grid, tid);
        __syncthreads();
    }
    bf16x8 kf[4], vf[4];
    {
        U128 t[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            t[ks].u = PRE ? *(const uint4*)(p.qk + (rowbase + kc) * p.ldqk + C + h * HD + ks * 16 + hf * 8)
                          : *(const uint4*)(p.qkv + (rowbase + kc) * p.ldqkv + C + h * HD + ks * 16 + hf * 8);
            vf[ks] = *(const bf16x8*)(p.qkv + (rowbase + kc) * p.ldqkv + 2 * C + h * HD + ks * 16 + hf * 8);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (!PRE && kc > 0) rope8_lds(t[ks], rt, p.grid, p.inv_grid, kc, ks * 2 + hf);
            kf[ks] = t[ks].h;
        }
    }
    f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
    const int k_base = (l31 >> 1) << 8, par8 = (l31 & 1) << 3, sw = l31 >> 1;
    int toff[2][2][2];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            toff[c2][dt][0] = tr_lane_off(c2 * 16 + 4 * hf, dt * 32, lane);
            toff[c2][dt][1] = tr_lane_off(c2 * 16 + 8 + 4 * hf, dt * 32, lane);
        }

    for (int q0 = 0; q0 < (SINGLE ? 1 : p.Ntok); q0 += CHQ) {
        if (q0) __syncthreads();
        qr.template store<!PRE, false, CT>(Ql, rt, p.grid, p.inv_grid, q0, p.Ntok, tid);
        gr.template store<false>(Gl, rt, p.grid, p.inv_grid, q0, p.Ntok, tid);
        if (tid < CHQ) { lse_s[tid] = lse_r; dsum_s[tid] = dsum_r; }
        __syncthreads();
        if (!SINGLE && PF && q0 + CHQ < p.Ntok) {
            qr.load(qsrc, rowbase, qld, h * HD, q0 + CHQ, p.Ntok, tid);
            gr.load(p.dout, rowbase, p.ldo, h * HD, q0 + CHQ, p.Ntok, tid);
            load_stats(q0 + CHQ);
        }
        auto tile = [&](int t) {
            f32x16 s, dp;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = t * (16 * 256) + k_base + (((par8 | (ks * 2 + hf)) ^ sw) << 4);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Ql + off), kf[ks], ks ? s : Z16, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Gl + off), vf[ks], ks ? dp : Z16, 0, 0, 0);
            }
            // a lane owns ONE key: a padding key's column is never stored, so nothing is masked here; padding queries carry lse = +inf
            // (p = exp2(-inf) = 0).  Four consecutive queries' statistics per 16-byte LDS read (accumulator rows 4k .. 4k + 3 of a half).
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const float4 l4 = *(const float4*)(lse_s + t * 32 + 8 * k4 + 4 * hf), d4 = *(const float4*)(dsum_s + t * 32 + 8 * k4 + 4 * hf);
                const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, ds[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = k4 * 4 + r;
                    const float pv = __builtin_amdgcn_exp2f(fmaf(s[e], sl2, -ls[r]));
                    s[e] = pv;
                    dp[e] = pv * fmaf(dp[e], p.scale, -ds[r]);
                }
            }
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const bf16x8 pb = pack8(s, c2), db = pack8(dp, c2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int o0 = t * (16 * 256) + toff[c2][dt][0], o1 = t * (16 * 256) + toff[c2][dt][1];
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join2(tr_read4(Gl + o0), tr_read4(Gl + o1)), pb, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join2(tr_read4(Ql + o0), tr_read4(Ql + o1)), db, dk[dt], 0, 0, 0);
                }
            }
        };
        if (wave_active) {
#pragma unroll
            for (int t = 0; t < CH; ++t) {
                if (q0 + t * 32 >= p.Ntok) break;          // a tile of padding queries only: P = 0
                tile(t);
            }
        }
    }
    if (wave_active && key < p.Ntok) {
        const size_t pos = (size_t)(key > 0 ? key - 1 : 0) * HD;
        __bf16* row = p.out + (rowbase + key) * p.ldqkv + h * HD;
        store_grad_tile(dk, row + C, key > 0, p.cos_t + pos, p.sin_t + pos, hf);
        store_grad_tile(dv, row + 2 * C, false, nullptr, nullptr, hf);
    }
}

// Block schedule of the 8-wave long-sequence kernels (map_block): returns the 1-D grid size.  CS_ATTN_NOSPLIT=1 (read per launch): the
// rectangular grid of round 5.
int att_num_cus() {                    // compute units of the current device, read once per process (one 8-wave workgroup of these kernels per CU)
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}
int set_schedule(AttnArgs& a, int B, int Ntok, int H, dim3& grid) {
    const int nt = (Ntok + 31) / 32, fb = nt / 8, rem = nt % 8, BH = B * H, F = fb * BH, G = att_num_cus();
    int R = F % G;
    if (getenv("CS_ATTN_NOSPLIT") || fb == 0 || 2 * R > G) R = 0;
    if (R == 0) {                                                       // nothing to split: the rectangular grid
        a.sch_on = 0;
        grid = dim3((Ntok + 255) / 256, BH);
        return 0;
    }
    a.sch_on = 1; a.sch_f0 = F - R; a.sch_r = R; a.sch_bh = BH; a.sch_fb = fb; a.sch_rem = rem;
    a.sch_absorb = rem > 0 && rem <= 4 && R >= BH;                      // every pair's last full block is among the split ones
    grid = dim3(F + R + ((rem > 0 && !a.sch_absorb) ? BH : 0), 1);
    return 1;
}

template <typename K>
void set_lds(K kernel, size_t bytes) { (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); }

int check_common(const char* who, int B, int Ntok, int H, int ldqkv, int ldo) {
    CS_CHECK_ARG(B > 0 && Ntok > 1 && H > 0, "%s: bad shape B=%d N=%d H=%d", who, B, Ntok, H);
    CS_CHECK_ARG(ldqkv % 8 == 0 && ldo % 8 == 0 && ldqkv >= 3 * H * HD && ldo >= H * HD, "%s: bad leading dimensions", who);
    return 0;
}

// ------------------------------------------------------------------------------------------------ CLS-query attention
// The CLS query against all keys, for the frozen teacher's last block, whose output is consumed at the CLS row only
// (eva_vit_model.py:505-519 returns x[:, 0]).  HBM-bound: K and V are streamed once, fully coalesced.
//   * one workgroup per (crop, group of HG heads); a key's 64 head-dim values are spread over 8 lanes (16 B each), so a wave
//     reads 8 whole 128-byte key rows per instruction and the RoPE table entries of a (key, chunk) are loaded once and
//     reused across the HG heads;
//   * arithmetic mirrors attn_fwd_kernel: keys 1.. rotated and rounded to bf16, fp32 scores, exp2(s - max) rounded to
//     bf16 for the P.V product, fp32 row sum of the unrounded exponentials.
// Round 4: the same kernel serves the OpenAI-CLIP family's extra query tokens (cs_attn_query_fwd: open_clip/transformer.py:736-834) -- qpb
// query rows share one image's keys / values, `allow` [queries][Ntok] says which keys a query may attend (no rotary tables there).
template <int HG>
__global__ __launch_bounds__(256) void attn_cls_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ kv,
                                                       const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                       __bf16* __restrict__ out, int Ntok, int H, int ldq, int ldkv, int ldo, float scale,
                                                       int qpb, const unsigned char* __restrict__ allow) {
    extern __shared__ float sm[];                 // [HG][Npad] scores -> probabilities | [HG] row sums | [4 waves][HG*64] partial outputs
    const int Npad = (Ntok + 3) & ~3;
    float* rsum = sm + HG * Npad;
    float* part = rsum + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, h0 = blockIdx.y * HG;
    const int c = tid & 7, kg = tid >> 3;         // 16-byte chunk of the head dim; key slot (32 per pass)
    const __bf16* kbase = kv + (size_t)(b / qpb) * Ntok * ldkv + h0 * HD + c * 8;
    const unsigned char* arow = allow ? allow + (size_t)b * Ntok : nullptr;
    const __bf16* vbase = kbase + H * HD;
    float qf[HG][8];
#pragma unroll
    for (int h = 0; h < HG; ++h) {
        U128 t;
        t.u = *(const uint4*)(q + (size_t)b * ldq + (h0 + h) * HD + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[h][j] = bf2f(t.e[j]);
    }
    const float sl2 = scale * LOG2E;
    for (int key = kg; key < Ntok; key += 32) {
        U128 kk[HG];
#pragma unroll
        for (int h = 0; h < HG; ++h) kk[h].u = *(const uint4*)(kbase + (size_t)key * ldkv + h * HD);
        if (key > 0 && cos_t) {
            const float* cs = cos_t + (size_t)(key - 1) * HD + c * 8;
            const float* sn = sin_t + (size_t)(key - 1) * HD + c * 8;
#pragma unroll
            for (int h = 0; h < HG; ++h) rope8(kk[h], cs, sn);
        }
#pragma unroll
        for (int h = 0; h < HG; ++h) {
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) dot += qf[h][j] * bf2f(kk[h].e[j]);
            dot += __shfl_xor(dot, 1);
            dot += __shfl_xor(dot, 2);
            dot += __shfl_xor(dot, 4);
            if (c == 0) sm[h * Npad + key] = (arow && !arow[key]) ? -INFINITY : dot * sl2;      // masked key: p = exp2(-inf) = 0 exactly
        }
    }
    __syncthreads();
    for (int h = wave; h < HG; h += 4) {          // one wave per head: max, exponentials, row sum
        float* row = sm + h * Npad;
        float mx = -INFINITY;
        for (int key = lane; key < Ntok; key += 64) mx = fmaxf(mx, row[key]);
        mx = wave_max(mx);
        if (mx == -INFINITY) mx = 0.f;             // a query row that allows no key at all (the reference always allows key 0): p = 0 everywhere,
        float sum = 0.f;                           // output row = 0 instead of exp2(-inf + inf) = NaN
        for (int key = lane; key < Ntok; key += 64) {
            const float e = __builtin_amdgcn_exp2f(row[key] - mx);
            sum += e;
            row[key] = bf2f(f2bf(e));
        }
        sum = wave_sum(sum);
        if (lane == 0) rsum[h] = sum > 0.f ? sum : 1.f;
    }
    __syncthreads();
    float acc[HG][8];
#pragma unroll
    for (int h = 0; h < HG; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[h][j] = 0.f;
    for (int key = kg; key < Ntok; key += 32) {
        U128 vv[HG];
#pragma unroll
        for (int h = 0; h < HG; ++h) vv[h].u = *(const uint4*)(vbase + (size_t)key * ldkv + h * HD);
#pragma unroll
        for (int h = 0; h < HG; ++h) {
            const float pk = sm[h * Npad + key];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[h][j] += pk * bf2f(vv[h].e[j]);
        }
    }
    // the 8 key slots of a wave (lanes 8 apart) are combined in registers; only one partial row per wave crosses the LDS (round 4: 6 KB
    // instead of 48 KB of partials -- the kernel is a pure K / V stream and ran two workgroups per CU, 3.1 TB/s)
#pragma unroll
    for (int h = 0; h < HG; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[h][j];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 8) part[wave * (HG * HD) + h * HD + c * 8 + j] = v;
        }
    __syncthreads();
    for (int o = tid; o < HG * HD; o += 256) {
        const float v = (part[o] + part[HG * HD + o]) + (part[2 * HG * HD + o] + part[3 * HG * HD + o]);
        out[(size_t)b * ldo + h0 * HD + o] = f2bf(v / rsum[o / HD]);
    }
}

}  // namespace

// C ABI ------------------------------------------------------------------------------------------
// qkv [B*N, ldqkv] bf16 (q|k|v, un-rotated, bias already added); cos/sin [(N-1), 64] f32; out [B*N, ldo] bf16;
// lse [B*H, N] f32 or null.  Head dim fixed at 64 (both EVA02 towers).
static int attn_fwd_impl(const void* qkv, const float* cos_t, const float* sin_t, void* out, float* lse, float* stats_part, int B, int Ntok,
                         int H, int ldqkv, int ldo, float scale, hipStream_t stream) {
    if (check_common("cs_attn_fwd", B, Ntok, H, ldqkv, ldo)) return -1;
    AttnArgs a{};
    a.qkv = (const __bf16*)qkv; a.cos_t = cos_t; a.sin_t = sin_t; a.out = (__bf16*)out; a.lse_out = lse;
    a.stats_part = stats_part; a.Mtot = (long)B * Ntok;
    int g = (int)(sqrtf((float)(Ntok - 1)) + 0.5f);
    CS_CHECK_ARG(g * g == Ntok - 1, "cs_attn_fwd: Ntok - 1 = %d is not a square token grid (the RoPE tables are read separably)", Ntok - 1);
    a.grid = g; a.inv_grid = 1.f / (float)g;
#ifdef CS_ABLATION_SWITCHES
    static const int dbg_env = getenv("CS_ATTN_DBG") ? atoi(getenv("CS_ATTN_DBG")) : 0;
    a.dbg = dbg_env;
    static const char* trace_path = getenv("CS_ATTN_TRACE");
    static unsigned long long* trace_buf = nullptr;
    static size_t trace_cap = 0;
    if (trace_path && trace_cap < (size_t)B * H * 64) {
        if (trace_buf) (void)hipFree(trace_buf);
        trace_cap = (size_t)B * H * 64;
        if (hipMalloc((void**)&trace_buf, trace_cap) != hipSuccess) { trace_buf = nullptr; trace_cap = 0; }
    }
    a.trace = trace_buf;
#else
    a.dbg = 0;
#endif
    a.Ntok = Ntok; a.H = H; a.ldqkv = ldqkv; a.ldo = ldo; a.scale = scale;
    constexpr int CH = 7;
    static const size_t lds_pad = getenv("CS_ATTN_LDSPAD") ? (size_t)atoi(getenv("CS_ATTN_LDSPAD")) : 0;     // occupancy experiments
    const size_t lds = (size_t)CH * 32 * 128 + (size_t)HD * VT_LD * 2 + (size_t)4 * g * 32 * sizeof(float) + lds_pad;
    CS_CHECK_ARG(lds <= 160 * 1024, "cs_attn_fwd: token grid %d too large for the LDS RoPE tables", g);
    // A/B switch, read per launch: CS_ATTN_FWD4=1 = four waves per unit, three units per CU (round 6: a tie, see attn_fwd4_kernel)
    const char* f4 = getenv("CS_ATTN_FWD4");
    if (f4 && f4[0] == '1' && Ntok <= K4_ROWS && g <= 14) {
        static bool once4 = (set_lds(attn_fwd4_kernel<4, false>, 160 * 1024), set_lds(attn_fwd4_kernel<4, true>, 160 * 1024), true);
        (void)once4;
        const size_t lds4 = (size_t)K4_ROWS * 128 + (size_t)HD * VT4_LD * 2 + (size_t)2 * g * 16 * sizeof(float) + lds_pad;
        if (getenv("CS_ATTN_DEBUG")) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, attn_fwd4_kernel<4, true>, 256, lds4);
            fprintf(stderr, "[cs_attn] fwd4: %d resident workgroups per CU (lds %zu)\n", nb, lds4);
        }
        if (Ntok > 192) hipLaunchKernelGGL((attn_fwd4_kernel<4, true>), dim3(1, B * H), dim3(256), lds4, stream, a);
        else hipLaunchKernelGGL((attn_fwd4_kernel<4, false>), dim3(1, B * H), dim3(256), lds4, stream, a);
    } else if (Ntok <= CH * 32) {
        // whole sequence in one LDS image: eight waves, one query tile each, two workgroups (16 waves) per CU; when only the last key tile is
        // ragged (Ntok > 32 (CH - 1): the 14x14 grid), the variant whose ragged-tile code exists once
        static bool once = (set_lds(attn_fwd8_kernel<false, 0>, 160 * 1024), set_lds(attn_fwd8_kernel<true, 0>, 160 * 1024),
                            set_lds(attn_fwd8_kernel<false, 1>, 160 * 1024), set_lds(attn_fwd8_kernel<true, 1>, 160 * 1024), true);
        (void)once;
        // A/B switch, read per launch.  Default: the transposed V image of rounds 1-4.  CS_ATTN_FWD8_VROW=1: V row-major + transposing reads --
        // bit-identical outputs, measured 2 % SLOWER per 2048-crop launch (811-816 -> 830-831 us, profiles/r05_h_fwd8_row_major_v.txt): the
        // 8-byte transposing reads double the V fragment instructions of the attend phase and meet 2-way bank conflicts, which costs more than
        // the in-register transposes of the staging phase save.
        const bool vt = getenv("CS_ATTN_FWD8_VROW") == nullptr;
        const size_t lds8 = vt ? lds : (size_t)2 * CH * 32 * 128 + (size_t)4 * g * 32 * sizeof(float) + lds_pad;
        if (getenv("CS_ATTN_DEBUG")) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, attn_fwd8_kernel<true, 0>, 512, lds8);
            fprintf(stderr, "[cs_attn] fwd8: %d resident workgroups per CU (lds %zu)\n", nb, lds8);
        }
        const bool tail = Ntok > (CH - 1) * 32;
        if (vt) {
            if (tail) hipLaunchKernelGGL((attn_fwd8_kernel<true, 0>), dim3(1, B * H), dim3(512), lds8, stream, a);
            else hipLaunchKernelGGL((attn_fwd8_kernel<false, 0>), dim3(1, B * H), dim3(512), lds8, stream, a);
        } else {
            if (tail) hipLaunchKernelGGL((attn_fwd8_kernel<true, 1>), dim3(1, B * H), dim3(512), lds8, stream, a);
            else hipLaunchKernelGGL((attn_fwd8_kernel<false, 1>), dim3(1, B * H), dim3(512), lds8, stream, a);
        }
    } else {
        static bool once = (set_lds(attn_fwd_kernel<CH>, 160 * 1024), set_lds(attn_fwd2_kernel<CH, 4>, 160 * 1024), true);
        (void)once;
        if (getenv("CS_ATTN_FWD_V1")) {                    // A/B switch, read per launch: the synchronous round-1 form
            hipLaunchKernelGGL((attn_fwd_kernel<CH>), dim3((Ntok + 255) / 256, B * H), dim3(512), lds, stream, a);
        } else {
            const size_t lds2 = (size_t)2 * CH * 32 * 128 + (size_t)4 * g * 32 * sizeof(float);
            dim3 grid2;
            set_schedule(a, B, Ntok, H, grid2);
            hipLaunchKernelGGL((attn_fwd2_kernel<CH, 4>), grid2, dim3(512), lds2, stream, a);
        }
    }
    CS_LAUNCH_CHECK();
#ifdef CS_ABLATION_SWITCHES
    if (a.trace && Ntok <= CH * 32) {     // the LAST launch's timeline survives in the file
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> host((size_t)B * H * 8);
        (void)hipMemcpy(host.data(), a.trace, host.size() * 8, hipMemcpyDeviceToHost);
        if (FILE* f = fopen(trace_path, "wb")) { fwrite(host.data(), 8, host.size(), f); fclose(f); }
    }
#endif
    return 0;
}

extern "C" int cs_attn_fwd(const void* qkv, const float* cos_t, const float* sin_t, void* out, float* lse, int B, int Ntok, int H,
                           int ldqkv, int ldo, float scale, hipStream_t stream) {
    return attn_fwd_impl(qkv, cos_t, sin_t, out, lse, nullptr, B, Ntok, H, ldqkv, ldo, scale, stream);
}

// cs_attn_fwd that also emits stats_part [H][B*Ntok][2] f32: per head, (sum, sum of squares) of each output row's 64 values
// (after rounding to bf16) -- the LayerNorm statistics of inner_attn_ln without another pass (see cs_ln_stats_finalize).
extern "C" int cs_attn_fwd_stats(const void* qkv, const float* cos_t, const float* sin_t, void* out, float* lse, float* stats_part, int B,
                                 int Ntok, int H, int ldqkv, int ldo, float scale, hipStream_t stream) {
    CS_CHECK_ARG(stats_part != nullptr && ((uintptr_t)stats_part % 8) == 0, "cs_attn_fwd_stats: stats_part must be an 8-byte aligned buffer");
    return attn_fwd_impl(qkv, cos_t, sin_t, out, lse, stats_part, B, Ntok, H, ldqkv, ldo, scale, stream);
}

// dsum [B*H, Ntok] f32, and for sequences of more than one key chunk the rotated q | k of the launch ([B*Ntok, 2C] bf16, 256-byte aligned)
static size_t bwd_dsum_bytes(int B, int Ntok, int H) { return (((size_t)B * H * Ntok * sizeof(float)) + 255) & ~(size_t)255; }
extern "C" size_t cs_attn_bwd_workspace(int B, int Ntok, int H) {
    return bwd_dsum_bytes(B, Ntok, H) + (Ntok > 7 * 32 ? (size_t)B * Ntok * 2 * H * HD * sizeof(__bf16) : 0);
}

// o, dout [B*N, ldo] bf16; lse from the forward; dqkv [B*N, ldqkv] bf16 receives d(q|k|v) w.r.t. the *un-rotated* q,k.
extern "C" int cs_attn_bwd(const void* qkv, const void* o, const void* dout, const float* lse, const float* cos_t, const float* sin_t,
                           void* dqkv, void* workspace, int B, int Ntok, int H, int ldqkv, int ldo, float scale, hipStream_t stream) {
    if (check_common("cs_attn_bwd", B, Ntok, H, ldqkv, ldo)) return -1;
    CS_CHECK_ARG(workspace != nullptr && lse != nullptr, "cs_attn_bwd: workspace and lse are required");
    float* dsum = (float*)workspace;
    const long total = (long)B * Ntok * H * 8;
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, stream, (const __bf16*)o, (const __bf16*)dout, dsum, B, Ntok, H, ldo);
    CS_LAUNCH_CHECK();
    AttnArgs a{};
    a.qkv = (const __bf16*)qkv; a.dout = (const __bf16*)dout; a.cos_t = cos_t; a.sin_t = sin_t; a.lse_in = lse; a.dsum = dsum;
    a.out = (__bf16*)dqkv; a.Ntok = Ntok; a.H = H; a.ldqkv = ldqkv; a.ldo = ldo; a.scale = scale;
    dim3 grid((Ntok + 255) / 256, B * H), block(512);
    constexpr int CH = 7;
    constexpr int CHK = CH * 32, VLD = CHK + 4;
    const int g = (int)(sqrtf((float)(Ntok - 1)) + 0.5f);
    // the round-1 kernels (transposed images, RoPE from the full global tables): the A/B switch, read per launch -- and the form that serves a
    // token count that is not a square grid + 1 (the re-staged kernels read the tables separably, row part | column part)
    const bool v1 = getenv("CS_ATTN_BWD_V1") != nullptr || g * g != Ntok - 1;
    if (!v1) {
        a.grid = g; a.inv_grid = 1.f / (float)g;
        const size_t rope = (size_t)4 * g * 32 * sizeof(float);
        const size_t lds_dq = (size_t)2 * CHK * 128 + rope, lds_dkv = (size_t)2 * CHK * 128 + (size_t)2 * CHK * sizeof(float) + rope;
        CS_CHECK_ARG(lds_dkv <= 160 * 1024, "cs_attn_bwd: token grid %d too large for the LDS RoPE tables", g);
        static bool once = (set_lds(attn_bwd_dq2_kernel<CH, false>, 160 * 1024), set_lds(attn_bwd_dkv2_kernel<CH, false>, 160 * 1024),
                            set_lds(attn_bwd_dq2_kernel<CH, true>, 160 * 1024), set_lds(attn_bwd_dkv2_kernel<CH, true>, 160 * 1024),
                            set_lds(attn_bwd_dq2_kernel<CH, false, true>, 160 * 1024), set_lds(attn_bwd_dkv2_kernel<CH, false, true>, 160 * 1024), true);
        (void)once;
        if (Ntok <= CHK) {
            hipLaunchKernelGGL((attn_bwd_dq2_kernel<CH, true>), grid, block, lds_dq, stream, a);
            CS_LAUNCH_CHECK();
            hipLaunchKernelGGL((attn_bwd_dkv2_kernel<CH, true>), grid, block, lds_dkv, stream, a);
        } else {
            dim3 grid2;
            set_schedule(a, B, Ntok, H, grid2);
            if (getenv("CS_ATTN_NOPRE")) {                 // A/B switch, read per launch: every row block rotates its operands itself (round 5)
                hipLaunchKernelGGL((attn_bwd_dq2_kernel<CH, false>), grid2, block, lds_dq, stream, a);
                CS_LAUNCH_CHECK();
                hipLaunchKernelGGL((attn_bwd_dkv2_kernel<CH, false>), grid2, block, lds_dkv, stream, a);
            } else {
                const int C2 = 2 * H * HD;
                __bf16* qk = (__bf16*)((char*)workspace + bwd_dsum_bytes(B, Ntok, H));
                const long items = (long)B * Ntok * (C2 / 8);
                hipLaunchKernelGGL(rope_qk_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, (const __bf16*)qkv, cos_t, sin_t, qk,
                                   (long)B * Ntok, Ntok, H * HD, ldqkv);
                CS_LAUNCH_CHECK();
                a.qk = qk; a.ldqk = C2;
                hipLaunchKernelGGL((attn_bwd_dq2_kernel<CH, false, true>), grid2, block, lds_dq - rope, stream, a);
                CS_LAUNCH_CHECK();
                hipLaunchKernelGGL((attn_bwd_dkv2_kernel<CH, false, true>), grid2, block, lds_dkv - rope, stream, a);
            }
        }
        CS_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds_dq = (size_t)2 * CHK * 128 + (size_t)HD * VLD * 2;
    const size_t lds_dkv = (size_t)2 * CHK * 128 + (size_t)2 * HD * VLD * 2 + (size_t)2 * CHK * sizeof(float);
    static bool once = (set_lds(attn_bwd_dq_kernel<CH>, lds_dq), set_lds(attn_bwd_dkv_kernel<CH>, lds_dkv), true);
    (void)once;
    hipLaunchKernelGGL((attn_bwd_dq_kernel<CH>), grid, block, lds_dq, stream, a);
    CS_LAUNCH_CHECK();
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<CH>), grid, block, lds_dkv, stream, a);
    CS_LAUNCH_CHECK();
    return 0;
}

static int attn_cls_launch(const void* q, const void* kv, const float* cos_t, const float* sin_t, void* out, int rows, int Ntok, int H,
                           int ldq, int ldkv, int ldo, float scale, int qpb, const unsigned char* allow, hipStream_t stream) {
    const int HG = H % 6 == 0 ? 6 : (H % 4 == 0 ? 4 : (H % 3 == 0 ? 3 : (H % 2 == 0 ? 2 : 1)));
    const size_t lds = ((size_t)HG * ((Ntok + 3) & ~3) + 8 + (size_t)4 * HG * HD) * sizeof(float);
    CS_CHECK_ARG(lds <= 160 * 1024, "cs_attn_cls_fwd: Ntok=%d too large", Ntok);
    const dim3 grid(rows, H / HG), block(256);
#define CS_CLS_LAUNCH(G)                                                                                                      \
    {                                                                                                                         \
        static bool once = (set_lds(attn_cls_kernel<G>, 160 * 1024), true);                                                   \
        (void)once;                                                                                                           \
        hipLaunchKernelGGL(attn_cls_kernel<G>, grid, block, lds, stream, (const __bf16*)q, (const __bf16*)kv, cos_t, sin_t,   \
                           (__bf16*)out, Ntok, H, ldq, ldkv, ldo, scale, qpb, allow);                                         \
    }
    switch (HG) {
        case 6: CS_CLS_LAUNCH(6) break;
        case 4: CS_CLS_LAUNCH(4) break;
        case 3: CS_CLS_LAUNCH(3) break;
        case 2: CS_CLS_LAUNCH(2) break;
        default: CS_CLS_LAUNCH(1) break;
    }
#undef CS_CLS_LAUNCH
    CS_LAUNCH_CHECK();
    return 0;
}

// q [B, ldq] bf16: the CLS-token queries (H*64 wide, bias added; token 0 is never rotated); kv [B*N, ldkv] bf16 = k|v
// (un-rotated, bias added); out [B, ldo] bf16 = the CLS row of softmax(q k^T * scale) v.
extern "C" int cs_attn_cls_fwd(const void* q, const void* kv, const float* cos_t, const float* sin_t, void* out, int B, int Ntok, int H,
                               int ldq, int ldkv, int ldo, float scale, hipStream_t stream) {
    CS_CHECK_ARG(q && kv && cos_t && sin_t && out, "cs_attn_cls_fwd: null pointer");
    CS_CHECK_ARG(B > 0 && Ntok > 1 && H > 0, "cs_attn_cls_fwd: bad sizes B=%d Ntok=%d H=%d", B, Ntok, H);
    CS_CHECK_ARG(ldq % 8 == 0 && ldkv % 8 == 0 && ldq >= H * HD && ldkv >= 2 * H * HD && ldo >= H * HD,
                 "cs_attn_cls_fwd: row strides must be multiples of 8 and cover the heads (ldq=%d ldkv=%d ldo=%d)", ldq, ldkv, ldo);
    CS_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)kv % 16) == 0, "cs_attn_cls_fwd: q/kv must be 16-byte aligned");
    return attn_cls_launch(q, kv, cos_t, sin_t, out, B, Ntok, H, ldq, ldkv, ldo, scale, 1, nullptr, stream);
}

// Extra query tokens of the OpenAI-CLIP family's mask-attention pooling (open_clip/transformer.py:736-834, reached through
// extract_type='v1' :660-671 and encode_masks(mask_attn=True), model.py:245-247): Q query rows per image against the image's own keys /
// values of the same depth, key j of query row r allowed iff allow[r * Ntok + j] != 0 (key 0 = the CLS token, always allowed there).
// q [B*Q, ldq] bf16; kv [B*Ntok, ldkv] bf16 = k|v; out [B*Q, ldo] bf16.  No rotary embedding in this family.  Inference only.
extern "C" int cs_attn_query_fwd(const void* q, const void* kv, const unsigned char* allow, void* out, int B, int Q, int Ntok, int H,
                                 int ldq, int ldkv, int ldo, float scale, hipStream_t stream) {
    CS_CHECK_ARG(q && kv && allow && out, "cs_attn_query_fwd: null pointer");
    CS_CHECK_ARG(B > 0 && Q > 0 && Ntok > 1 && H > 0, "cs_attn_query_fwd: bad sizes B=%d Q=%d Ntok=%d H=%d", B, Q, Ntok, H);
    CS_CHECK_ARG(ldq % 8 == 0 && ldkv % 8 == 0 && ldq >= H * HD && ldkv >= 2 * H * HD && ldo >= H * HD,
                 "cs_attn_query_fwd: row strides must be multiples of 8 and cover the heads (ldq=%d ldkv=%d ldo=%d)", ldq, ldkv, ldo);
    CS_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)kv % 16) == 0, "cs_attn_query_fwd: q/kv must be 16-byte aligned");
    return attn_cls_launch(q, kv, nullptr, nullptr, out, B * Q, Ntok, H, ldq, ldkv, ldo, scale, Q, allow, stream);
}
