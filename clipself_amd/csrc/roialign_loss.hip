// RoIAlign(1x1, adaptive sampling, aligned) over the NHWC token map + row-wise cosine distillation loss (fwd/bwd).
//
// Reference call sites:
//   src/open_clip/eva_clip/eva_vit_model.py:625-629,655-664   roi_align(x, boxes*[w,h], (1,1), 1.0, -1, True)[...,0,0]
//   src/training/clipself.py:42-47                              F.normalize both, 1 - mean(sum(s*t))
// The RoIAlign arithmetic itself is torchvision's (absent wheel); algorithm restated in oracle/roi_align_ref.py.
//
// MI355X layout: the student's token map stays token-major [B, Ntok, E] fp32 exactly as the head GEMM wrote it
// (channels contiguous -> every sample row is a coalesced E*4-byte read); there is no NCHW view and no
// .contiguous() copy.  Bilinear weights are separable, so each box first folds its gh x gw sample grid into
// per-row / per-column weights Wy[h], Wx[w] (sequentially, fixed order -> deterministic) and then touches each
// map cell at most once:   pooled = sum_{r,c} Wy[r] Wx[c] F[r,c] / max(gh*gw,1).
#include "cs_common.h"

namespace {

constexpr int MAXGRID = 128;   // token grid side limit (1024/8)

// torchvision's per-axis sample handling, float32 like its T=float path
__device__ void axis_weights(float start, float extent, int grid, int size, float* w /*[size]*/) {
    for (int i = 0; i < size; ++i) w[i] = 0.f;
    for (int i = 0; i < grid; ++i) {
        // un-contracted float32 steps (no FMA), same order as torchvision / the numpy oracle
        float c = __fadd_rn(start, __fdiv_rn(__fmul_rn((float)i + 0.5f, extent), (float)grid));
        if (c < -1.0f || c > (float)size) continue;
        if (c <= 0.f) c = 0.f;
        int lo = (int)c, hi;
        if (lo >= size - 1) { hi = lo = size - 1; c = (float)lo; } else { hi = lo + 1; }
        const float l = __fsub_rn(c, (float)lo);
        w[lo] += __fsub_rn(1.f, l);
        w[hi] += l;
    }
}

struct RoiGeom { int b, gh, gw; float inv_count; };

__device__ __forceinline__ RoiGeom box_setup(const float* __restrict__ roi, int gh_map, int gw_map, float* Wy, float* Wx, int tid) {
    // roi = (batch, x0, y0, x1, y1) with coordinates normalised to [0,1]; _denormalize_boxes multiplies by (w, h)
    const float x0 = __fsub_rn(__fmul_rn(roi[1], (float)gw_map), 0.5f), y0 = __fsub_rn(__fmul_rn(roi[2], (float)gh_map), 0.5f);
    const float x1 = __fsub_rn(__fmul_rn(roi[3], (float)gw_map), 0.5f), y1 = __fsub_rn(__fmul_rn(roi[4], (float)gh_map), 0.5f);
    const float rw = __fsub_rn(x1, x0), rh = __fsub_rn(y1, y0);
    RoiGeom g;
    g.b = (int)roi[0];
    g.gh = (int)ceilf(rh);
    g.gw = (int)ceilf(rw);
    const int cnt = g.gh * g.gw;
    g.inv_count = 1.f / (float)(cnt > 1 ? cnt : 1);
    if (tid == 0) axis_weights(y0, rh, g.gh > 0 ? g.gh : 0, gh_map, Wy);
    if (tid == 64) axis_weights(x0, rw, g.gw > 0 ? g.gw : 0, gw_map, Wx);
    return g;
}

// feat [B, Ntok, E] f32 (token 0 = CLS, skipped), rois [K,5], pooled [K,E] f32
__global__ __launch_bounds__(256) void roialign_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ rois,
                                                           float* __restrict__ pooled, int Ntok, int gh_map, int gw_map, int E, int tok_off) {
    __shared__ float Wy[MAXGRID], Wx[MAXGRID];
    const int k = blockIdx.x, tid = threadIdx.x;
    const RoiGeom g = box_setup(rois + (size_t)k * 5, gh_map, gw_map, Wy, Wx, tid);
    __syncthreads();
    const bool empty = g.gh <= 0 || g.gw <= 0;
    const float* fb = feat + ((size_t)g.b * Ntok + tok_off) * E;
    // the cells a box touches are a sub-rectangle of the map: walk that, not the whole map (the recipe's 64 x 64 grid: 4096 cells per box and
    // thread, ~250 of them with a weight; zero weights inside the range are still skipped, so the additions and their order are unchanged)
    int r_lo = 0, r_hi = gh_map, c_lo = 0, c_hi = gw_map;
    while (r_lo < r_hi && Wy[r_lo] == 0.f) ++r_lo;
    while (r_hi > r_lo && Wy[r_hi - 1] == 0.f) --r_hi;
    while (c_lo < c_hi && Wx[c_lo] == 0.f) ++c_lo;
    while (c_hi > c_lo && Wx[c_hi - 1] == 0.f) --c_hi;
    for (int ch = tid * 4; ch < E; ch += 1024) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!empty) {
            for (int r = r_lo; r < r_hi; ++r) {
                const float wy = Wy[r];
                if (wy == 0.f) continue;
                for (int c = c_lo; c < c_hi; ++c) {
                    const float w = wy * Wx[c];
                    if (w == 0.f) continue;
                    const float4 v = *(const float4*)(fb + (size_t)(r * gw_map + c) * E + ch);
                    acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
                }
            }
        }
        *(float4*)(pooled + (size_t)k * E + ch) = make_float4(acc.x * g.inv_count, acc.y * g.inv_count, acc.z * g.inv_count, acc.w * g.inv_count);
    }
}

// Weight of map cell `cell` along one axis: the same samples, in the same order, as axis_weights() adds into w[cell] -> bit-identical to
// the forward's Wy[cell] / Wx[cell].
__device__ float axis_weight_at(float start, float extent, int grid, int size, int cell) {
    float w = 0.f;
    for (int i = 0; i < grid; ++i) {
        float c = __fadd_rn(start, __fdiv_rn(__fmul_rn((float)i + 0.5f, extent), (float)grid));
        if (c < -1.0f || c > (float)size) continue;
        if (c <= 0.f) c = 0.f;
        int lo = (int)c, hi;
        if (lo >= size - 1) { hi = lo = size - 1; c = (float)lo; } else { hi = lo + 1; }
        const float l = __fsub_rn(c, (float)lo);
        if (lo == cell) w += __fsub_rn(1.f, l);
        if (hi == cell) w += l;
    }
    return w;
}

// Backward as a GATHER: one workgroup owns RB_CELLS consecutive cells of one grid row of one image and adds up, box by box in ascending
// box order, what every box of that image sends to them:  dfeat[b, r, c, :] += sum_k Wy_k[r] Wx_k[c] dpooled[k, :] / count_k.
// No atomics, a fixed summation order -> the gradient is bit-reproducible (torchvision's backward, and round 2's scatter kernel, add with
// float atomics in arrival order: 2.8e-4 relative run-to-run differences on the step's gradients).
constexpr int RB_CELLS = 16, RB_EPT = 4;          // cells per workgroup; channels per thread (256 * RB_EPT channels per workgroup, wider maps: more workgroups)
__global__ __launch_bounds__(256) void roialign_bwd_gather_kernel(const float* __restrict__ dpooled, const float* __restrict__ rois,
                                                                  float* __restrict__ dfeat, int K, int Ntok, int gh_map, int gw_map, int E,
                                                                  int tok_off, int cell_groups) {
    __shared__ int list[256];
    __shared__ int wave_cnt[4];
    __shared__ float Wxs[RB_CELLS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cg = blockIdx.x % cell_groups, ch0 = (blockIdx.x / cell_groups) * (256 * RB_EPT);      // cell group, first channel of this workgroup
    const int c0 = cg * RB_CELLS, r = blockIdx.y, b = blockIdx.z;
    float acc[RB_CELLS][RB_EPT];
#pragma unroll
    for (int c = 0; c < RB_CELLS; ++c)
#pragma unroll
        for (int j = 0; j < RB_EPT; ++j) acc[c][j] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 256) {
        // ordered compaction of the boxes k0 .. k0+255 that belong to image b
        const int k = k0 + tid;
        const bool mine = k < K && (int)rois[(size_t)k * 5] == b;
        const unsigned long long m = __ballot(mine);
        if (lane == 0) wave_cnt[wv] = __popcll(m);
        __syncthreads();
        int base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wv) base += wave_cnt[w];
            total += wave_cnt[w];
        }
        if (mine) list[base + __popcll(m & ((1ull << lane) - 1ull))] = k;
        __syncthreads();
        for (int n = 0; n < total; ++n) {                  // workgroup-uniform control flow below: every value comes from the box record
            const float* roi = rois + (size_t)list[n] * 5;
            const float x0 = __fsub_rn(__fmul_rn(roi[1], (float)gw_map), 0.5f), y0 = __fsub_rn(__fmul_rn(roi[2], (float)gh_map), 0.5f);
            const float x1 = __fsub_rn(__fmul_rn(roi[3], (float)gw_map), 0.5f), y1 = __fsub_rn(__fmul_rn(roi[4], (float)gh_map), 0.5f);
            const float rw = __fsub_rn(x1, x0), rh = __fsub_rn(y1, y0);
            const int gh = (int)ceilf(rh), gw = (int)ceilf(rw);
            if (gh <= 0 || gw <= 0) continue;
            const float wy = axis_weight_at(y0, rh, gh, gh_map, r);
            if (wy == 0.f) continue;
            const int cnt = gh * gw;
            const float inv_count = 1.f / (float)(cnt > 1 ? cnt : 1);
            __syncthreads();                               // the previous box's Wxs has been consumed
            if (tid < RB_CELLS) Wxs[tid] = c0 + tid < gw_map ? axis_weight_at(x0, rw, gw, gw_map, c0 + tid) : 0.f;
            __syncthreads();
            float gv[RB_EPT];
#pragma unroll
            for (int j = 0; j < RB_EPT; ++j) {
                const int ch = ch0 + tid + 256 * j;
                gv[j] = ch < E ? dpooled[(size_t)list[n] * E + ch] * inv_count : 0.f;
            }
#pragma unroll
            for (int c = 0; c < RB_CELLS; ++c) {
                const float w = wy * Wxs[c];
                if (w != 0.f) {
#pragma unroll
                    for (int j = 0; j < RB_EPT; ++j) acc[c][j] += w * gv[j];
                }
            }
        }
        __syncthreads();                                   // list / wave_cnt are rewritten by the next chunk
    }
#pragma unroll
    for (int c = 0; c < RB_CELLS; ++c) {
        float* cell = dfeat + ((size_t)b * Ntok + tok_off + (size_t)r * gw_map + min(c0 + c, gw_map - 1)) * E;
#pragma unroll
        for (int j = 0; j < RB_EPT; ++j) {
            const int ch = ch0 + tid + 256 * j;
            if (c0 + c < gw_map && ch < E && acc[c][j] != 0.f) cell[ch] += acc[c][j];
        }
    }
}

// ---- cosine loss ----------------------------------------------------------------------------
// per box: cos_k = <s/|s|, t/|t|>  (F.normalize eps 1e-12).  stats[k] = (cos, 1/max(|s|,eps), 1/max(|t|,eps))
__global__ __launch_bounds__(256) void cosine_rows_kernel(const float* __restrict__ s, const float* __restrict__ t, float* __restrict__ stats,
                                                          int K, int E) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= K) return;
    float ss = 0.f, tt = 0.f, st = 0.f;
    for (int c = lane * 4; c < E; c += 256) {
        const float4 a = *(const float4*)(s + (size_t)k * E + c), b = *(const float4*)(t + (size_t)k * E + c);
        ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
        tt += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
        st += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    ss = wave_sum(ss); tt = wave_sum(tt); st = wave_sum(st);
    const float is = 1.f / fmaxf(sqrtf(ss), 1e-12f), it = 1.f / fmaxf(sqrtf(tt), 1e-12f);
    if (lane == 0) { stats[k * 3] = st * is * it; stats[k * 3 + 1] = is; stats[k * 3 + 2] = it; }
}

// loss = weight * (1 - mean_k cos_k); single workgroup, fixed-order tree -> deterministic
__global__ __launch_bounds__(256) void cosine_reduce_kernel(const float* __restrict__ stats, float* __restrict__ loss, int K, float weight) {
    __shared__ float part[256];
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) s += stats[k * 3];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = weight * (1.f - part[0] / (float)K);
}

// d loss / d s_k = -(weight*gscale/K) * (t_hat - cos_k * s_hat) / |s|
__global__ __launch_bounds__(256) void cosine_bwd_kernel(const float* __restrict__ s, const float* __restrict__ t, const float* __restrict__ stats,
                                                         float* __restrict__ ds, int K, int E, float coef,
                                                         const float* __restrict__ upstream) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)K * E) return;
    if (upstream) coef *= upstream[0];        // d(total)/d(loss) left on the device: no host sync in backward
    const int k = (int)(i / E);
    const float cosv = stats[k * 3], is = stats[k * 3 + 1], it = stats[k * 3 + 2];
    ds[i] = coef * (t[i] * it - cosv * s[i] * is) * is;
}

}  // namespace

// C ABI ------------------------------------------------------------------------------------------
// feat: token-major map [B, Ntok, E] f32; the grid occupies tokens tok_off .. tok_off+gh*gw-1 (tok_off = 1 skips CLS).
// rois [K,5] f32 = (image index, x0, y0, x1, y1) with box coordinates normalised to [0,1] (reference batch contract).
extern "C" int cs_roialign_fwd(const float* feat, const float* rois, float* pooled, int K, int Ntok, int grid_h, int grid_w, int E,
                               int tok_off, hipStream_t stream) {
    CS_CHECK_ARG(grid_h > 0 && grid_w > 0 && grid_h <= MAXGRID && grid_w <= MAXGRID && E % 4 == 0, "cs_roialign_fwd: bad grid/E");
    CS_CHECK_ARG(tok_off + grid_h * grid_w <= Ntok, "cs_roialign_fwd: grid does not fit the token map");
    if (K == 0) return 0;
    hipLaunchKernelGGL(roialign_fwd_kernel, dim3(K), dim3(256), 0, stream, feat, rois, pooled, Ntok, grid_h, grid_w, E, tok_off);
    CS_LAUNCH_CHECK();
    return 0;
}
// dfeat [B, Ntok, E] f32 += the gradient of every box (the caller zeroes it, or accumulates); B = images in the map.  Deterministic: a
// gather per map cell over the image's boxes in ascending box order, no atomics.
extern "C" int cs_roialign_bwd(const float* dpooled, const float* rois, float* dfeat, int K, int B, int Ntok, int grid_h, int grid_w, int E,
                               int tok_off, hipStream_t stream) {
    CS_CHECK_ARG(grid_h > 0 && grid_w > 0 && grid_h <= MAXGRID && grid_w <= MAXGRID, "cs_roialign_bwd: bad grid");
    CS_CHECK_ARG(tok_off + grid_h * grid_w <= Ntok, "cs_roialign_bwd: grid does not fit the token map");
    CS_CHECK_ARG(B > 0 && B <= 65535 && E > 0, "cs_roialign_bwd: B=%d images (1..65535), E=%d channels", B, E);
    if (K == 0) return 0;
    const int cell_groups = (grid_w + RB_CELLS - 1) / RB_CELLS, chunks = (E + 256 * RB_EPT - 1) / (256 * RB_EPT);     // E > 1024 (bigG-class heads): channel chunks
    hipLaunchKernelGGL(roialign_bwd_gather_kernel, dim3(cell_groups * chunks, grid_h, B), dim3(256), 0, stream, dpooled, rois,
                       dfeat, K, Ntok, grid_h, grid_w, E, tok_off, cell_groups);
    CS_LAUNCH_CHECK();
    return 0;
}
// stats: workspace [K,3] f32 kept for the backward.
extern "C" int cs_cosine_loss_fwd(const float* student, const float* teacher, float* stats, float* loss, int K, int E, float weight,
                                  hipStream_t stream) {
    CS_CHECK_ARG(K > 0 && E % 4 == 0, "cs_cosine_loss_fwd: need K > 0 and E %% 4 == 0");
    hipLaunchKernelGGL(cosine_rows_kernel, dim3((K + 3) / 4), dim3(256), 0, stream, student, teacher, stats, K, E);
    CS_LAUNCH_CHECK();
    hipLaunchKernelGGL(cosine_reduce_kernel, dim3(1), dim3(256), 0, stream, stats, loss, K, weight);
    CS_LAUNCH_CHECK();
    return 0;
}
extern "C" int cs_cosine_loss_bwd(const float* student, const float* teacher, const float* stats, float* dstudent, int K, int E,
                                  float weight, float grad_scale, const float* upstream, hipStream_t stream) {
    CS_CHECK_ARG(K > 0, "cs_cosine_loss_bwd: K must be positive");
    const float coef = -weight * grad_scale / (float)K;
    hipLaunchKernelGGL(cosine_bwd_kernel, dim3((int)(((long)K * E + 255) / 256)), dim3(256), 0, stream, student, teacher, stats, dstudent, K, E, coef, upstream);
    CS_LAUNCH_CHECK();
    return 0;
}

// ---- RegionCLIP federated BCE (src/training/region_clip.py:47-56) --------------------------------------------
// z = logits * temp over the `ns` sampled noun columns; target = one-hot at tgt[k] (column index inside the sample,
// -1 = none); row loss = sum_c [max(z,0) - z*t + log(1 + exp(-|z|))]; loss = weight * mean_k row loss.
namespace {

__global__ __launch_bounds__(256) void fed_bce_rows_kernel(const float* __restrict__ logits, long ldz, const int* __restrict__ tgt,
                                                           float* __restrict__ rowloss, int K, int ns, float temp) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= K) return;
    const int t = tgt[k];
    float s = 0.f;
    for (int c = lane; c < ns; c += 64) {
        const float z = logits[(size_t)k * ldz + c] * temp;
        s += fmaxf(z, 0.f) - (c == t ? z : 0.f) + log1pf(__expf(-fabsf(z)));
    }
    s = wave_sum(s);
    if (lane == 0) rowloss[k] = s;
}

__global__ __launch_bounds__(256) void fed_bce_reduce_kernel(const float* __restrict__ rowloss, float* __restrict__ loss, int K, float weight) {
    __shared__ float part[256];
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) s += rowloss[k];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = weight * part[0] / (float)K;
}

// dz[k,c] = coef * (sigmoid(z) - t) * temp  for c < ns, 0 for the padding columns [ns, ldd)   (bf16: dgrad GEMM operand)
__global__ __launch_bounds__(256) void fed_bce_bwd_kernel(const float* __restrict__ logits, long ldz, const int* __restrict__ tgt,
                                                          __bf16* __restrict__ dz, long ldd, int K, int ns, float temp, float coef,
                                                          const float* __restrict__ upstream) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)K * ldd) return;
    const int k = (int)(i / ldd), c = (int)(i - (long)k * ldd);
    float v = 0.f;
    if (c < ns) {
        if (upstream) coef *= upstream[0];
        const float z = logits[(size_t)k * ldz + c] * temp;
        v = coef * temp * (1.f / (1.f + __expf(-z)) - (c == tgt[k] ? 1.f : 0.f));
    }
    dz[i] = f2bf(v);
}

}  // namespace

extern "C" int cs_fed_bce_fwd(const float* logits, long ldz, const int* tgt, float* rowloss, float* loss, int K, int ns, float temp,
                              float weight, hipStream_t stream) {
    CS_CHECK_ARG(K > 0 && ns > 0 && ldz >= ns, "cs_fed_bce_fwd: bad shape");
    hipLaunchKernelGGL(fed_bce_rows_kernel, dim3((K + 3) / 4), dim3(256), 0, stream, logits, ldz, tgt, rowloss, K, ns, temp);
    CS_LAUNCH_CHECK();
    hipLaunchKernelGGL(fed_bce_reduce_kernel, dim3(1), dim3(256), 0, stream, rowloss, loss, K, weight);
    CS_LAUNCH_CHECK();
    return 0;
}
extern "C" int cs_fed_bce_bwd(const float* logits, long ldz, const int* tgt, void* dz_bf16, long ldd, int K, int ns, float temp,
                              float weight, const float* upstream, hipStream_t stream) {
    CS_CHECK_ARG(K > 0 && ns > 0 && ldz >= ns && ldd >= ns, "cs_fed_bce_bwd: bad shape");
    hipLaunchKernelGGL(fed_bce_bwd_kernel, dim3((int)(((long)K * ldd + 255) / 256)), dim3(256), 0, stream, logits, ldz, tgt, (__bf16*)dz_bf16,
                       ldd, K, ns, temp, weight / (float)K, upstream);
    CS_LAUNCH_CHECK();
    return 0;
}
