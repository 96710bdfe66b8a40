// Region crops on the GPU: crop -> bicubic resize (longest side to S) -> zero pad -> /255 -> normalise, for the CLIPSelf
// teacher inputs.  Reference semantics: GridDistillDataset._obtain_image_crops (src/training/data.py:226-245)
// applies transforms[1] = [ResizeMaxSize(S, fill=0), ToTensor, Normalize] (src/open_clip/transform.py:26-49,93-99) to
// PIL `image.crop(box)`; the det image goes through [ResizeLongest(S), ToTensor, Normalize] (:169-191).  The arithmetic of the resize
// is Pillow's (third-party, not vendored by the reference): ImagingResample with the bicubic filter (a = -0.5), support scaled by
// the down-sampling factor, coefficients normalised in double precision and rounded to 22-bit fixed point, horizontal pass then
// vertical pass, each pass rounding to uint8.  This file restates that algorithm so that results are bit-identical to Pillow's
// (tests/test_gpu_ops.py compares with the Pillow installed in the image; oracle/pil_crops_ref.py drives it).
//
// One call handles the K boxes of ONE decoded image (HWC uint8, resident in HBM): pass 1 writes the horizontally resampled rows of
// every crop to a workspace, pass 2 resamples vertically, pads and normalises into out[K,3,S,S] fp32.
#include "cs_common.h"

// Pillow is compiled for baseline x86-64 (no FMA): keep every multiply and add separately rounded, as its C code executes them.
#pragma clang fp contract(off)

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ double bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0;
    if (x < 2.0) return (((x - 5.0) * x + 8.0) * x - 4.0) * a;
    return 0.0;
}

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// Python's round() on a float: round half to even = rint under the default rounding mode
__device__ __forceinline__ int py_round(double v) { return (int)rint(v); }

struct CropGeom {
    int x0, y0, cw, ch;      // integer crop rectangle (PIL Image.crop rounds the box)
    int nw, nh;              // resized size: longest side -> S
    int off_x, off_y;        // top-left of the resized crop inside the S x S canvas
};

__device__ __forceinline__ CropGeom geometry(const float* __restrict__ box, int W, int H, int S, int pad_center) {
    CropGeom g;
    const int x0 = py_round((double)box[0]), y0 = py_round((double)box[1]);
    const int x1 = py_round((double)box[2]), y1 = py_round((double)box[3]);
    g.x0 = x0; g.y0 = y0;
    g.cw = max(x1 - x0, 1); g.ch = min(max(y1 - y0, 1), H);       // the workspace holds H rows per crop
    const double scale = (double)S / (double)max(g.cw, g.ch);
    g.nw = max(py_round((double)g.cw * scale), 1);
    g.nh = max(py_round((double)g.ch * scale), 1);
    const int pw = S - g.nw, ph = S - g.nh;
    g.off_x = pad_center ? pw / 2 : 0;
    g.off_y = pad_center ? ph / 2 : 0;
    return g;
}

// Pillow's precompute_coeffs for one output position: taps [xmin, xmin+n) of an input of size in_size resampled to out_size,
// 22-bit fixed-point weights written to kk (at most MAXTAPS).
constexpr int MAXTAPS = 64;
__device__ __forceinline__ int coeffs(int in_size, int out_size, int xx, int& xmin, int* kk) {
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double center = ((double)xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    const int n = min(xmax - xmin, MAXTAPS);
    double k[MAXTAPS];
    double ww = 0.0;
    for (int x = 0; x < n; ++x) {
        const double w = bicubic(((double)(x + xmin) - center + 0.5) * ss);
        k[x] = w;
        ww += w;
    }
    for (int x = 0; x < n; ++x) {
        const double v = ww != 0.0 ? k[x] / ww : k[x];
        kk[x] = (int)(v < 0.0 ? -0.5 + v * (double)(1 << PRECISION_BITS) : 0.5 + v * (double)(1 << PRECISION_BITS));
    }
    return n;
}

// pass 1: tmp[k][y][x][c] (y < ch, x < nw) = horizontal resample of crop k's rows
__global__ __launch_bounds__(256) void crop_hpass_kernel(const uint8_t* __restrict__ src, int H, int W, const float* __restrict__ boxes,
                                                         int S, int pad_center, uint8_t* __restrict__ tmp, int tmp_rows) {
    const int k = blockIdx.z;
    const CropGeom g = geometry(boxes + 4 * k, W, H, S, pad_center);
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    if (x >= g.nw) return;
    int kk[MAXTAPS], xmin;
    const int n = coeffs(g.cw, g.nw, x, xmin, kk);
    for (int y = blockIdx.y * 4 + (threadIdx.x >> 6); y < g.ch; y += gridDim.y * 4) {
        const int sy = g.y0 + y;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        if (sy >= 0 && sy < H) {                              // PIL's crop fills what lies outside the image with zeros
            for (int t = 0; t < n; ++t) {
                const int sx = g.x0 + xmin + t;
                if (sx < 0 || sx >= W) continue;
                const uint8_t* px = src + ((size_t)sy * W + sx) * 3;
                s0 += px[0] * kk[t]; s1 += px[1] * kk[t]; s2 += px[2] * kk[t];
            }
        }
        uint8_t* o = tmp + (((size_t)k * tmp_rows + y) * S + x) * 3;
        o[0] = (uint8_t)clip8(s0 >> PRECISION_BITS); o[1] = (uint8_t)clip8(s1 >> PRECISION_BITS); o[2] = (uint8_t)clip8(s2 >> PRECISION_BITS);
    }
}

// pass 2: out[k][c][Y][X] = ((vertical resample | 0 in the padding) / 255 - mean[c]) / std[c]
__global__ __launch_bounds__(256) void crop_vpass_kernel(const uint8_t* __restrict__ tmp, int tmp_rows, int H, int W,
                                                         const float* __restrict__ boxes, int S, int pad_center, float m0, float m1, float m2,
                                                         float d0, float d1, float d2, float* __restrict__ out) {
    const int k = blockIdx.z;
    const CropGeom g = geometry(boxes + 4 * k, W, H, S, pad_center);
    const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X >= S || Y >= S) return;
    int v0 = 0, v1 = 0, v2 = 0;
    const int x = X - g.off_x, y = Y - g.off_y;
    if (x >= 0 && x < g.nw && y >= 0 && y < g.nh) {
        int kk[MAXTAPS], ymin;
        const int n = coeffs(g.ch, g.nh, y, ymin, kk);
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int t = 0; t < n; ++t) {
            const uint8_t* px = tmp + (((size_t)k * tmp_rows + ymin + t) * S + x) * 3;
            s0 += px[0] * kk[t]; s1 += px[1] * kk[t]; s2 += px[2] * kk[t];
        }
        v0 = clip8(s0 >> PRECISION_BITS); v1 = clip8(s1 >> PRECISION_BITS); v2 = clip8(s2 >> PRECISION_BITS);
    }
    const size_t plane = (size_t)S * S, o = (size_t)k * 3 * plane + (size_t)Y * S + X;
    out[o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v0, 255.f), m0), d0);
    out[o + plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v1, 255.f), m1), d1);
    out[o + 2 * plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v2, 255.f), m2), d2);
}

}  // namespace

extern "C" size_t cs_crop_resize_workspace(int H, int K, int S) { return (size_t)K * H * S * 3; }

// src [H,W,3] uint8 (decoded RGB image in HBM); boxes [K,4] f32 = (x0,y0,x1,y1) in pixels of src (rounded like PIL's Image.crop);
// out [K,3,S,S] f32.  pad_center 1 = ResizeMaxSize (crop transform: centred), 0 = ResizeLongest (det transform: right/bottom pad).
// A box covering the whole image with pad_center 0 is the det-image transform itself.  Down-sampling factors up to 15x.
extern "C" int cs_crop_resize_u8(const void* src, int H, int W, const float* boxes, int K, int S, int pad_center, const float* mean3,
                                 const float* std3, float* out, void* workspace, hipStream_t stream) {
    CS_CHECK_ARG(src && boxes && out && workspace && mean3 && std3, "cs_crop_resize_u8: null pointer");
    CS_CHECK_ARG(H > 0 && W > 0 && K > 0 && S > 0 && K <= 65535, "cs_crop_resize_u8: bad sizes H=%d W=%d K=%d S=%d", H, W, K, S);
    CS_CHECK_ARG((long)max(H, W) * 4 + 1 <= (long)MAXTAPS * S, "cs_crop_resize_u8: down-sampling factor too large for %d taps", MAXTAPS);
    dim3 block(256);
    hipLaunchKernelGGL(crop_hpass_kernel, dim3((S + 63) / 64, 16, K), block, 0, stream, (const uint8_t*)src, H, W, boxes, S, pad_center,
                       (uint8_t*)workspace, H);
    CS_LAUNCH_CHECK();
    hipLaunchKernelGGL(crop_vpass_kernel, dim3((S + 63) / 64, (S + 3) / 4, K), block, 0, stream, (const uint8_t*)workspace, H, H, W, boxes, S,
                       pad_center, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out);
    CS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ --multiscale
// Bilinear resize of fp32 planes, the arithmetic of torch's F.interpolate(mode='bilinear', align_corners=False) that the reference applies
// to the student images under --multiscale (src/training/clipself.py:17-27): src = scale * (dst + 0.5) - 0.5 clamped at 0, scale = in / out.
namespace {

__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int H, int W,
                                                              int Ho, int Wo, float sh, float sw) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= Wo || y >= Ho) return;
    const float fy = fmaxf(sh * ((float)y + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * ((float)x + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int yp = y0 < H - 1 ? 1 : 0, xp = x0 < W - 1 ? 1 : 0;
    const float ly1 = fy - (float)y0, ly0 = 1.f - ly1, lx1 = fx - (float)x0, lx0 = 1.f - lx1;
    for (int p = blockIdx.z; p < planes; p += gridDim.z) {
        const float* s = in + (size_t)p * H * W + (size_t)y0 * W + x0;
        const float top = lx0 * s[0] + lx1 * s[xp], bot = lx0 * s[(size_t)yp * W] + lx1 * s[(size_t)yp * W + xp];
        out[(size_t)p * Ho * Wo + (size_t)y * Wo + x] = ly0 * top + ly1 * bot;
    }
}

}  // namespace

// in [planes, H, W] f32 -> out [planes, Ho, Wo] f32 (planes = batch * channels), both contiguous
extern "C" int cs_resize_bilinear_f32(const float* in, float* out, int planes, int H, int W, int Ho, int Wo, hipStream_t stream) {
    CS_CHECK_ARG(in && out && planes > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "cs_resize_bilinear_f32: bad arguments");
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((Wo + 63) / 64, (Ho + 3) / 4, planes < 1024 ? planes : 1024), dim3(256), 0, stream, in, out,
                       planes, H, W, Ho, Wo, (float)H / (float)Ho, (float)W / (float)Wo);
    CS_LAUNCH_CHECK();
    return 0;
}
