// Flat multi-tensor AdamW for the student (decoupled weight decay, bias-corrected), one launch for all
// trainable tensors.  Reference: torch.optim.AdamW as constructed in src/training/main.py:198-213
// (no-decay group = ndim<2 / "ln" / "bias" / "logit_scale"; decay group wd=args.wd), stepped at train.py:115;
// parameters whose grad is None are skipped entirely (no decay either) -- SURVEY.md D7.
//
// MI355X layout: all trainable parameters live in ONE fp32 master buffer (every tensor starts on a 64-element
// boundary, total padded to 256) with same-layout fp32 grad / exp_avg / exp_avg_sq buffers and a same-layout bf16 "shadow" that the
// MFMA GEMMs read.  A flag byte per 64 elements carries (bit0) "has a gradient this step" and (bit1)
// "weight decay applies".  The kernel is a pure HBM stream: 4 fp32 reads + 3 fp32 writes + 1 bf16 write per element.
#include "cs_common.h"

namespace {

struct AdamArgs {
    float* p; const float* g; float* m; float* v; __bf16* shadow; const uint8_t* flags;
    long nchunks;
    float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, grad_scale;
};

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
    // one wave per 256-element chunk, 4 elements per lane
    const long chunk = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (chunk >= a.nchunks) return;
    const uint8_t f = a.flags[chunk * 4 + ((threadIdx.x & 63) >> 4)];   // one flag byte per 64 elements (= 16 lanes)
    if (!(f & 1)) return;
    const long i = chunk * 256 + (threadIdx.x & 63) * 4;
    float4 p = *(const float4*)(a.p + i);
    const float4 g4 = *(const float4*)(a.g + i);
    float4 m = *(const float4*)(a.m + i), v = *(const float4*)(a.v + i);
    const float decay = (f & 2) ? 1.f - a.lr * a.wd : 1.f;
    const float step = a.lr / a.bc1;
    float pp[4] = {p.x, p.y, p.z, p.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w}, mm[4] = {m.x, m.y, m.z, m.w}, vv[4] = {v.x, v.y, v.z, v.w};
    U64 sh;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float g = gg[e] * a.grad_scale;
        pp[e] *= decay;                                       // p.mul_(1 - lr*wd)
        mm[e] = a.beta1 * mm[e] + (1.f - a.beta1) * g;        // exp_avg.lerp_(grad, 1-beta1)
        vv[e] = a.beta2 * vv[e] + (1.f - a.beta2) * g * g;    // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
        const float denom = sqrtf(vv[e]) / a.bc2_sqrt + a.eps;
        pp[e] -= step * (mm[e] / denom);
        sh.e[e] = f2bf(pp[e]);
    }
    *(float4*)(a.p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
    *(float4*)(a.m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
    *(float4*)(a.v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    if (a.shadow) *(uint2*)(a.shadow + i) = sh.u;
}

}  // namespace

// n must be a multiple of 256; flags has n/64 bytes (bit0 = active, bit1 = decay).  `step` is the 1-based AdamW
// step count (bias corrections computed here in double like torch's scalar path).
extern "C" int cs_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, const uint8_t* flags, long n,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                             hipStream_t stream) {
    CS_CHECK_ARG(n > 0 && n % 256 == 0, "cs_adamw_step: n must be a positive multiple of 256");
    CS_CHECK_ARG(step >= 1, "cs_adamw_step: step is 1-based");
    AdamArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.shadow = (__bf16*)shadow_bf16; a.flags = flags; a.nchunks = n / 256;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay; a.grad_scale = grad_scale;
    a.bc1 = (float)(1.0 - __builtin_pow((double)beta1, (double)step));
    a.bc2_sqrt = (float)__builtin_sqrt(1.0 - __builtin_pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adamw_kernel, dim3((int)((a.nchunks + 3) / 4)), dim3(256), 0, stream, a);
    CS_LAUNCH_CHECK();
    return 0;
}
