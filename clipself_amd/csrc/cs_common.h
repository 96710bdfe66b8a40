// Shared device helpers for the CLIPSelf hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CS_WAVE 64

// ---- error plumbing (no exceptions across the C ABI) -------------------------------------
extern "C" const char* cs_last_error();
void cs_set_error(const char* fmt, ...);
#define CS_CHECK_ARG(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            cs_set_error(__VA_ARGS__);     \
            return -1;                     \
        }                                  \
    } while (0)
#define CS_LAUNCH_CHECK()                                                    \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) {                                             \
            cs_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return -2;                                                       \
        }                                                                    \
    } while (0)

// ---- bf16 <-> f32 --------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(__bf16 v) { return (float)v; }
__device__ __forceinline__ __bf16 f2bf(float v) { return (__bf16)v; }   // RNE, v_cvt_pk_bf16_f32 on gfx950

__device__ __forceinline__ float bfbits2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

union U128 {
    uint4 u;
    bf16x8 h;
    __bf16 e[8];
};
union U64 {
    uint2 u;
    bf16x4 h;
    __bf16 e[4];
};

// ---- wave reductions (64 lanes) --------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// sum over aligned groups of 8 lanes with DPP moves only (no LDS traffic): quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum_lanes8(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    return v;
}

__device__ __forceinline__ float sum_lanes16(float v) {      // aligned groups of 16 lanes: + row_mirror (lane i <-> 15 - i)
    v = sum_lanes8(v);
    v += dpp_move<0x140>(v);
    return v;
}

// MFMA 32x32 accumulator register r of lane l holds C[row][col] with
//   col = l & 31,  row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)        (cdna_hip_programming.md §3)
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
