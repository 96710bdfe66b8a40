// Runtime plumbing of the C ABI that is not a kernel: HIP streams restricted to a range of compute units.
//
// The frozen teacher's pass over batch i+1 and the student's step on batch i are independent (src/training/clipself.py:36-40: the teacher
// runs under no_grad on its own crops); on one GPU they share the chip.  Two in-order queues do not share it well by themselves -- a
// 256-workgroup persistent GEMM of one tower owns every CU for 1-2.5 ms and the other tower's kernels wait -- so the step can give each
// tower its own compute units: persistent grids sized to the partition (cs_gemm_nt flags bits 20-27) and, optionally, queues whose CU mask
// the hardware dispatcher enforces (this file).
#include <hip/hip_runtime.h>
#include <cstdint>
#include "cs_common.h"
#include "gemm_common.h"

extern "C" int cs_num_compute_units(void) { return cs_num_cus(); }

// A stream whose kernels may only run on compute units [first_cu, first_cu + n_cus) of the mask enumeration.  On a multi-XCD part the
// driver deals mask bit i to XCD i % 8 (then round-robin over that XCD's shader engines), so a contiguous bit range whose ends are
// multiples of 8 takes the same number of CUs from every XCD -- what the XCD-aware rasters of the persistent kernels assume.
extern "C" int cs_stream_create_cu_mask(int first_cu, int n_cus, hipStream_t* out) {
    const int total = cs_num_cus();
    CS_CHECK_ARG(out != nullptr, "cs_stream_create_cu_mask: out is NULL");
    CS_CHECK_ARG(first_cu >= 0 && n_cus >= 8 && first_cu + n_cus <= total && first_cu % 8 == 0 && n_cus % 8 == 0,
                 "cs_stream_create_cu_mask: [%d, %d) must be a multiple-of-8 range inside the device's %d compute units", first_cu, first_cu + n_cus, total);
    uint32_t mask[32] = {0};
    const int words = (total + 31) / 32;
    CS_CHECK_ARG(words <= 32, "cs_stream_create_cu_mask: %d compute units", total);
    for (int i = first_cu; i < first_cu + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    const hipError_t e = hipExtStreamCreateWithCUMask(out, (uint32_t)words, mask);
    if (e != hipSuccess) {
        cs_set_error("cs_stream_create_cu_mask: hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
        return -2;
    }
    return 0;
}

extern "C" int cs_stream_destroy(hipStream_t stream) {
    const hipError_t e = hipStreamDestroy(stream);
    if (e != hipSuccess) {
        cs_set_error("cs_stream_destroy: %s", hipGetErrorString(e));
        return -2;
    }
    return 0;
}
