// f64 instantiation: compile with -ffp-contract=off (see build.py).
#define RT_REAL double
#include "rt_kernels.inc"

namespace rt {
// Test hook (rt_debug_divide): quotients by the shared-denominator route (Recip<double>, three numerators per
// denominator as unit() uses it) and by the plain `/`, for a bit-for-bit comparison on the device.
__global__ void divide_check_kernel(const double* __restrict__ n, const double* __restrict__ d, uint64_t count, double* __restrict__ shared,
                                    double* __restrict__ plain) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const SharedRecip<double> r(d[i]);
    shared[i] = r.div(n[i]);
    plain[i] = n[i] / d[i];
}
// (rt_debug_divide_fast): the unguarded quotient routine (RecipCore: what unit(v) and v / s use)
__global__ void divide_fast_kernel(const double* __restrict__ n, const double* __restrict__ d, uint64_t count, double* __restrict__ out, uint32_t* __restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
#ifdef RT_HAVE_RECIP_CORE  // (device pass only: the routine is built from device builtins)
    const RecipCore f(d[i]);
    out[i] = f.quot(n[i]);
    flag[i] = RecipCore::moderate(d[i]) ? 0u : 1u;
#endif
}
// Test hook (rt_debug_convert): Rust's `as i32` / `as u32` by the hardware conversion the device code uses and by the spelled-out form
__global__ void convert_kernel(const double* __restrict__ x, uint64_t count, int32_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const double v = x[i];
    const float f = (float)v;
    int32_t* o = out + i * 8;
    o[0] = as_i32(v); o[1] = (int32_t)as_u32(v); o[2] = as_i32(f); o[3] = (int32_t)as_u32(f);
    o[4] = as_i32_portable(v); o[5] = (int32_t)as_u32_portable(v); o[6] = as_i32_portable(f); o[7] = (int32_t)as_u32_portable(f);
}
hipError_t launch_convert_check(const double* d_x, uint64_t count, int32_t* d_out, hipStream_t stream) {
    hipLaunchKernelGGL(convert_kernel, dim3((uint32_t)((count + 255u) / 256u)), dim3(256), 0, stream, d_x, count, d_out);
    return hipGetLastError();
}
hipError_t launch_divide_fast(const double* d_n, const double* d_d, uint64_t count, double* d_out, uint32_t* d_flag, hipStream_t stream) {
    hipLaunchKernelGGL(divide_fast_kernel, dim3((uint32_t)((count + 255u) / 256u)), dim3(256), 0, stream, d_n, d_d, count, d_out, d_flag);
    return hipGetLastError();
}
hipError_t launch_divide_check(const double* d_n, const double* d_d, uint64_t count, double* d_shared, double* d_plain, hipStream_t stream) {
    hipLaunchKernelGGL(divide_check_kernel, dim3((uint32_t)((count + 255u) / 256u)), dim3(256), 0, stream, d_n, d_d, count, d_shared, d_plain);
    return hipGetLastError();
}
}  // namespace rt
