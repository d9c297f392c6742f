// f32 instantiation (+ the type-independent resolve kernel).
#define RT_REAL float
#include "rt_kernels.inc"

namespace rt {
hipError_t launch_resolve(const float* d_accum, uint32_t width, uint32_t height, uint32_t spp, uint8_t* d_rgb, hipStream_t stream) {
    uint32_t n = width * height * 3u;
    hipLaunchKernelGGL(resolve_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, d_accum, width, height, spp, d_rgb);
    return hipGetLastError();
}
}  // namespace rt
