// f32 instantiation (+ the type-independent resolve kernel).
#define RT_REAL float
#include "rt_kernels.inc"

namespace rt {
// src/ray_tracing.rs:112-117,126-128
__global__ void resolve_kernel(const float* __restrict__ accum, uint32_t width, uint32_t height, uint32_t spp, uint8_t* __restrict__ rgb) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= width * height * 3) return;
    const uint32_t c = idx % 3, pix = idx / 3, x = pix % width, y = pix / width;
    double col = (double)accum[idx] / (double)spp;
    double g = ::sqrt(col);
    g = g < 0.0 ? 0.0 : (g > 0.99999 ? 0.99999 : g);
    double q = g * 256.0;
    uint8_t b = (q != q) ? 0 : (q >= 255.0 ? 255 : (q <= 0.0 ? 0 : (uint8_t)q));
    rgb[((size_t)(height - 1 - y) * width + x) * 3 + c] = b;
}
hipError_t launch_resolve(const float* d_accum, uint32_t width, uint32_t height, uint32_t spp, uint8_t* d_rgb, hipStream_t stream) {
    uint32_t n = width * height * 3u;
    hipLaunchKernelGGL(resolve_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, d_accum, width, height, spp, d_rgb);
    return hipGetLastError();
}
}  // namespace rt
