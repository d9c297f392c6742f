// rt_launch.hpp — host-callable launchers, one set per arithmetic type.  The f64 set lives in a
// translation unit compiled with -ffp-contract=off (every operation rounds separately, like the
// reference's Rust f64 code); the f32 set is compiled with contraction enabled.
#pragma once
#include "rt_device.hpp"
#include "rt_variants.hpp"

namespace rt {

// What a render launch needs from the scene handle besides the scene: a zeroable device word for the persistent
// kernel's job counter (one per launch in flight) and the device's CU count.
// deterministic: RT_RENDER_DETERMINISTIC (rt_hip.h) — the wave kernel's one-lane-per-pixel form, one chunk per tile.
struct LaunchCtx { uint32_t* job_counter; uint32_t cu_count; bool deterministic; };
template <class T> hipError_t launch_render(const DScene<T>& sc, uint32_t features, const RenderArgs& a, bool no_sort, const LaunchCtx& ctx,
                                           float* d_accum, unsigned long long* d_stats, hipStream_t stream);
template <class T> hipError_t launch_trace_primary(const DScene<T>& sc, uint32_t features, uint32_t width, uint32_t height, double time,
                                                   double medium_xi, int32_t* d_ids, double* d_t, hipStream_t stream);
template <class T> hipError_t launch_sample_pixel(const DScene<T>& sc, uint32_t features, const RenderArgs& a, uint32_t x, uint32_t y,
                                                  uint32_t n, double* d_out, hipStream_t stream);
hipError_t launch_convert_check(const double* d_x, uint64_t count, int32_t* d_out, hipStream_t stream);
hipError_t launch_divide_fast(const double* d_n, const double* d_d, uint64_t count, double* d_out, uint32_t* d_flag, hipStream_t stream);
hipError_t launch_divide_check(const double* d_n, const double* d_d, uint64_t count, double* d_shared, double* d_plain, hipStream_t stream);
hipError_t launch_resolve(const float* d_accum, uint32_t width, uint32_t height, uint32_t spp, uint8_t* d_rgb, hipStream_t stream);
}  // namespace rt
