// TEST INFRASTRUCTURE ONLY — lets g++ compile the device headers (ray-tracing_amd/csrc/rt_device.hpp,
// rt_path.hpp, rt_pack.hpp) for the CPU, so the not-gpu test suite can run the very templates the
// HIP kernels instantiate against the oracle.  Never part of the product: librt_hip.so is built by
// hipcc without RT_HOST_SIM and contains no host rendering path.
#pragma once
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <cmath>
#include <cstddef>
#include <cstdint>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
