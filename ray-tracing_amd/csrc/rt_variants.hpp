// rt_variants.hpp — the feature masks the hot kernels are instantiated for, smallest first, and the
// rule that picks one: the first mask that covers every feature of the scene (F_ALL always does).
#pragma once
#include "rt_device.hpp"

namespace rt {
constexpr uint32_t V_CORNELL = F_XFORM | F_SPHERE | F_RECT | F_CUBOID | F_DIELECTRIC | F_LIGHTS;           // cornell_box
constexpr uint32_t V_SMOKE = V_CORNELL | F_MEDIUM;                                                          // cornell_smoke
constexpr uint32_t V_SPHERES = F_BVH | F_SPHERE | F_MOVING | F_METAL | F_DIELECTRIC | F_TEXTURED | F_UV;   // random_scene, earth, checker spheres
constexpr uint32_t V_SPHERES_FLAT = V_SPHERES & ~F_BVH;                                                      // earth, two_checker_spheres (no LDS node table)
constexpr uint32_t V_NOISE = F_SPHERE | F_TEXTURED | F_NOISE;                                               // two_perlin_spheres
#ifndef RT_V_FINAL_MASK
#define RT_V_FINAL_MASK (0x3fffu & ~F_FROSTED)
#endif
constexpr uint32_t V_FINAL = RT_V_FINAL_MASK;  // final_scene: every feature but FrostedDielectric (only random_scene_light has one)

// Register budget of the persistent render kernel per instantiation, as waves per SIMD (512 VGPRs per lane and SIMD: 2 -> 256,
// 3 -> 168, 4 -> 128), and the LDS budget of the staged node table (32-byte entries), sized so that LDS admits as many blocks per CU
// as the register budget does.  All chosen from measurements; DESIGN.md §4 has the history (round 1: 4 waves for the three-way
// walk; round 2: 3 waves, because at 4 the f64 variants spilled 430-670 B per lane and the scratch of the resident waves outgrew the
// L2 — up to 257 GB per launch went to memory).  State after round 3, f64:
//  * spheres + BVH (random_scene): 4 waves, 96 B of scratch per lane, 1024-entry table per 256-lane block; 1.04 GB per launch.
//  * generic variant (final_scene): 4 waves, 112 B (0 B at 3 waves, 9 % slower), 7.1 GB per launch, L2 hit 97 % — and not a byte to
//    spare: 4096 resident waves x 64 lanes x 112 B = 29.4 MB of the 32 MB of L2; at 128 B the traffic doubles (measured).  864
//    entries per 256-lane block, or one 1024-lane block per CU with 3945 (wave_block, below) when the trees need more.  The
//    catch-all with FrostedDielectric: 3 waves, 80 B (240 B at 4 waves, 5 % slower).
//  * no tree: flat (u,v) spheres (earth) 4 waves / 16 B; noise (two_perlin_spheres) 4 waves / 32 B; plain variants 4 waves / 16-32 B.
//  tests/test_kernel_budgets.py holds the built library to these figures.
//  What made 4 waves affordable: the large libm bodies (atan2 / asin, sincos, sin, log) are calls, not inlined code
//  (rt_device.hpp), the media search has a branch of its own (F_MEDIUM_BVH), and radiance / counters are not carried per lane.
//  f32: 4 waves everywhere.  -DRT_WAVES_PER_SIMD=n / -DRT_WPS_*=n / -DRT_LDS_BVH_ENTRIES=n build the other forms.
template <class T, uint32_t F> __device__ __host__ constexpr int waves_per_simd() {
#ifdef RT_WAVES_PER_SIMD
    return RT_WAVES_PER_SIMD;
#else
#ifndef RT_WPS_SPHERES_BVH
#define RT_WPS_SPHERES_BVH 4
#endif
#ifndef RT_WPS_FLAT_UV
#define RT_WPS_FLAT_UV 4
#endif
    if (F & F_BVH) {
#ifndef RT_WPS_GENERIC
#define RT_WPS_GENERIC 4
#endif
        // the generic variant (final_scene); the catch-all with FrostedDielectric (random_scene_light) spills 240 B at 128 registers
        // and measures 5 % faster at 3 waves (1380 -> 1455 Msamples/s)
        if (F & (F_MEDIUM | F_XFORM | F_LIGHTS | F_NOISE)) return sizeof(T) == 8 ? ((F & F_FROSTED) ? 3 : RT_WPS_GENERIC) : 4;
        return sizeof(T) == 8 ? RT_WPS_SPHERES_BVH : 4;                                     // spheres + BVH (random_scene)
    }
#ifndef RT_WPS_PLAIN
#define RT_WPS_PLAIN 4  // (the wave kernel of the Cornell variants — RT_RENDER_NO_SORT / _DETERMINISTIC: 140 registers at 3 waves, 32 B of spills at 4: +8 % / +6 %)
#endif
    return sizeof(T) == 8 ? ((F & F_UV) ? RT_WPS_FLAT_UV : ((F & F_NOISE) ? 4 : RT_WPS_PLAIN)) : 4;
#endif
}
// Lanes per block of the persistent wave kernel.  Its waves never meet at a barrier after start-up, so a variant that stages a
// BVH table in LDS can run as ONE block per CU (BIG: 256 lanes x waves per SIMD = 1024): the CU then holds one copy of the node
// table instead of one per 256-lane block, i.e. room for four times the entries.  Round 3, same box, f64 (then at 3 waves): final_scene
// (2798 entries; 3 blocks per CU held 1184 each: 42 % of the fetches from LDS, the rest from L2) 319.5 -> 334.7 Msamples/s at
// 1024^2 @ 64 spp; random_scene, whose 959 entries fit either way, LOSES 1-4 % (a fixed ~0.1 ms per launch and ~1 %) — so the
// launcher takes the big block only when the scene's trees do not fit the small block's table (rt_kernels.inc: launch_variant).
template <class T, uint32_t F, bool BIG> __device__ __host__ constexpr uint32_t wave_block() {
    return ((F & F_BVH) && BIG) ? 256u * (uint32_t)waves_per_simd<T, F>() : 256u;
}
template <class T, uint32_t F> __device__ __host__ constexpr uint32_t lds_perlin_tables();
template <class T, uint32_t F, bool BIG = false> __device__ __host__ constexpr uint32_t lds_bvh_entries() {
#ifdef RT_LDS_BVH_ENTRIES
    return RT_LDS_BVH_ENTRIES;
#else
    if (wave_block<T, F, BIG>() > 256u) {
        // one block per CU: all of the CU's 160 KB less the per-wave pixel sums (1.5 KB each), the Perlin table(s) and 2 KB to spare
        return (163840u - wave_block<T, F, BIG>() / 64u * 1536u - lds_perlin_tables<T, F>() * (uint32_t)sizeof(DPerlin<T>) - 2048u
#ifdef RT_PHASE_CLOCK
                - 4096u  // (the profiling build's per-wave clock slots)
#endif
                ) / 32u;
    }
    // f64 generic: 4 blocks x (27 KB + 6.75 KB Perlin table + 6 KB of pixel sums), or 3 blocks x (37 KB + ...)
    if (sizeof(T) == 8 && (F & F_BVH) && (F & (F_MEDIUM | F_XFORM | F_LIGHTS | F_NOISE))) return waves_per_simd<T, F>() == 4 ? 864u : 1184u;
    return (sizeof(T) == 4 && (F & F_NOISE)) ? 896u : 1024u;  // f32 generic: 4 blocks x (28 KB + 4.75 KB Perlin table + 6 KB)
#endif
}
// Perlin tables staged in LDS (scenes.rs uses one per NoiseTexture: two_perlin_spheres 2, final_scene 1)
template <class T, uint32_t F> __device__ __host__ constexpr uint32_t lds_perlin_tables() {
    return (F & F_NOISE) ? ((F & F_BVH) ? 1u : 2u) : 0u;
}
// The block-sorted kernel waits at two barriers per iteration: it wants resident waves even at the price of a few
// spills.  Measured on cornell_box / cornell_smoke (profiles/): f64 best at 4 waves/SIMD (128 registers,
// 44 B of spills; at 5 waves 96 registers spill 176 B and lose 15 %), f32 best at 7 (73 registers: 4 -> 7 waves
// +17 % / +31 %).  The figure must be one the kernel's LDS admits — the compiler drops an unreachable request
// altogether and then allocates up to 256 registers.
#ifndef RT_SORT_TILE_H
#define RT_SORT_TILE_H 8  // the sorted kernel's tile is 8 pixels wide and this high (rt_kernels.inc)
#endif
#ifndef RT_SORT_BLOCK
#define RT_SORT_BLOCK 256  // lanes per block of the sorted kernel (rt_kernels.inc)
#endif
template <class T, uint32_t F> __device__ __host__ constexpr uint32_t sorted_lds_bytes() {
    return (12u + ((F & F_MOVING) ? 1u : 0u) + ((F & F_UV) ? 2u : 0u)) * RT_SORT_BLOCK * (uint32_t)sizeof(T) + 24u * RT_SORT_BLOCK + 384u * RT_SORT_TILE_H + 144u +
           lds_perlin_tables<T, F>() * (uint32_t)sizeof(DPerlin<T>);
}
#ifndef RT_SORTED_UV_WAVES
#define RT_SORTED_UV_WAVES 3
#endif
template <class T, uint32_t F> __device__ __host__ constexpr int waves_per_simd_sorted() {
#ifdef RT_SORTED_MAX_WAVES
    constexpr uint32_t most = RT_SORTED_MAX_WAVES;
#else
    // f64 with media (cornell_smoke): at 4 waves the persistent-block kernel spills 204 B per lane, 53 MB of scratch for the resident
    // lanes, more than the L2 holds (29.8 GB per 31 ms launch went to the memory side); at 3 waves 24 B: +2 % (2255 -> 2298 Msamples/s
    // at 1024 spp per launch, 2170 -> 2218 at 64).  Without media 3 waves lose 12 % (cornell_box: spills are 112 B there and fit).
    // ROUND 3: with the media search restructured (F_MEDIUM_BVH) and the range-limited sincos as a call the media variant spills 64 B at
    // 4 waves (0 at 3): 4 waves measure cornell_smoke 2484 -> 2754 Msamples/s @ 64 spp per launch, 2574 -> 2885 @ 1024.
    constexpr uint32_t most = sizeof(T) == 8 ? 4u : 7u;
#endif
    // f64 with (u,v) carried: 2 waves in rounds 1-2 (168 registers spilled 280 B per lane, 7 % slower on earth); with the (u,v) and
    // sincos bodies out of line (round 3) the sorted (u,v) variants need 168 registers and spill nothing: RT_SORTED_UV_WAVES = 3
    // blocks per CU that LDS admits x waves per block / 4 SIMDs
    constexpr uint32_t by_lds = 163840u / sorted_lds_bytes<T, F>() * (RT_SORT_BLOCK / 64u) / 4u;
    return (sizeof(T) == 8 && (F & F_UV)) ? (by_lds >= 3u ? RT_SORTED_UV_WAVES : (int)by_lds) : (by_lds >= most ? (int)most : (int)(by_lds < 1u ? 1u : by_lds));
}
}  // namespace rt

#define RT_FOR_VARIANT(features, CALL)                                       \
    do {                                                                     \
        if (((features) & ~rt::V_CORNELL) == 0) { CALL(rt::V_CORNELL); }     \
        else if (((features) & ~rt::V_SMOKE) == 0) { CALL(rt::V_SMOKE); }    \
        else if (((features) & ~rt::V_SPHERES_FLAT) == 0) { CALL(rt::V_SPHERES_FLAT); } \
        else if (((features) & ~rt::V_SPHERES) == 0) { CALL(rt::V_SPHERES); }\
        else if (((features) & ~rt::V_NOISE) == 0) { CALL(rt::V_NOISE); }    \
        else if (((features) & ~rt::V_FINAL) == 0) { CALL(rt::V_FINAL); }    \
        else { CALL(rt::F_ALL); }                                            \
    } while (0)
