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

// Register budget of the persistent render kernel per instantiation, as waves per SIMD (512 VGPRs per lane and
// SIMD: 2 -> 256, 3 -> 168, 4 -> 128), and the LDS budget of the staged BVH head (32-byte entries), sized so that
// LDS admits as many blocks per CU as the register budget does.  All chosen from measurements (profiles/):
//  * BVH walks are chains of dependent LDS fetches: they want resident waves more than they mind spills.
//    spheres + BVH (random_scene): 4 waves with a 1024-entry head (32 KB + 6 KB tile sums per block) — f64 861 ->
//    1173 Msamples/s against 2 waves / 2048 entries, f32 1232 -> 1414.  The generic variant (final_scene) stops at
//    3 waves (f64 122 -> 144; at 4 its 128 registers spill it to half speed).
//    With the two-way walk over the device's own trees (round 2) a node step is cheaper and the f64 spills weigh more:
//    f64 3 waves 2026 against 1753 at 4 (and 1664 at 2); f32 stays at 4 (2970 against 2591 at 3).
//    The generic f64 variant runs as fast at 2 waves as at 3 (final_scene 1024^2 @ 64 spp: 227 against 225 Msamples/s) with
//    236 B of spills per lane instead of 576 B: at 3 waves the scratch of all resident waves (113 MB) does not fit the
//    32 MB of L2 and 257 GB per launch went to HBM; at 2 waves it fits.  f32 kept 3 (370 against 272).
//    LATER IN ROUND 2, once the uniform scene reads were scalar loads again (rt_kernels.inc: the job counter as a restrict
//    parameter: 46 fewer live vector registers' worth of records), the balance tipped: 3 waves with 1184 LDS entries measure
//    278 against 239 Msamples/s (f64, final_scene 1024^2 @ 64 spp; 314 against 255 at 2048^2) and f32 at 4 waves with 896 entries
//    418 against 375 — issue slots busy 74 % instead of 58 %.  The price is the spill traffic again: 524 B per lane, 196 GB per
//    launch through the memory side (810 GB/s, a tenth of the HBM peak; profiles/r02_final_*).  Throughput decides: 3 / 4 waves.
//    -DRT_WAVES_PER_SIMD=2 -DRT_LDS_BVH_ENTRIES=2048 builds the low-traffic form (0.45 GB per launch).
//    END OF ROUND 2 (scalar scene reads, streaming jobs, camera read at path start), measured again: spheres + BVH f64 at 4 waves is
//    now 5-7 % FASTER than at 3 (random_scene 800x450 @ 100 spp 2281 -> 2388, @ 1600 spp 2505 -> 2689) — with 428 instead of 288 B
//    of spills per lane, 112 MB of scratch resident and 11.7 GB instead of 0.65 GB per 15 ms launch through the memory side
//    (profile of that build: issue slots busy 87 %, L2 hit 78 %).  Kept at 3: eighteen times the traffic for 5 % is the wrong
//    trade for a kernel that shares the memory system with whatever else a node runs; -DRT_WAVES_PER_SIMD=4 builds the other form.
//    The generic f64 variant stays at 3 either way (4 waves with an 864-entry head: 672 B of spills, -3 %).
//  * no BVH: f64 3 waves (fits 168 without spilling, +23 % over 2), except the (u,v) / texture variants whose 168
//    registers spill 272 B per lane (2 waves faster) and the noise variant (4 waves: +3 %); f32 4 waves.
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
        if (F & (F_MEDIUM | F_XFORM | F_LIGHTS | F_NOISE)) return sizeof(T) == 8 ? RT_WPS_GENERIC : 4;  // the generic variant (final_scene)
        return sizeof(T) == 8 ? RT_WPS_SPHERES_BVH : 4;                                     // spheres + BVH (random_scene)
    }
    return sizeof(T) == 8 ? ((F & F_UV) ? RT_WPS_FLAT_UV : ((F & F_NOISE) ? 4 : 3)) : 4;
#endif
}
// Lanes per block of the persistent wave kernel.  Its waves never meet at a barrier after start-up, so a variant that stages a
// BVH table in LDS can run as ONE block per CU (BIG: 256 lanes x waves per SIMD — 768 in f64, 1024 in f32): the CU then holds one
// copy of the node table instead of one per 256-lane block, i.e. room for 3-4x the entries.  Round 3, same box, f64: final_scene
// (2798 entries; 3 blocks per CU hold 1184 each: 42 % of the fetches from LDS, the rest from L2) 319.5 -> 334.7 Msamples/s at
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
    if (sizeof(T) == 8 && (F & F_BVH) && (F & (F_MEDIUM | F_XFORM | F_LIGHTS | F_NOISE))) return RT_WPS_GENERIC == 4 ? 864u : 1184u;  // f64 generic: 3 blocks x (37 KB + 6.75 KB Perlin table + 6 KB)
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
    // f64 with (u,v) carried: 168 registers (3 waves) spill 280 B per lane and measured 7 % slower than 2 waves (earth)
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
