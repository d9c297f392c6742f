// ORACLE — TEST INFRASTRUCTURE ONLY.
// Random numbers. The reference draws from rand 0.7.3 `SmallRng` (= rand_pcg 0.2.1 Pcg64Mcg on
// 64-bit targets, Cargo.lock:552-553,594-595) seeded `from_entropy` everywhere
// (ray_tracing.rs:93, scenes.rs:10,174,844, texture.rs:122,131) plus `thread_rng`/`rand::random`
// (bvh.rs:89, objects.rs:540). No reference output depends on a reproducible stream; only the
// distributions matter ("parity unpinned — statistical only" for every random draw).
//
// The oracle therefore uses the SAME counter-keyed generator the HIP path uses, so the f64 device
// instantiation can be compared sample-for-sample:
//   key   = (pixel << 32) | sample            (scene construction uses key = ~stream)
//   state = two splitmix64 outputs of (seed + key * GOLDEN)
//   draw  = xoshiro128++ (Blackman & Vigna, public domain algorithm), 32 bits per step
// Distributions restated from rand 0.7.3:
//   gen::<f64>()        53 high bits * 2^-53 in [0,1)
//   gen_range(lo,hi)    52-bit fraction scaled into [lo,hi), redrawn if rounding reaches hi
//   gen_range(0,n) int  widening multiply with rejection zone (unbiased)
#pragma once
#include <cstdint>

namespace orc {

struct Rng {
    uint32_t s0, s1, s2, s3;

    static inline uint64_t splitmix64(uint64_t& z) {
        z += 0x9E3779B97F4A7C15ull;
        uint64_t r = z;
        r = (r ^ (r >> 30)) * 0xBF58476D1CE4E5B9ull;
        r = (r ^ (r >> 27)) * 0x94D049BB133111EBull;
        return r ^ (r >> 31);
    }
    static Rng from_key(uint64_t seed, uint64_t key) {
        uint64_t z = seed + key * 0x9E3779B97F4A7C15ull;
        uint64_t a = splitmix64(z);
        uint64_t b = splitmix64(z);
        Rng r;
        r.s0 = (uint32_t)a;
        r.s1 = (uint32_t)(a >> 32);
        r.s2 = (uint32_t)b;
        r.s3 = (uint32_t)(b >> 32);
        if ((r.s0 | r.s1 | r.s2 | r.s3) == 0) r.s0 = 1;
        return r;
    }
    static Rng for_sample(uint64_t seed, uint32_t pixel, uint32_t sample) {
        return from_key(seed, ((uint64_t)pixel << 32) | sample);
    }
    static Rng for_scene(uint64_t seed, uint32_t stream) { return from_key(seed, ~(uint64_t)stream); }

    static inline uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    inline uint32_t next_u32() {
        uint32_t result = rotl(s0 + s3, 7) + s0;
        uint32_t t = s1 << 9;
        s2 ^= s0;
        s3 ^= s1;
        s1 ^= s2;
        s0 ^= s3;
        s2 ^= t;
        s3 = rotl(s3, 11);
        return result;
    }
    inline uint64_t next_u64() {
        uint64_t hi = next_u32();
        uint64_t lo = next_u32();
        return (hi << 32) | lo;
    }
    inline double gen() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
    inline double gen_range(double lo, double hi) {
        for (;;) {
            double u = (double)(next_u64() >> 12) * (1.0 / 4503599627370496.0);
            double v = lo + (hi - lo) * u;
            if (v < hi) return v;
        }
    }
    inline uint32_t gen_index(uint32_t n) {
        uint64_t m = (uint64_t)next_u32() * n;
        uint32_t l = (uint32_t)m;
        if (l < n) {
            uint32_t thresh = (0u - n) % n;
            while (l < thresh) {
                m = (uint64_t)next_u32() * n;
                l = (uint32_t)m;
            }
        }
        return (uint32_t)(m >> 32);
    }
};

}  // namespace orc
