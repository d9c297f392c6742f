// ORACLE — TEST INFRASTRUCTURE ONLY.  Command-line front end: renders a scene with the CPU
// restatement and reports Msamples/s (the bench's `cpu_baseline` leg, kind "port") or writes a
// binary PPM for eyeballing.
//   oracle_render <scene> <W> <H> <spp> <threads> [seed] [image.ppm] [out.ppm]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "ray_tracing.hpp"

namespace orc {
Scene build_scene(const std::string& name, double aspect, uint64_t seed, const std::string& image_path);
}

int main(int argc, char** argv) {
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s scene W H spp threads [seed] [image.ppm] [out.ppm]\n", argv[0]);
        return 2;
    }
    std::string name = argv[1];
    uint32_t W = (uint32_t)std::atoi(argv[2]), H = (uint32_t)std::atoi(argv[3]), spp = (uint32_t)std::atoi(argv[4]);
    int threads = std::atoi(argv[5]);
    uint64_t seed = argc > 6 ? std::strtoull(argv[6], nullptr, 0) : 0x5EED0001ull;
    std::string image = argc > 7 ? argv[7] : "";
    std::string out = argc > 8 ? argv[8] : "";
    try {
        orc::Scene sc = orc::build_scene(name, (double)W / (double)H, seed, image);
        std::vector<double> accum((size_t)W * H * 3, 0.0);
        auto t0 = std::chrono::steady_clock::now();
        orc::Counters c = orc::render(sc, W, H, 0, spp, 50, seed, threads, accum.data());
        double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        double n = (double)W * H * spp;
        std::printf("{\"scene\": \"%s\", \"W\": %u, \"H\": %u, \"spp\": %u, \"threads\": %d, \"seed\": %llu, \"seconds\": %.6f, "
                    "\"msamples_per_s\": %.6f, \"casts_per_path\": %.4f, \"prim_tests_per_path\": %.3f, "
                    "\"node_visits_per_path\": %.3f, \"flops_per_path\": %.2f, \"nan_retries\": %llu}\n",
                    name.c_str(), W, H, spp, threads, (unsigned long long)seed, sec, n / sec / 1e6,
                    c.paths ? (double)c.casts / c.paths : 0.0, c.paths ? (double)c.prim_tests / c.paths : 0.0,
                    c.paths ? (double)c.node_visits / c.paths : 0.0, c.paths ? c.flops / c.paths : 0.0, c.nan_retries);
        if (!out.empty()) {
            std::vector<uint8_t> rgb((size_t)W * H * 3);
            orc::resolve(accum.data(), W, H, spp, rgb.data());
            FILE* f = std::fopen(out.c_str(), "wb");
            if (!f) { std::perror("fopen"); return 1; }
            std::fprintf(f, "P6\n%u %u\n255\n", W, H);
            std::fwrite(rgb.data(), 1, rgb.size(), f);
            std::fclose(f);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
