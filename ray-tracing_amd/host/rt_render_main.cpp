// rt_render — native launcher: what `run_ray_tracing` is in the reference (src/ray_tracing.rs:61-135), written against the
// two C headers only (include/rt_host.h: scene construction + flattening, include/rt_hip.h: the device path).  No Python,
// no PyTorch: this is the call sequence a Rust host would make through the bindings of INTEGRATION.md.
//
//   rt_render --scene cornell_box --width 1024 --height 1024 --spp 1024 --out out.png [--gpus N]
//
// One host THREAD per GPU (the reference: one thread per x-strip job, ray_tracing.rs:83-94).  GPU g renders the sample indices
// spp_shard(spp, g, N) of every pixel, the float sums are added by ONE rt_allreduce_accum (RCCL through the C-ABI), GPU 0
// resolves and the image is written as an 8-bit RGB PNG (ray_tracing.rs:112-133).  Prints one JSON record.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_hip.h"
#include "../../include/rt_host.h"

namespace {

struct Options {
    std::string scene = "final_scene", out = "output/ray_tracing.png", image;  // ray_tracing.rs:78,133
    uint32_t width = 400, height = 400, spp = 64, spp_per_launch = 256;         // ray_tracing.rs:75
    int depth = 50, gpus = 1;                                                   // ray_tracing.rs:105
    uint64_t seed = 0x5EED0001ull;
    uint32_t precision = RT_F64;
    bool deterministic = false;  // RT_RENDER_DETERMINISTIC: reproducible bits (one lane per pixel, samples summed in order)
};

[[noreturn]] void die(const std::string& msg) {
    std::fprintf(stderr, "rt_render: %s\n", msg.c_str());
    std::exit(1);
}

Options parse(int argc, char** argv) {
    Options o;
    for (int i = 1; i < argc; i++) {
        const std::string k = argv[i];
        auto val = [&]() -> const char* {
            if (i + 1 >= argc) die("missing value after " + k);
            return argv[++i];
        };
        if (k == "--scene") o.scene = val();
        else if (k == "--out") o.out = val();
        else if (k == "--image") o.image = val();
        else if (k == "--width") o.width = (uint32_t)std::strtoul(val(), nullptr, 0);
        else if (k == "--height") o.height = (uint32_t)std::strtoul(val(), nullptr, 0);
        else if (k == "--spp") o.spp = (uint32_t)std::strtoul(val(), nullptr, 0);
        else if (k == "--spp-per-launch") o.spp_per_launch = (uint32_t)std::strtoul(val(), nullptr, 0);
        else if (k == "--depth") o.depth = (int)std::strtol(val(), nullptr, 0);
        else if (k == "--gpus") o.gpus = (int)std::strtol(val(), nullptr, 0);
        else if (k == "--seed") o.seed = std::strtoull(val(), nullptr, 0);
        else if (k == "--precision") {
            const std::string p = val();
            if (p == "f64") o.precision = RT_F64;
            else if (p == "f32") o.precision = RT_F32;
            else die("--precision must be f64 or f32");
        } else if (k == "--deterministic") o.deterministic = true;
        else if (k == "--help" || k == "-h") {
            std::printf("usage: rt_render [--scene NAME] [--width W] [--height H] [--spp N] [--depth D] [--seed S] [--precision f64|f32]\n"
                        "                 [--spp-per-launch K] [--gpus N] [--deterministic] [--image earthmap.(jpg|ppm)] [--out file.png]\nscenes:");
            for (const char* const* n = rth_scene_names(); *n; n++) std::printf(" %s", *n);
            std::printf("\n");
            std::exit(0);
        } else die("unknown option " + k);
    }
    if (o.width < 2 || o.height < 2 || o.spp == 0 || o.spp_per_launch == 0 || o.gpus < 1 || o.gpus > 64) die("bad width/height/spp/gpus");
    return o;
}

// ---- PNG, 8-bit RGB, rows top first; deflate "stored" blocks (no compression library needed) -----------------------
uint32_t crc32_of(const uint8_t* p, size_t n, uint32_t crc) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return ~crc;
}
void put_be32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void put_chunk(std::vector<uint8_t>& png, const char tag[4], const std::vector<uint8_t>& data) {
    put_be32(png, (uint32_t)data.size());
    const size_t at = png.size();
    png.insert(png.end(), tag, tag + 4);
    png.insert(png.end(), data.begin(), data.end());
    put_be32(png, crc32_of(png.data() + at, png.size() - at, 0));
}
bool write_png(const std::string& path, const uint8_t* rgb, uint32_t w, uint32_t h) {
    std::vector<uint8_t> raw;  // filter byte 0 + the row
    raw.reserve((size_t)h * (w * 3 + 1));
    for (uint32_t y = 0; y < h; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;  // adler32
    for (size_t off = 0; off < raw.size() || off == 0;) {
        const size_t n = raw.size() - off < 65535u ? raw.size() - off : 65535u;
        const bool last = off + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back(n & 0xff); z.push_back(n >> 8); z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = off; i < off + n; i++) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        off += n;
        if (last) break;
    }
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w); put_be32(ihdr, h);
    const uint8_t tail[5] = {8, 2, 0, 0, 0};  // 8 bits, colour type 2 (RGB)
    ihdr.insert(ihdr.end(), tail, tail + 5);
    put_chunk(png, "IHDR", ihdr);
    put_chunk(png, "IDAT", z);
    put_chunk(png, "IEND", {});
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
    return std::fclose(f) == 0 && ok;
}

// GPU g of N renders samples [g*spp/N, (g+1)*spp/N) of every pixel (disjoint RNG keys; renderer.py: spp_shard)
void spp_shard(uint32_t spp, int g, int n, uint32_t& begin, uint32_t& count) {
    begin = (uint32_t)((uint64_t)spp * g / n);
    count = (uint32_t)((uint64_t)spp * (g + 1) / n) - begin;
}

struct Shared {
    const Options* opt;
    const RtSceneDesc* desc;
    uint8_t comm_id[RT_COMM_ID_BYTES];
    std::atomic<int> failed{0}, arrived{0};
    std::string error[64];
    std::vector<uint8_t> rgb;      // GPU 0: the resolved image
    uint64_t stats[RT_STATS_WORDS] = {0, 0, 0, 0};
    double seconds_render = 0.0;
};

// Failures BEFORE the communicator exists are recorded and agreed on by all GPU threads (nobody enters a collective unless every
// thread's set-up succeeded).  Once a collective may be pending, one thread that simply returned would leave the others blocked in
// rt_comm_create / rt_allreduce_accum / rt_stream_synchronize for ever: from there on a failure prints its message and ends the
// whole process.
#define CHECK(expr)                                                                                      \
    do {                                                                                                 \
        if ((expr) != RT_OK) { sh->error[g] = std::string(#expr) + ": " + rt_last_error(); return false; } \
    } while (0)
#define FATAL(expr)                                                                                      \
    do {                                                                                                 \
        if ((expr) != RT_OK) {                                                                           \
            std::fprintf(stderr, "rt_render: GPU %d: %s: %s\n", g, #expr, rt_last_error());             \
            std::fflush(stderr);                                                                         \
            std::_Exit(1);                                                                               \
        }                                                                                                \
    } while (0)

struct Gpu {
    RtScene* scene = nullptr;
    void *d_accum = nullptr, *d_stats = nullptr;
    uint64_t floats = 0;
};

// everything that can fail for reasons of this GPU alone (out of memory, a scene the device rejects): before any collective
bool gpu_setup(Shared* sh, int g, Gpu& r) {
    const Options& o = *sh->opt;
    CHECK(rt_init(g));  // (selects the device for this thread)
    CHECK(rt_scene_create(sh->desc, &r.scene));
    r.floats = (uint64_t)o.width * o.height * 3u;
    CHECK(rt_device_alloc(&r.d_accum, r.floats * sizeof(float)));
    CHECK(rt_device_alloc(&r.d_stats, RT_STATS_WORDS * sizeof(uint64_t)));
    CHECK(rt_device_zero(r.d_accum, r.floats * sizeof(float), nullptr));
    CHECK(rt_device_zero(r.d_stats, RT_STATS_WORDS * sizeof(uint64_t), nullptr));
    CHECK(rt_stream_synchronize(nullptr));
    return true;
}

void gpu_thread(Shared* sh, int g) {
    const Options& o = *sh->opt;
    const int n = o.gpus;
    Gpu r;
    if (!gpu_setup(sh, g, r)) sh->failed++;
    sh->arrived++;
    while (sh->arrived.load() < n) std::this_thread::yield();  // every thread has counted itself in, failed or not
    if (sh->failed.load() != 0) return;                        // (main prints the messages; the process ends, device memory with it)
    RtComm* comm = nullptr;
    if (n > 1) FATAL(rt_comm_create(sh->comm_id, n, g, &comm));  // collective: every GPU's thread is here
    const uint64_t floats = r.floats;
    void *d_accum = r.d_accum, *d_stats = r.d_stats, *d_rgb = nullptr;
    RtScene* scene = r.scene;
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t begin, count;
    spp_shard(o.spp, g, n, begin, count);
    for (uint32_t s = begin; s < begin + count;) {  // ray_tracing.rs:99-111, in launches of spp_per_launch samples
        const uint32_t k = begin + count - s < o.spp_per_launch ? begin + count - s : o.spp_per_launch;
        RtRenderParams p;
        std::memset(&p, 0, sizeof p);
        p.width = o.width; p.height = o.height; p.spp_begin = s; p.spp_count = k; p.max_depth = o.depth;
        p.max_nan_retries = 64; p.seed = o.seed; p.precision = o.precision; p.flags = o.deterministic ? RT_RENDER_DETERMINISTIC : 0u;
        FATAL(rt_render(scene, &p, (float*)d_accum, (uint64_t*)d_stats, nullptr));
        s += k;
    }
    if (comm) {  // the path's one exchange step (ray_tracing.rs:119-131: the gather into the single image)
        FATAL(rt_allreduce_accum(comm, (float*)d_accum, floats, nullptr));
        FATAL(rt_allreduce_stats(comm, (uint64_t*)d_stats, nullptr));
    }
    FATAL(rt_stream_synchronize(nullptr));
    if (g == 0) {
        sh->seconds_render = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        FATAL(rt_device_alloc(&d_rgb, floats));
        FATAL(rt_resolve((const float*)d_accum, o.width, o.height, o.spp, (uint8_t*)d_rgb, nullptr));  // ray_tracing.rs:112-117,126-128
        sh->rgb.resize(floats);
        FATAL(rt_copy_to_host(sh->rgb.data(), d_rgb, floats, nullptr));
        FATAL(rt_copy_to_host(sh->stats, d_stats, sizeof sh->stats, nullptr));
        FATAL(rt_device_free(d_rgb));
    }
    FATAL(rt_device_free(d_accum));
    FATAL(rt_device_free(d_stats));
    if (comm) FATAL(rt_comm_destroy(comm));
    FATAL(rt_scene_destroy(scene));
}

}  // namespace

int main(int argc, char** argv) {
    const Options o = parse(argc, argv);
    const auto t0 = std::chrono::steady_clock::now();
    std::string image = o.image;
    if (image.empty()) {  // texture.rs:70 opens "images/earthmap.jpg" relative to the working directory; here: the raster shipped beside the program
        std::error_code ec;
        const std::filesystem::path exe = std::filesystem::read_symlink("/proc/self/exe", ec);
        const std::filesystem::path beside = exe.parent_path() / "assets" / "earthmap.ppm";
        if (!ec && std::filesystem::exists(beside, ec)) image = beside.string();
    }
    RthScene* hs = rth_scene_build(o.scene.c_str(), (double)o.width / (double)o.height, o.seed, image.empty() ? nullptr : image.c_str());
    if (!hs) die(std::string("scene: ") + rth_last_error());
    const double seconds_build = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    Shared sh;
    sh.opt = &o;
    sh.desc = rth_scene_desc(hs);
    // every GPU is checked here, one after the other, before any thread waits for the others in a collective
    for (int g = 0; g < o.gpus; g++)
        if (rt_init(g) != RT_OK) die("GPU " + std::to_string(g) + ": " + rt_last_error());
    if (o.gpus > 1 && rt_comm_unique_id(sh.comm_id) != RT_OK) die(std::string("rt_comm_unique_id: ") + rt_last_error());
    std::vector<std::thread> pool;
    for (int g = 1; g < o.gpus; g++) pool.emplace_back(gpu_thread, &sh, g);
    gpu_thread(&sh, 0);
    for (std::thread& t : pool) t.join();
    rth_scene_free(hs);
    if (sh.failed) {
        for (int g = 0; g < o.gpus; g++)
            if (!sh.error[g].empty()) std::fprintf(stderr, "rt_render: GPU %d: %s\n", g, sh.error[g].c_str());
        return 1;
    }
    if (!o.out.empty()) {
        std::error_code ec;
        const std::filesystem::path dir = std::filesystem::absolute(o.out).parent_path();
        std::filesystem::create_directories(dir, ec);  // (ray_tracing.rs:133 writes into output/)
        if (!write_png(o.out, sh.rgb.data(), o.width, o.height)) die("cannot write " + o.out);
    }
    const double samples = (double)o.width * o.height * o.spp;
    std::printf("{\"scene\": \"%s\", \"seed\": %llu, \"width\": %u, \"height\": %u, \"spp\": %u, \"max_depth\": %d, \"precision\": \"%s\", "
                "\"n_gpus\": %d, \"seconds_scene_build\": %.4f, \"seconds_render\": %.4f, \"msamples_per_s\": %.3f, \"casts_per_sample\": %.4f, "
                "\"nan_retries\": %llu, \"dropped_samples\": %llu, \"out\": \"%s\"}\n",
                o.scene.c_str(), (unsigned long long)o.seed, o.width, o.height, o.spp, o.depth, o.precision == RT_F64 ? "f64" : "f32", o.gpus,
                seconds_build, sh.seconds_render, samples / sh.seconds_render / 1e6,
                sh.stats[0] ? (double)sh.stats[1] / (double)sh.stats[0] : 0.0, (unsigned long long)sh.stats[2], (unsigned long long)sh.stats[3],
                o.out.c_str());
    return 0;
}
