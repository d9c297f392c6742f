// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU f64 restatement of reference src/texture.rs (SolidColor, CheckerTextureUV, CheckerTexture,
// NoiseTexture, ImageTexture, Perlin).
// Third-party arithmetic not in /root/reference: JPEG decode (image 0.23.7 -> jpeg-decoder
// 0.1.20, texture.rs:70-72).  The oracle reads an already-decoded RGB8 raster (binary PPM made
// by tools/decode_earthmap.py with Pillow); texel values may differ from jpeg-decoder's by
// +-1-2 LSB — "parity unpinned" for texel values, absorbed by the 1 % RMSE image bar.
#pragma once
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "vec3.hpp"

namespace orc {

// Rust `as i32` / `as u32`: truncating, SATURATING float->int casts; NaN -> 0.
inline int32_t rust_as_i32(double x) {
    if (x != x) return 0;
    if (x >= 2147483647.0) return INT32_MAX;
    if (x <= -2147483648.0) return INT32_MIN;
    return (int32_t)x;
}
inline uint32_t rust_as_u32(double x) {
    if (x != x) return 0;
    if (x >= 4294967295.0) return UINT32_MAX;
    if (x <= 0.0) return 0;
    return (uint32_t)x;
}

// texture.rs:7-9
struct Texture {
    virtual ~Texture() {}
    virtual Vec3 value(double u, double v, Vec3 p) const = 0;
};
using TexPtr = std::shared_ptr<Texture>;

// texture.rs:11-19
struct SolidColor : Texture {
    Vec3 color;
    explicit SolidColor(Vec3 c) : color(c) {}
    Vec3 value(double, double, Vec3) const override { return color; }
};

// texture.rs:21-34
struct CheckerTextureUV : Texture {
    TexPtr odd, even;
    CheckerTextureUV(TexPtr o, TexPtr e) : odd(std::move(o)), even(std::move(e)) {}
    Vec3 value(double u, double v, Vec3 p) const override {
        if (rust_as_i32(u / 0.01) % 2 == rust_as_i32(v / 0.01) % 2) return odd->value(u, v, p);
        return even->value(u, v, p);
    }
};

// texture.rs:36-50
struct CheckerTexture : Texture {
    TexPtr odd, even;
    CheckerTexture(TexPtr o, TexPtr e) : odd(std::move(o)), even(std::move(e)) {}
    Vec3 value(double u, double v, Vec3 p) const override {
        double sines = std::sin(10.0 * p.x) * std::sin(10.0 * p.y) * std::sin(10.0 * p.z);
        if (sines < 0.0) return odd->value(u, v, p);
        return even->value(u, v, p);
    }
};

// texture.rs:100-177.  perm/ranvec come from four entropy-seeded generators in the reference
// (texture.rs:122,131); the oracle draws them from the seeded scene generator in the order
// perm_x, perm_y, perm_z, ranvec.  shuffle = rand 0.7.3 SliceRandom::shuffle (Fisher-Yates from
// the end: for i in (1..len).rev() { swap(i, gen_range(0, i+1)) }).
struct Perlin {
    int32_t perm_x[256], perm_y[256], perm_z[256];
    Vec3 ranvec[256];
    static void generate_perm(int32_t* p, Rng& rng) {
        for (int i = 0; i < 256; i++) p[i] = i;
        for (int i = 255; i >= 1; i--) {
            uint32_t j = rng.gen_index((uint32_t)i + 1);
            std::swap(p[i], p[j]);
        }
    }
    explicit Perlin(Rng& rng) {
        generate_perm(perm_x, rng);
        generate_perm(perm_y, rng);
        generate_perm(perm_z, rng);
        for (int i = 0; i < 256; i++) ranvec[i] = Vec3::random(-1.0, 1.0, rng).unit();
    }
    // texture.rs:139-165
    double noise(Vec3 p) const {
        double u = p.x - std::floor(p.x);
        double v = p.y - std::floor(p.y);
        double w = p.z - std::floor(p.z);
        int32_t i = rust_as_i32(std::floor(p.x));
        int32_t j = rust_as_i32(std::floor(p.y));
        int32_t k = rust_as_i32(std::floor(p.z));
        double uu = u * u * (3.0 - 2.0 * u);
        double vv = v * v * (3.0 - 2.0 * v);
        double ww = w * w * (3.0 - 2.0 * w);
        double accum = 0.0;
        for (int di = 0; di < 2; di++)
            for (int dj = 0; dj < 2; dj++)
                for (int dk = 0; dk < 2; dk++) {
                    // wrapping add then & 255 on two's complement (texture.rs:153-155)
                    Vec3 c = ranvec[perm_x[(uint32_t)(i + di) & 255u] ^ perm_y[(uint32_t)(j + dj) & 255u] ^
                                    perm_z[(uint32_t)(k + dk) & 255u]];
                    accum += ((double)di * uu + (1.0 - (double)di) * (1.0 - uu)) *
                             ((double)dj * vv + (1.0 - (double)dj) * (1.0 - vv)) *
                             ((double)dk * ww + (1.0 - (double)dk) * (1.0 - ww)) *
                             (c * Vec3(u - (double)di, v - (double)dj, w - (double)dk));
                }
        return accum;
    }
    // texture.rs:166-176
    double turb(Vec3 p, int depth) const {
        double accum = 0.0;
        Vec3 temp_p = p;
        double weight = 1.0;
        for (int d = 0; d < depth; d++) {
            accum += weight * noise(temp_p);
            weight *= 0.5;
            temp_p *= 2.0;
        }
        return std::fabs(accum);
    }
};

// texture.rs:52-61
struct NoiseTexture : Texture {
    std::shared_ptr<Perlin> noise;
    double scale;
    NoiseTexture(std::shared_ptr<Perlin> n, double s) : noise(std::move(n)), scale(s) {}
    Vec3 value(double, double, Vec3 p) const override {
        return Vec3::ones() * 0.5 * (1.0 + std::sin(scale * p.z + 10.0 * noise->turb(p, 7)));
    }
};

// Decoded RGB8 raster, row 0 = top of the file (image::RgbImage layout).
struct RgbImage {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;
    static std::shared_ptr<RgbImage> load_ppm(const std::string& path) {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("ImageTexture:failed to open file " + path);
        auto img = std::make_shared<RgbImage>();
        int maxv = 0;
        if (std::fscanf(f, "P6 %u %u %d", &img->width, &img->height, &maxv) != 3 || maxv != 255) {
            std::fclose(f);
            throw std::runtime_error("ImageTexture: not a binary 8-bit PPM: " + path);
        }
        std::fgetc(f);  // single whitespace after maxval
        img->rgb.resize((size_t)img->width * img->height * 3);
        size_t n = std::fread(img->rgb.data(), 1, img->rgb.size(), f);
        std::fclose(f);
        if (n != img->rgb.size()) throw std::runtime_error("ImageTexture: short read: " + path);
        return img;
    }
};

// texture.rs:63-98
struct ImageTexture : Texture {
    std::shared_ptr<RgbImage> img;
    explicit ImageTexture(std::shared_ptr<RgbImage> i) : img(std::move(i)) {}
    Vec3 value(double u, double v, Vec3) const override {
        // num::clamp(x, lo, hi): if x < lo {lo} else if x > hi {hi} else {x}  (NaN passes through)
        u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
        v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        uint32_t x = rust_as_u32(u * (double)img->width);
        if (x > img->width - 1) x = img->width - 1;
        uint32_t y = rust_as_u32(v * (double)img->height);
        if (y > img->height - 1) y = img->height - 1;
        double color_scale = 1.0 / 255.0;
        const uint8_t* px = &img->rgb[((size_t)(img->height - 1 - y) * img->width + x) * 3];
        return Vec3((double)px[0] * color_scale, (double)px[1] * color_scale, (double)px[2] * color_scale);
    }
};

}  // namespace orc
