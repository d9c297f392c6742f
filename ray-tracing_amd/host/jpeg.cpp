// jpeg.cpp — baseline JPEG decoder for ImageTexture (the reference opens images/earthmap.jpg through the `image`
// crate -> jpeg-decoder 0.1.20, src/texture.rs:68-74; that crate is not in the reference tree, so this follows the
// published format instead: ITU-T T.81 baseline sequential DCT, Huffman coding, JFIF YCbCr).
// Supported: SOF0, 8-bit samples, 1 (grey) or 3 (YCbCr) components, sampling factors 1 or 2 per axis, restart
// intervals, 8/16-bit quantisation tables.  Not supported (reported as errors): progressive / lossless / arithmetic
// coding, 12-bit, CMYK.  The inverse DCT is evaluated in double precision from its definition and rounded once;
// chroma of subsampled files is replicated (box filter).  Decoders legitimately differ by +-1..2 LSB in the IDCT /
// upsampling / colour rounding; nothing in the reference pins texel values (parity unpinned).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "scene_api.hpp"

namespace rth {
namespace {

struct Huff {
    // canonical code lengths -> (mincode, maxcode, valptr) per length, T.81 F.2.2.3
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    bool present = false;
};
struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int pred = 0;
    int stride = 0, rows = 0;          // padded plane size in samples
    std::vector<uint8_t> plane;
};
const int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Reader {
    const uint8_t* d;
    size_t n, pos = 0;
    uint32_t bits = 0;
    int nbits = 0;
    bool hit_marker = false;
    Reader(const uint8_t* p, size_t len) : d(p), n(len) {}
    int bit() {
        if (nbits == 0) {
            int b = 0;
            if (!hit_marker && pos < n) {
                b = d[pos];
                if (b == 0xFF) {
                    if (pos + 1 < n && d[pos + 1] == 0x00) pos += 2;       // stuffed byte
                    else { hit_marker = true; b = 0; }                      // a marker: feed zeros, leave it in place
                } else {
                    pos++;
                }
            }
            bits = (uint32_t)b;
            nbits = 8;
        }
        nbits--;
        return (int)((bits >> nbits) & 1u);
    }
    int receive(int s) { int v = 0; for (int i = 0; i < s; i++) v = (v << 1) | bit(); return v; }
    void align() { nbits = 0; }
};
int extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }
int decode_symbol(Reader& r, const Huff& h, bool& ok) {
    int code = 0;
    for (int len = 1; len <= 16; len++) {
        code = (code << 1) | r.bit();
        if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.valptr[len] + code - h.mincode[len]];
    }
    ok = false;
    return 0;
}
// 8x8 inverse DCT from the definition (T.81 A.3.3), separable, in double
void idct8x8(const double* in, uint8_t* out, int stride) {
    struct Table {
        double c[8][8];
        Table() {
            for (int x = 0; x < 8; x++)
                for (int u = 0; u < 8; u++) c[x][u] = (u == 0 ? std::sqrt(0.5) : 1.0) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
        }
    };
    static const Table table;  // function-local static: initialised once, thread-safe (two host threads may decode at once)
    const double (*c)[8] = table.c;
    double tmp[64];
    for (int v = 0; v < 8; v++)          // rows: over u
        for (int x = 0; x < 8; x++) {
            double s = 0;
            for (int u = 0; u < 8; u++) s += c[x][u] * in[v * 8 + u];
            tmp[v * 8 + x] = s;
        }
    for (int x = 0; x < 8; x++)
        for (int y = 0; y < 8; y++) {
            double s = 0;
            for (int v = 0; v < 8; v++) s += c[y][v] * tmp[v * 8 + x];
            double p = std::floor(s / 4.0 + 128.0 + 0.5);
            out[y * stride + x] = (uint8_t)(p < 0 ? 0 : (p > 255 ? 255 : p));
        }
}
uint8_t clamp8(double v) { v = std::floor(v + 0.5); return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

}  // namespace

bool decode_jpeg(const uint8_t* data, size_t n, RgbImage& out, std::string& err) {
    auto bad = [&](const std::string& m) { err = "jpeg: " + m; return false; };
    if (n < 4 || data[0] != 0xFF || data[1] != 0xD8) return bad("no SOI marker");
    uint16_t qt[4][64];
    bool qt_ok[4] = {false, false, false, false};
    Huff hdc[4], hac[4];
    std::vector<Comp> comps;
    int width = 0, height = 0, restart = 0;
    size_t pos = 2;
    bool have_sof = false;
    for (;;) {
        while (pos < n && data[pos] != 0xFF) pos++;
        while (pos < n && data[pos] == 0xFF) pos++;
        if (pos >= n) return bad("no scan found");
        const int m = data[pos++];
        if (m == 0xD9) return bad("EOI before any scan");
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (pos + 2 > n) return bad("truncated segment");
        const size_t len = ((size_t)data[pos] << 8) | data[pos + 1];
        if (len < 2 || pos + len > n) return bad("truncated segment");
        const uint8_t* s = data + pos + 2;
        const size_t sl = len - 2;
        if (m == 0xDB) {  // DQT
            size_t i = 0;
            while (i < sl) {
                const int pq = s[i] >> 4, tq = s[i] & 15;
                i++;
                if (tq > 3 || pq > 1 || i + (pq ? 128u : 64u) > sl) return bad("bad DQT");
                for (int k = 0; k < 64; k++) {
                    qt[tq][kZigzag[k]] = pq ? (uint16_t)((s[i] << 8) | s[i + 1]) : s[i];
                    i += pq ? 2 : 1;
                }
                qt_ok[tq] = true;
            }
        } else if (m == 0xC4) {  // DHT
            size_t i = 0;
            while (i < sl) {
                if (i + 17 > sl) return bad("bad DHT");
                const int tc = s[i] >> 4, th = s[i] & 15;
                if (tc > 1 || th > 3) return bad("bad DHT class/id");
                Huff& h = tc ? hac[th] : hdc[th];
                int counts[17], total = 0;
                for (int l = 1; l <= 16; l++) { counts[l] = s[i + l]; total += counts[l]; }
                i += 17;
                if (total > 256 || i + (size_t)total > sl) return bad("bad DHT length");
                std::memcpy(h.vals, s + i, (size_t)total);
                i += (size_t)total;
                int code = 0, k = 0;
                for (int l = 1; l <= 16; l++) {
                    h.valptr[l] = k;
                    h.mincode[l] = code;
                    code += counts[l];
                    k += counts[l];
                    h.maxcode[l] = counts[l] ? code - 1 : -1;
                    code <<= 1;
                }
                h.present = true;
            }
        } else if (m == 0xC0 || m == 0xC1) {  // SOF0 / SOF1 (extended sequential, Huffman) share the baseline layout at 8 bits
            if (sl < 6 || s[0] != 8) return bad("only 8-bit samples are supported");
            height = (s[1] << 8) | s[2];
            width = (s[3] << 8) | s[4];
            const int nc = s[5];
            if ((nc != 1 && nc != 3) || sl < 6u + 3u * nc || width == 0 || height == 0) return bad("unsupported component count or empty frame");
            comps.assign((size_t)nc, Comp());
            for (int c = 0; c < nc; c++) {
                comps[c].id = s[6 + 3 * c];
                comps[c].h = s[7 + 3 * c] >> 4;
                comps[c].v = s[7 + 3 * c] & 15;
                comps[c].tq = s[8 + 3 * c];
                if (comps[c].h < 1 || comps[c].h > 2 || comps[c].v < 1 || comps[c].v > 2 || comps[c].tq > 3) return bad("unsupported sampling factors");
            }
            have_sof = true;
        } else if (m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8)) {
            return bad("progressive / lossless / arithmetic-coded JPEG is not supported (baseline only)");
        } else if (m == 0xDD) {
            if (sl < 2) return bad("bad DRI");
            restart = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {  // SOS: baseline has one interleaved scan (or one per component for non-interleaved files)
            if (!have_sof) return bad("SOS before SOF");
            if (sl < 1) return bad("truncated SOS segment");
            const int ns = s[0];
            if (ns != (int)comps.size() || sl < 1u + 2u * ns + 3u) return bad("non-interleaved scans are not supported");
            for (int k = 0; k < ns; k++) {
                const int id = s[1 + 2 * k];
                bool found = false;
                for (auto& c : comps)
                    if (c.id == id) { c.td = s[2 + 2 * k] >> 4; c.ta = s[2 + 2 * k] & 15; found = true; }
                if (!found) return bad("scan names an unknown component");
            }
            pos += len;
            break;
        }
        pos += len;
    }
    int hmax = 1, vmax = 1;
    for (auto& c : comps) { if (c.h > hmax) hmax = c.h; if (c.v > vmax) vmax = c.v; }
    if (comps.size() == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }
    const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
    for (auto& c : comps) {
        if (!qt_ok[c.tq] || c.td > 3 || c.ta > 3 || !hdc[c.td].present || !hac[c.ta].present) return bad("scan uses an undefined table");
        c.stride = mcux * c.h * 8;
        c.rows = mcuy * c.v * 8;
        c.plane.assign((size_t)c.stride * c.rows, 0);
    }
    Reader r(data + pos, n - pos);
    int until_restart = restart;
    bool ok = true;
    for (int my = 0; my < mcuy && ok; my++)
        for (int mx = 0; mx < mcux && ok; mx++) {
            if (restart && until_restart == 0) {  // RSTn: byte-align, skip the marker, reset predictors
                r.align();
                r.hit_marker = false;
                if (r.pos + 1 < r.n && r.d[r.pos] == 0xFF && r.d[r.pos + 1] >= 0xD0 && r.d[r.pos + 1] <= 0xD7) r.pos += 2;
                else return bad("restart marker missing");
                for (auto& c : comps) c.pred = 0;
                until_restart = restart;
            }
            for (auto& c : comps)
                for (int by = 0; by < c.v; by++)
                    for (int bx = 0; bx < c.h; bx++) {
                        double blk[64];
                        for (double& b : blk) b = 0.0;
                        const int t = decode_symbol(r, hdc[c.td], ok);
                        if (t > 11) ok = false;
                        if (!ok) break;
                        c.pred += extend(r.receive(t), t);
                        blk[0] = (double)c.pred * qt[c.tq][0];
                        for (int k = 1; k < 64 && ok;) {
                            const int rs = decode_symbol(r, hac[c.ta], ok);
                            const int run = rs >> 4, sz = rs & 15;
                            if (sz == 0) {
                                if (run != 15) break;  // EOB
                                k += 16;
                                continue;
                            }
                            k += run;
                            if (k > 63) { ok = false; break; }
                            blk[kZigzag[k]] = (double)extend(r.receive(sz), sz) * qt[c.tq][kZigzag[k]];
                            k++;
                        }
                        if (!ok) break;
                        idct8x8(blk, c.plane.data() + (size_t)((my * c.v + by) * 8) * c.stride + (size_t)(mx * c.h + bx) * 8, c.stride);
                    }
            if (restart) until_restart--;
        }
    if (!ok) return bad("corrupt entropy-coded data");
    out.width = (uint32_t)width;
    out.height = (uint32_t)height;
    out.rgb.resize((size_t)width * height * 3);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            uint8_t* px = out.rgb.data() + ((size_t)y * width + x) * 3;
            auto at = [&](const Comp& c) { return (double)c.plane[(size_t)(y * c.v / vmax) * c.stride + (size_t)(x * c.h / hmax)]; };
            if (comps.size() == 1) {
                px[0] = px[1] = px[2] = (uint8_t)at(comps[0]);
            } else {  // JFIF: full-range BT.601
                const double Y = at(comps[0]), cb = at(comps[1]) - 128.0, cr = at(comps[2]) - 128.0;
                px[0] = clamp8(Y + 1.402 * cr);
                px[1] = clamp8(Y - 0.344136286 * cb - 0.714136286 * cr);
                px[2] = clamp8(Y + 1.772 * cb);
            }
        }
    return true;
}

}  // namespace rth
