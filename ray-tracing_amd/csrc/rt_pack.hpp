// rt_pack.hpp — HOST side of the device boundary: validates an RtSceneDesc (include/rt_hip.h) and
// converts it into the device record layouts of rt_device.hpp (in host vectors; rt_hip.hip uploads
// them).  No HIP calls here, so tests/hostsim can run the same packing on the CPU.
#pragma once
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_device.hpp"

namespace rt {

inline std::string& pack_error() { static thread_local std::string e; return e; }
inline int fail(int code, const std::string& msg) { pack_error() = msg; return code; }

template <class T> struct PackedScene {
    std::vector<DPrim<T>> prims;
    std::vector<DMaterial<T>> materials;
    std::vector<DTexture<T>> textures;
    std::vector<DPerlin<T>> perlins;
    std::vector<DXform<T>> xforms;
    std::vector<DInstance<T>> world, lights;
    std::vector<DBvhEntry<T>> bvh;        // the reference's trees, threaded (exact boxes)
    std::vector<int32_t> bvh_root_entry;
    std::vector<DBvhLite> bvh_lite;
    std::vector<DBvhLite> fast_lite;      // the device's own trees over the same leaves (SAH), threaded
    std::vector<int32_t> fast_root_entry;
    std::vector<uint32_t> world_order;    // DScene::world_order (identity beyond the zone)
    DScene<T> view;  // scalars filled by pack_scene; pointers by whoever owns the memory
};

// Checker textures nest by index (texture.rs:21-50).  Depth and "needs (u,v)" are computed once per texture index
// (memoised): a shared sub-texture is not re-expanded once per path leading to it, so a nested chain costs O(n), not 2^n.
struct TextureInfo {
    std::vector<int> depth;     // 0 = leaf kinds; 99 = cyclic, out of range or deeper than 16
    std::vector<char> needs_uv;
};
inline void texture_info_visit(const RtSceneDesc* d, uint32_t tex, TextureInfo& ti, std::vector<char>& state) {
    if (state[tex] == 2) return;
    if (state[tex] == 1) { ti.depth[tex] = 99; return; }  // on the current descent: a cycle
    state[tex] = 1;
    const RtTexture& t = d->textures[tex];
    int depth = 0; bool uv = t.kind == RT_TEX_CHECKER_UV || t.kind == RT_TEX_IMAGE;
    if (t.kind == RT_TEX_CHECKER_3D || t.kind == RT_TEX_CHECKER_UV) {
        if (t.odd >= d->n_textures || t.even >= d->n_textures) {
            depth = 99;
        } else {
            texture_info_visit(d, t.odd, ti, state);
            texture_info_visit(d, t.even, ti, state);
            const int a = ti.depth[t.odd], b = ti.depth[t.even];
            depth = (a > b ? a : b) + 1;
            if (depth > 16) depth = 99;
            uv = uv || ti.needs_uv[t.odd] || ti.needs_uv[t.even];
        }
    }
    ti.depth[tex] = depth; ti.needs_uv[tex] = uv ? 1 : 0;
    state[tex] = 2;
}
inline TextureInfo texture_info(const RtSceneDesc* d) {
    TextureInfo ti;
    ti.depth.assign(d->n_textures, 0); ti.needs_uv.assign(d->n_textures, 0);
    std::vector<char> state(d->n_textures, 0);
    for (uint32_t i = 0; i < d->n_textures; i++) texture_info_visit(d, i, ti, state);
    return ti;
}

inline int validate_instances(const RtSceneDesc* d, const RtInstance* ins, uint32_t n, const char* what) {
    for (uint32_t i = 0; i < n; i++) {
        const RtInstance& in = ins[i];
        std::string at = std::string(what) + "[" + std::to_string(i) + "]";
        if (in.geom_kind == RT_GEOM_PRIM) {
            if (in.geom_index >= d->n_prims) return fail(RT_ERR_INVALID, at + ": primitive index out of range");
        } else if (in.geom_kind == RT_GEOM_BVH) {
            if (in.geom_index >= d->bvh.n_nodes) return fail(RT_ERR_INVALID, at + ": BVH root out of range");
        } else {
            return fail(RT_ERR_INVALID, at + ": unknown geom_kind");
        }
        if ((uint64_t)in.xform_begin + in.xform_count > d->n_xforms) return fail(RT_ERR_INVALID, at + ": xform range out of range");
        if (in.is_medium) {
            if (in.phase_material >= d->n_materials) return fail(RT_ERR_INVALID, at + ": phase material out of range");
            if (!(in.density > 0.0)) return fail(RT_ERR_INVALID, at + ": medium density must be > 0");
        }
    }
    return RT_OK;
}

inline int validate_light_ranges(const RtSceneDesc* d);
inline int validate(const RtSceneDesc* d) {
    if (!d) return fail(RT_ERR_INVALID, "scene description is NULL");
    if (d->abi_version != RT_ABI_VERSION) return fail(RT_ERR_INVALID, "abi_version mismatch");
    // (an empty world is a valid ObjectList, src/objects.rs:38-48: every ray returns the background)
    if ((d->n_prims && !d->prims) || (d->n_materials && !d->materials) || (d->n_textures && !d->textures) ||
        (d->n_perlins && !d->perlins) || (d->n_images && !d->images) || (d->n_xforms && !d->xforms) || (d->n_world && !d->world) ||
        (d->n_lights && !d->lights))
        return fail(RT_ERR_INVALID, "NULL array with non-zero count");
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const RtTexture& t = d->textures[i];
        if (t.kind > RT_TEX_IMAGE) return fail(RT_ERR_INVALID, "texture kind out of range");
        if (t.kind == RT_TEX_NOISE && t.table >= d->n_perlins) return fail(RT_ERR_INVALID, "perlin index out of range");
        if (t.kind == RT_TEX_IMAGE && t.table >= d->n_images) return fail(RT_ERR_INVALID, "image index out of range");
    }
    {
        const TextureInfo ti = texture_info(d);
        for (uint32_t i = 0; i < d->n_textures; i++)
            if (ti.depth[i] > 16) return fail(RT_ERR_UNSUPPORTED, "checker textures nested deeper than 16 (or cyclic / out of range)");
    }
    for (uint32_t i = 0; i < d->n_images; i++)
        if (!d->images[i].rgb || d->images[i].width == 0 || d->images[i].height == 0) return fail(RT_ERR_INVALID, "empty image");
    for (uint32_t i = 0; i < d->n_perlins; i++)
        for (int k = 0; k < 256; k++)
            if ((uint32_t)d->perlins[i].perm_x[k] > 255u || (uint32_t)d->perlins[i].perm_y[k] > 255u || (uint32_t)d->perlins[i].perm_z[k] > 255u)
                return fail(RT_ERR_INVALID, "perlin permutation entry outside 0..255");
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const RtMaterial& m = d->materials[i];
        if (m.kind > RT_MAT_FROSTED_DIELECTRIC) return fail(RT_ERR_INVALID, "material kind out of range");
        bool textured = m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC;
        if (textured && m.texture >= d->n_textures) return fail(RT_ERR_INVALID, "material texture index out of range");
    }
    for (uint32_t i = 0; i < d->n_prims; i++) {
        if (d->prims[i].kind > RT_PRIM_CUBOID) return fail(RT_ERR_INVALID, "primitive kind out of range");
        if (d->prims[i].material >= d->n_materials) return fail(RT_ERR_INVALID, "primitive material index out of range");
    }
    for (uint32_t i = 0; i < d->n_xforms; i++)
        if (d->xforms[i].kind > RT_XFORM_ROTATE_Y) return fail(RT_ERR_INVALID, "xform kind out of range");
    const RtBvh& b = d->bvh;
    if (b.n_nodes) {
        if (!b.min_x || !b.min_y || !b.min_z || !b.max_x || !b.max_y || !b.max_z || !b.left || !b.right)
            return fail(RT_ERR_INVALID, "NULL BVH array");
        // a TREE (bvh.rs:79-83: children are owned boxes): every internal node has at most one parent, so threading
        // the nodes cannot expand one node several times (n nodes with left = right = i+1 would give 2^n entries)
        std::vector<char> has_parent(b.n_nodes, 0);
        for (uint32_t i = 0; i < b.n_nodes; i++) {
            const int32_t kids[2] = {b.left[i], b.right[i]};
            for (int k = 0; k < 2; k++) {
                if (kids[k] >= 0) {  // pre-order: a child node always follows its parent -> acyclic
                    if ((uint32_t)kids[k] <= i || (uint32_t)kids[k] >= b.n_nodes) return fail(RT_ERR_INVALID, "BVH child node index not in pre-order range");
                    if (has_parent[kids[k]]) return fail(RT_ERR_INVALID, "BVH node referenced by more than one parent (the node buffer must hold trees)");
                    has_parent[kids[k]] = 1;
                } else if ((uint32_t)(~kids[k]) >= d->n_prims) {
                    return fail(RT_ERR_INVALID, "BVH leaf primitive index out of range");
                }
            }
        }
        // an instance root that is also some node's child would be threaded twice (harmless, but bounded: <= n_world + n_lights copies)
    }
    int rc = validate_instances(d, d->world, d->n_world, "world");
    if (rc) return rc;
    if ((rc = validate_instances(d, d->lights, d->n_lights, "lights"))) return rc;
    return validate_light_ranges(d);
}

// SoA nodes -> threaded pre-order entries (internal + leaf), see rt_device.hpp DBvhEntry.
template <class T>
void thread_bvh(const RtBvh& b, int32_t node, std::vector<DBvhEntry<T>>& out, std::vector<int32_t>& root_entry) {
    // explicit stack: (node, state) — state 0 = emit node, 1 = after left, 2 = after right
    struct Frame { int32_t node; int state; size_t entry; };
    std::vector<Frame> st;
    st.push_back({node, 0, 0});
    root_entry[node] = (int32_t)out.size();
    auto emit_leaf = [&](int32_t child) {
        DBvhEntry<T> e;
        for (int k = 0; k < 3; k++) { e.mn[k] = T(0); e.mx[k] = T(0); }
        e.prim = ~child;
        e.skip = (int32_t)out.size() + 1;
        out.push_back(e);
    };
    while (!st.empty()) {
        Frame& f = st.back();
        if (f.state == 0) {
            DBvhEntry<T> e;
            e.mn[0] = (T)b.min_x[f.node]; e.mn[1] = (T)b.min_y[f.node]; e.mn[2] = (T)b.min_z[f.node];
            e.mx[0] = (T)b.max_x[f.node]; e.mx[1] = (T)b.max_y[f.node]; e.mx[2] = (T)b.max_z[f.node];
            e.prim = -1; e.skip = 0;
            f.entry = out.size();
            out.push_back(e);
            f.state = 1;
            int32_t l = b.left[f.node];
            if (l >= 0) { st.push_back({l, 0, 0}); } else emit_leaf(l);
        } else if (f.state == 1) {
            f.state = 2;
            int32_t r = b.right[f.node];
            if (r >= 0) { st.push_back({r, 0, 0}); } else emit_leaf(r);
        } else {
            out[f.entry].skip = (int32_t)out.size();
            st.pop_back();
        }
    }
}

// Conservative f64 box of a primitive over the ray times [time_lo, time_hi] (spheres: centre -+ |r|; moving spheres:
// hull of the two end positions, padded for the rounding of the centre arithmetic; rects padded by 1e-4 on the thin
// axis as objects.rs:245-250 does).  Used for the leaf entries' cull boxes and for the device's own trees.
inline void prim_cull_box(const RtPrim& p, double time_lo, double time_hi, double mn[3], double mx[3]) {
    const double big = 1e300;
    switch (p.kind) {
        case RT_PRIM_SPHERE: {
            double r = std::fabs(p.d[3]);
            for (int k = 0; k < 3; k++) { mn[k] = p.d[k] - r; mx[k] = p.d[k] + r; }
            break;
        }
        case RT_PRIM_MOVING_SPHERE: {
            double r = std::fabs(p.d[8]), den = p.d[7] - p.d[6];
            for (int k = 0; k < 3; k++) {
                double c0 = p.d[k] + (p.d[3 + k] - p.d[k]) * ((time_lo - p.d[6]) / den);
                double c1 = p.d[k] + (p.d[3 + k] - p.d[k]) * ((time_hi - p.d[6]) / den);
                double pad = 1e-9 * (std::fabs(c0) + std::fabs(c1) + r + 1.0);  // centre arithmetic rounding
                mn[k] = std::fmin(c0, c1) - r - pad; mx[k] = std::fmax(c0, c1) + r + pad;
                if (!(mn[k] == mn[k]) || !(mx[k] == mx[k])) { mn[k] = -big; mx[k] = big; }
            }
            break;
        }
        case RT_PRIM_CUBOID:
            for (int k = 0; k < 3; k++) { mn[k] = std::fmin(p.d[k], p.d[3 + k]); mx[k] = std::fmax(p.d[k], p.d[3 + k]); }
            break;
        default: {  // rects: a1,a2,b1,b2,k
            int axis = (int)p.kind - RT_PRIM_RECT_XY;  // 0: XY (k on z), 1: XZ (k on y), 2: YZ (k on x)
            int ia = axis == 2 ? 1 : 0, ib = axis == 0 ? 1 : 2, ik = 2 - axis;
            mn[ia] = std::fmin(p.d[0], p.d[1]); mx[ia] = std::fmax(p.d[0], p.d[1]);
            mn[ib] = std::fmin(p.d[2], p.d[3]); mx[ib] = std::fmax(p.d[2], p.d[3]);
            mn[ik] = p.d[4] - 0.0001; mx[ik] = p.d[4] + 0.0001;
        }
    }
    for (int k = 0; k < 3; k++)
        if (!std::isfinite(mn[k]) || !std::isfinite(mx[k])) { mn[k] = -big; mx[k] = big; }
}

// The device's own tree over the leaves of one reference tree: surface-area heuristic (full sweep up to 32 leaves,
// 16 bins above), written as RtBvh-style SoA rows in pre-order so that thread_bvh() serves both kinds of tree.
// The reference builds with a random split axis and a median split (bvh.rs:85-111); a ray visits ~30 % fewer nodes
// of a SAH tree (measured with tools/sim_counters.py: random_scene 32.9 -> 23.8, final_scene 21.5 -> 14.9 per cast).
// WHICH primitive is hit does not depend on the tree: see walk_fast() in rt_device.hpp.
struct SahTree {
    std::vector<double> box[6];
    std::vector<int32_t> left, right;
    int32_t add() { for (auto& b : box) b.push_back(0.0); left.push_back(0); right.push_back(0); return (int32_t)left.size() - 1; }
};
struct SahBox {
    double mn[3], mx[3];
    void reset() { for (int k = 0; k < 3; k++) { mn[k] = 1e308; mx[k] = -1e308; } }
    void grow(const double* a, const double* b) { for (int k = 0; k < 3; k++) { mn[k] = std::fmin(mn[k], a[k]); mx[k] = std::fmax(mx[k], b[k]); } }
    double area() const {
        double e[3];
        for (int k = 0; k < 3; k++) { e[k] = mx[k] - mn[k]; if (!(e[k] > 0.0)) e[k] = 0.0; if (e[k] > 1e150) e[k] = 1e150; }
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
    }
};
struct SahLeaf { int32_t prim, ref_parent; double mn[3], mx[3], c[3]; };
// eye: the camera's origin in the tree's object space.  The threaded layout visits a node's children in ONE fixed order for every ray;
// the child nearer to the eye goes first, so that camera rays (38 % of random_scene's casts) find their closest hit early and cull
// the rest of the tree with it.  Any order gives the same hit (rt_device.hpp: walk_fast).
inline int32_t sah_build(std::vector<SahLeaf>& leaves, size_t lo, size_t hi, SahTree& out, int depth, const double* eye) {
    // leaves[lo, hi), hi - lo >= 2; returns the node index
    const int32_t me = out.add();
    SahBox all; all.reset();
    SahBox cb; cb.reset();
    for (size_t i = lo; i < hi; i++) { all.grow(leaves[i].mn, leaves[i].mx); cb.grow(leaves[i].c, leaves[i].c); }
    for (int k = 0; k < 3; k++) { out.box[k][me] = all.mn[k]; out.box[3 + k][me] = all.mx[k]; }
    const size_t n = hi - lo;
    size_t mid = lo + n / 2;
    int best_axis = -1;
    if (depth < 48) {
        double best = 1e308;
        if (n <= 32) {  // full sweep on every axis
            std::vector<SahLeaf> tmp(leaves.begin() + lo, leaves.begin() + hi), keep;
            std::vector<double> right_area(n);
            for (int ax = 0; ax < 3; ax++) {
                std::stable_sort(tmp.begin(), tmp.end(), [ax](const SahLeaf& a, const SahLeaf& b) { return a.c[ax] < b.c[ax]; });
                SahBox b; b.reset();
                for (size_t i = n; i-- > 1;) { b.grow(tmp[i].mn, tmp[i].mx); right_area[i] = b.area(); }
                b.reset();
                for (size_t i = 1; i < n; i++) {
                    b.grow(tmp[i - 1].mn, tmp[i - 1].mx);
                    const double cost = b.area() * (double)i + right_area[i] * (double)(n - i);
                    if (cost < best) { best = cost; best_axis = ax; mid = lo + i; keep = tmp; }
                }
            }
            if (best_axis >= 0) std::copy(keep.begin(), keep.end(), leaves.begin() + lo);
        } else {        // 16 bins on the centroid range of every axis
            constexpr int NB = 16;
            double best_split = 0.0;
            for (int ax = 0; ax < 3; ax++) {
                const double c0 = cb.mn[ax], c1 = cb.mx[ax];
                if (!(c1 > c0)) continue;
                const double scale = NB / (c1 - c0);
                SahBox bins[NB]; size_t cnt[NB] = {0};
                for (auto& b : bins) b.reset();
                for (size_t i = lo; i < hi; i++) {
                    int bi = (int)((leaves[i].c[ax] - c0) * scale);
                    bi = bi < 0 ? 0 : (bi >= NB ? NB - 1 : bi);
                    bins[bi].grow(leaves[i].mn, leaves[i].mx); cnt[bi]++;
                }
                double ra[NB]; size_t rc[NB];
                SahBox b; b.reset(); size_t c = 0;
                for (int k = NB - 1; k >= 1; k--) { if (cnt[k]) b.grow(bins[k].mn, bins[k].mx); c += cnt[k]; ra[k] = c ? b.area() : 0.0; rc[k] = c; }
                b.reset(); c = 0;
                for (int k = 1; k < NB; k++) {
                    if (cnt[k - 1]) b.grow(bins[k - 1].mn, bins[k - 1].mx);
                    c += cnt[k - 1];
                    if (c == 0 || rc[k] == 0) continue;
                    const double cost = b.area() * (double)c + ra[k] * (double)rc[k];
                    if (cost < best) { best = cost; best_axis = ax; best_split = c0 + (double)k / scale; }
                }
            }
            if (best_axis >= 0) {
                const int ax = best_axis;
                // recompute bin membership exactly as above for the partition
                const double c0 = cb.mn[ax], scale = NB / (cb.mx[ax] - c0);
                const int ksplit = (int)std::lround((best_split - c0) * scale);
                auto it = std::stable_partition(leaves.begin() + lo, leaves.begin() + hi, [&](const SahLeaf& l) {
                    int bi = (int)((l.c[ax] - c0) * scale);
                    bi = bi < 0 ? 0 : (bi >= NB ? NB - 1 : bi);
                    return bi < ksplit;
                });
                mid = (size_t)(it - leaves.begin());
                if (mid == lo || mid == hi) { best_axis = -1; mid = lo + n / 2; }
            }
        }
    }
    if (best_axis < 0) {  // all centroids equal / too deep: median split in the given order
        mid = lo + n / 2;
    }
#ifndef RT_SAH_NO_EYE_ORDER
    if (eye) {
        auto dist2 = [&](size_t a, size_t b) {
            SahBox bx; bx.reset();
            for (size_t i = a; i < b; i++) bx.grow(leaves[i].mn, leaves[i].mx);
            double s2 = 0.0;
            for (int k = 0; k < 3; k++) {
                const double dk = eye[k] < bx.mn[k] ? bx.mn[k] - eye[k] : (eye[k] > bx.mx[k] ? eye[k] - bx.mx[k] : 0.0);
                s2 += dk * dk;
            }
            return s2;
        };
        if (dist2(mid, hi) < dist2(lo, mid)) {  // (NaN compares false: order kept)
            std::rotate(leaves.begin() + lo, leaves.begin() + mid, leaves.begin() + hi);
            mid = lo + (hi - mid);
        }
    }
#endif
    // children in pre-order: node, left subtree, right subtree (RtBvh convention)
    // (a leaf child is the complement of its final position in `leaves`: ranges of one element do not move any more)
    const int32_t l = (mid - lo == 1) ? ~(int32_t)lo : sah_build(leaves, lo, mid, out, depth + 1, eye);
    const int32_t r = (hi - mid == 1) ? ~(int32_t)mid : sah_build(leaves, mid, hi, out, depth + 1, eye);
    out.left[me] = l; out.right[me] = r;
    return me;
}

template <class T> void convert_instances(const RtInstance* ins, uint32_t n, const std::vector<DPrim<T>>& prims, std::vector<DInstance<T>>& out) {
    for (uint32_t i = 0; i < n; i++) {
        DInstance<T> o;
        std::memset(&o, 0, sizeof o);
        if (ins[i].geom_kind == RT_GEOM_PRIM) o.prim = prims[ins[i].geom_index];  // (rt_device.hpp: DInstance)
        o.density = (T)ins[i].density;
        o.geom_kind = ins[i].geom_kind; o.geom_index = ins[i].geom_index;
        o.xform_begin = ins[i].xform_begin; o.xform_count = ins[i].xform_count;
        o.is_medium = ins[i].is_medium ? 1u : 0u; o.phase_material = ins[i].phase_material;
        out.push_back(o);
    }
}


// Light primitives that Object::random samples by gen_range, and the camera's shutter interval, need a range the
// generator can serve: rand 0.7.3's gen_range panics on low >= high and on a non-finite width in the reference; here both
// are validation errors — in BOTH arithmetic modes, because the device redraws until a value falls below `high`
// (rt_device.hpp: Rng::gen_range) and would never stop on a range that is empty or not finite after rounding to f32.
inline bool samplable_range(double lo, double hi) {
    const float flo = (float)lo, fhi = (float)hi;
    return lo < hi && std::isfinite(hi - lo) && flo < fhi && std::isfinite(fhi - flo);  // (false for NaN; +-inf fail the widths)
}
inline int validate_light_ranges(const RtSceneDesc* d) {
    const char* why = " has an empty or non-finite range (gen_range needs low < high, both finite, also after rounding to f32)";
    for (uint32_t i = 0; i < d->n_lights; i++) {
        const RtInstance& in = d->lights[i];
        if (in.is_medium || in.geom_kind != RT_GEOM_PRIM) continue;
        const RtPrim& p = d->prims[in.geom_index];
        if (p.kind >= RT_PRIM_RECT_XY && p.kind <= RT_PRIM_RECT_YZ) {
            if (!samplable_range(p.d[0], p.d[1]) || !samplable_range(p.d[2], p.d[3])) return fail(RT_ERR_INVALID, std::string("light rect") + why);
        } else if (p.kind == RT_PRIM_CUBOID) {
            for (int k = 0; k < 3; k++)
                if (!samplable_range(p.d[k], p.d[3 + k])) return fail(RT_ERR_INVALID, std::string("light cuboid") + why);
        }
    }
    if (!samplable_range(d->camera.t1, d->camera.t2))
        return fail(RT_ERR_INVALID, "camera shutter needs t1 < t2, both finite, also after rounding to f32 (gen_range, src/camera.rs:63)");
    return RT_OK;
}

// Which code paths the scene can reach (rt_device.hpp feature mask).
inline uint32_t scene_features(const RtSceneDesc* d) {
    uint32_t f = 0;
    auto scan = [&](const RtInstance* ins, uint32_t n, bool world) {
        for (uint32_t i = 0; i < n; i++) {
            if (world && ins[i].geom_kind == RT_GEOM_BVH) f |= F_BVH;
            if (world && ins[i].is_medium) f |= F_MEDIUM;
            if (world && ins[i].is_medium && ins[i].geom_kind == RT_GEOM_BVH) f |= F_MEDIUM_BVH;
            if (ins[i].xform_count) f |= F_XFORM;
        }
    };
    scan(d->world, d->n_world, true);
    scan(d->lights, d->n_lights, false);
    for (uint32_t i = 0; i < d->n_prims; i++) {
        switch (d->prims[i].kind) {
            case RT_PRIM_SPHERE: f |= F_SPHERE; break;
            case RT_PRIM_MOVING_SPHERE: f |= F_MOVING; break;
            case RT_PRIM_CUBOID: f |= F_CUBOID; break;
            default: f |= F_RECT; break;
        }
    }
    const TextureInfo ti = texture_info(d);
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const RtMaterial& m = d->materials[i];
        if (m.kind == RT_MAT_METAL) f |= F_METAL;
        if (m.kind == RT_MAT_DIELECTRIC) f |= F_DIELECTRIC;
        if (m.kind == RT_MAT_FROSTED_DIELECTRIC) f |= F_DIELECTRIC | F_FROSTED;
        if (m.kind == RT_MAT_ISOTROPIC) f |= F_MEDIUM;
        bool textured = m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC;
        if (textured && ti.needs_uv[m.texture]) f |= F_UV;
    }
    for (uint32_t i = 0; i < d->n_textures; i++) {
        if (d->textures[i].kind != RT_TEX_SOLID) f |= F_TEXTURED;
        if (d->textures[i].kind == RT_TEX_NOISE) f |= F_NOISE;
    }
    if (d->n_lights) f |= F_LIGHTS;
    return f;
}

// RtSceneDesc (f64) -> device records in T.  `d` must have passed validate().
template <class T> void pack_scene(const RtSceneDesc* d, PackedScene<T>& h) {
    for (uint32_t i = 0; i < d->n_prims; i++) {
        DPrim<T> p;
        for (int k = 0; k < 10; k++) p.d[k] = (T)d->prims[i].d[k];
        p.kind = d->prims[i].kind; p.material = d->prims[i].material;
        h.prims.push_back(p);
    }
    const TextureInfo ti = texture_info(d);
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const RtMaterial& m = d->materials[i];
        DMaterial<T> o;
        for (int k = 0; k < 3; k++) o.albedo[k] = (T)m.albedo[k];
        o.fuzz = (T)m.fuzz; o.ref_idx = (T)m.ref_idx; o.texture = m.texture;
        bool textured = m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC;
        o.kind = m.kind | ((textured && ti.needs_uv[m.texture]) ? 0x100u : 0u);  // bit 8: shading needs (u,v)
        h.materials.push_back(o);
    }
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const RtTexture& t = d->textures[i];
        DTexture<T> o;
        for (int k = 0; k < 3; k++) o.color[k] = (T)t.color[k];
        o.scale = (T)t.scale; o.kind = t.kind; o.odd = t.odd; o.even = t.even; o.table = t.table;
        h.textures.push_back(o);
    }
    for (uint32_t i = 0; i < d->n_perlins; i++) {
        const RtPerlin& p = d->perlins[i];
        DPerlin<T> o;
        for (int k = 0; k < 256; k++) {
            o.perm[0][k] = (uint8_t)p.perm_x[k]; o.perm[1][k] = (uint8_t)p.perm_y[k]; o.perm[2][k] = (uint8_t)p.perm_z[k];
            for (int c = 0; c < 3; c++) o.ranvec[k][c] = (T)p.ranvec[k][c];
            if (DPerlin<T>::kStride > 3) o.ranvec[k][DPerlin<T>::kStride - 1] = T(0);
        }
        h.perlins.push_back(o);
    }
    for (uint32_t i = 0; i < d->n_xforms; i++) {
        DXform<T> o;
        for (int k = 0; k < 3; k++) o.v[k] = (T)d->xforms[i].v[k];
        o.kind = d->xforms[i].kind;
        h.xforms.push_back(o);
    }
    convert_instances<T>(d->world, d->n_world, h.prims, h.world);
    convert_instances<T>(d->lights, d->n_lights, h.prims, h.lights);
    h.bvh_root_entry.assign(d->bvh.n_nodes, -1);
    auto add_roots = [&](const RtInstance* ins, uint32_t n) {
        for (uint32_t i = 0; i < n; i++)
            if (ins[i].geom_kind == RT_GEOM_BVH && h.bvh_root_entry[ins[i].geom_index] < 0)
                thread_bvh<T>(d->bvh, (int32_t)ins[i].geom_index, h.bvh, h.bvh_root_entry);
    };
    add_roots(d->world, d->n_world);
    add_roots(d->lights, d->n_lights);
    // f32 copies of the boxes, rounded outward so that they contain the exact boxes; a leaf gets the box of its
    // primitive (computed from the f64 description; moving spheres over the time hull below)
    const double time_lo = std::fmin(0.0, d->camera.t1), time_hi = std::fmax(1.0, d->camera.t2);
    auto down = [](double x) { float f = (float)x; return ((double)f > x) ? std::nextafterf(f, -INFINITY) : f; };
    auto up = [](double x) { float f = (float)x; return ((double)f < x) ? std::nextafterf(f, INFINITY) : f; };
    auto lite_of = [&](const DBvhEntry<T>& e, const double* nmn, const double* nmx) {
        DBvhLite l;
        double mn[3], mx[3];
        if (e.prim < 0) { for (int k = 0; k < 3; k++) { mn[k] = nmn[k]; mx[k] = nmx[k]; } }
        else prim_cull_box(d->prims[e.prim], time_lo, time_hi, mn, mx);
        for (int k = 0; k < 3; k++) { l.mn[k] = down(mn[k]); l.mx[k] = up(mx[k]); }
        l.skip = e.skip; l.prim = e.prim;
        return l;
    };
    for (const DBvhEntry<T>& e : h.bvh) {
        double mn[3], mx[3];
        for (int k = 0; k < 3; k++) { mn[k] = (double)e.mn[k]; mx[k] = (double)e.mx[k]; }
        h.bvh_lite.push_back(lite_of(e, mn, mx));
    }
    // the device's own trees: one per reference tree that an instance uses, over the same leaves
    // A leaf entry of such a tree carries, in `skip` (a leaf's successor is always the next entry), the entry of its
    // PARENT NODE IN THE REFERENCE'S TREE: walk_fast validates a hit against that node's exact box.
    h.fast_root_entry.assign(d->bvh.n_nodes, -1);
    {
        SahTree sah;
        std::vector<SahLeaf> all_leaves;
        std::vector<std::pair<uint32_t, int32_t>> roots;  // (reference root node, SAH root node)
        auto build_for = [&](const RtInstance* ins, uint32_t n) {
            for (uint32_t i = 0; i < n; i++) {
                if (ins[i].geom_kind != RT_GEOM_BVH) continue;
                const uint32_t ref_root = ins[i].geom_index;
                bool seen = false;
                for (auto& r : roots) seen |= r.first == ref_root;
                if (seen) continue;
                const size_t first = all_leaves.size();
                const int32_t e0 = h.bvh_root_entry[ref_root], e1 = h.bvh[e0].skip;
                std::vector<int32_t> open;  // internal entries whose subtree is not finished: the innermost is the parent
                for (int32_t e = e0; e < e1; e++) {
                    while (!open.empty() && h.bvh[open.back()].skip <= e) open.pop_back();
                    if (h.bvh[e].prim < 0) { open.push_back(e); continue; }
                    SahLeaf l; l.prim = h.bvh[e].prim; l.ref_parent = open.back();
                    prim_cull_box(d->prims[l.prim], time_lo, time_hi, l.mn, l.mx);
                    for (int k = 0; k < 3; k++) l.c[k] = 0.5 * (l.mn[k] + l.mx[k]);
                    all_leaves.push_back(l);
                }
                // (a tree's root is an internal node with two children: >= 2 leaves)
                // the camera's origin in this instance's object space (transforms.rs:15-17, 90-97: outermost wrapper first)
                double eye[3] = {d->camera.origin[0], d->camera.origin[1], d->camera.origin[2]};
                for (uint32_t j = 0; j < ins[i].xform_count; j++) {
                    const RtXform& x = d->xforms[ins[i].xform_begin + j];
                    if (x.kind == RT_XFORM_TRANSLATE) { for (int k = 0; k < 3; k++) eye[k] -= x.v[k]; }
                    else { const double sn = x.v[0], cs = x.v[1], ex = eye[0], ez = eye[2]; eye[0] = cs * ex - sn * ez; eye[2] = sn * ex + cs * ez; }
                }
                roots.push_back({ref_root, sah_build(all_leaves, first, all_leaves.size(), sah, 0, eye)});
            }
        };
        build_for(d->world, d->n_world);
        build_for(d->lights, d->n_lights);
        if (!roots.empty()) {
            RtBvh fb;
            fb.n_nodes = (uint32_t)sah.left.size();
            fb.min_x = sah.box[0].data(); fb.min_y = sah.box[1].data(); fb.min_z = sah.box[2].data();
            fb.max_x = sah.box[3].data(); fb.max_y = sah.box[4].data(); fb.max_z = sah.box[5].data();
            fb.left = sah.left.data(); fb.right = sah.right.data();
            std::vector<DBvhEntry<double>> entries;
            std::vector<int32_t> root_entry(fb.n_nodes, -1);
            for (auto& r : roots) {
                thread_bvh<double>(fb, r.second, entries, root_entry);
                h.fast_root_entry[r.first] = root_entry[r.second];
            }
            for (const DBvhEntry<double>& e : entries) {
                DBvhEntry<T> et; et.prim = e.prim; et.skip = e.skip;
                if (e.prim >= 0) { et.prim = all_leaves[e.prim].prim; et.skip = all_leaves[e.prim].ref_parent; }
                h.fast_lite.push_back(lite_of(et, e.mn, e.mx));
            }
        }
    }
    // Ray times at which every moving sphere of every reference tree lies inside ALL its ancestors' boxes (with a slack far
    // below the node classification's own margin): walk_run's geometric node culling is valid exactly there (rt_device.hpp:
    // walk_begin).  Boxes built by the reference for (0, 1) give [0, 1]; trees without moving spheres give every time.
    double cull_lo = -1.7976931348623157e308, cull_hi = 1.7976931348623157e308;
    bool nested = true;
    {
        std::vector<int32_t> open;
        for (int32_t e = 0; e < (int32_t)h.bvh.size(); e++) {
            while (!open.empty() && h.bvh[open.back()].skip <= e) open.pop_back();
            const DBvhEntry<T>& en = h.bvh[e];
            if (en.prim < 0) {
                // walk_fast vouches a hit by the loose test of ONE reference node and lets the nesting of the boxes speak for the
                // ancestors: a description whose child box is not inside its parent's (never the reference's own trees:
                // surrounding_box, bvh.rs:11-25) gets no fast walk
                if (!open.empty())
                    for (int k = 0; k < 3; k++)
                        if (!((double)en.mn[k] >= (double)h.bvh[open.back()].mn[k] && (double)en.mx[k] <= (double)h.bvh[open.back()].mx[k])) nested = false;
                open.push_back(e);
                continue;
            }
            const RtPrim& p = d->prims[en.prim];
            if (p.kind != RT_PRIM_MOVING_SPHERE) {  // a primitive that does not move must lie inside every ancestor box, at all times
                double pmn[3], pmx[3];
                prim_cull_box(p, 0.0, 1.0, pmn, pmx);
                for (int32_t a : open)
                    for (int k = 0; k < 3; k++) {
                        const double mn = (double)h.bvh[a].mn[k], mx = (double)h.bvh[a].mx[k];
                        const double slack = 1e-7 * (std::fabs(mn) + std::fabs(mx) + 1.0);
                        if (!(pmn[k] >= mn - slack && pmx[k] <= mx + slack)) { cull_lo = 1.0; cull_hi = 0.0; }
                    }
                continue;
            }
            const double r = std::fabs(p.d[8]), den = p.d[7] - p.d[6], t0 = p.d[6];
            for (int32_t a : open) {
                for (int k = 0; k < 3; k++) {
                    const double mn = (double)h.bvh[a].mn[k], mx = (double)h.bvh[a].mx[k];
                    const double slack = 1e-7 * (std::fabs(mn) + std::fabs(mx) + r + 1.0);
                    const double v = (p.d[3 + k] - p.d[k]) / den;            // centre(t) = c1 + v (t - t0)
                    const double L = mn - slack + r - p.d[k], U = mx + slack - r - p.d[k];  // L <= v (t - t0) <= U
                    double lo, hi;
                    if (v > 0.0) { lo = t0 + L / v; hi = t0 + U / v; }
                    else if (v < 0.0) { lo = t0 + U / v; hi = t0 + L / v; }
                    else if (v == 0.0 && L <= 0.0 && 0.0 <= U) continue;      // does not move on this axis and fits
                    else { lo = 1.0; hi = 0.0; }                             // never inside (or NaN): no time qualifies
                    if (!(lo <= hi)) { lo = 1.0; hi = 0.0; }
                    cull_lo = std::fmax(cull_lo, lo); cull_hi = std::fmin(cull_hi, hi);
                }
            }
        }
        if (!(cull_lo <= cull_hi)) { cull_lo = 1.0; cull_hi = 0.0; }  // empty: no ray time qualifies
    }
    DScene<T>& out = h.view;
    out.time_lo = (T)time_lo; out.time_hi = (T)time_hi;
    const double tmax = sizeof(T) == 4 ? 3.4028234663852886e38 : 1.7976931348623157e308;  // (stay inside T's range)
    out.node_cull_lo = (T)std::fmax(cull_lo, -tmax); out.node_cull_hi = (T)std::fmin(cull_hi, tmax);
    out.fast_lo = nested ? out.time_lo : T(1); out.fast_hi = nested ? out.time_hi : T(0);
    out.n_bvh = (uint32_t)h.fast_lite.size();
    out.n_bvh_lds = 0;
    out.bvh_lds = nullptr;
    out.n_perlins = d->n_perlins;
    out.n_perlin_lds = 0;
    out.perlin_lds = nullptr;
    out.tex_cache = nullptr;
    out.n_world = d->n_world;
    {   // the zone: leading run of the world list before its first medium; plain primitives are tested first, trees last (stable)
        uint32_t nz = 0;
        while (nz < d->n_world && !d->world[nz].is_medium) nz++;
        h.world_order.clear();
        for (uint32_t i = 0; i < nz; i++) if (d->world[i].geom_kind != RT_GEOM_BVH) h.world_order.push_back(i);
        for (uint32_t i = 0; i < nz; i++) if (d->world[i].geom_kind == RT_GEOM_BVH) h.world_order.push_back(i);
        bool identity = true;
        for (uint32_t i = 0; i < nz; i++) identity &= h.world_order[i] == i;
        for (uint32_t i = nz; i < d->n_world; i++) h.world_order.push_back(i);
#ifdef RT_NO_ZONE_ORDER
        identity = true;
#endif
        out.n_zone = identity ? 0u : nz;
    }
    out.n_materials = d->n_materials;
    out.n_lights = d->n_lights;
    for (int k = 0; k < 3; k++) out.background[k] = (T)d->background[k];
    const RtCamera& c = d->camera;
    for (int k = 0; k < 3; k++) {
        out.cam.origin[k] = (T)c.origin[k]; out.cam.llc[k] = (T)c.lower_left_corner[k];
        out.cam.horizontal[k] = (T)c.horizontal[k]; out.cam.vertical[k] = (T)c.vertical[k];
        out.cam.u[k] = (T)c.u[k]; out.cam.v[k] = (T)c.v[k]; out.cam.w[k] = (T)c.w[k];
    }
    out.cam.lens_radius = (T)c.lens_radius; out.cam.t1 = (T)c.t1; out.cam.t2 = (T)c.t2;
}

}  // namespace rt
