// Sanitizer run of the CPU oracle (test infrastructure): `make -C oracle asan` links this driver and oracle.cpp with
// AddressSanitizer and UndefinedBehaviorSanitizer and runs it. It walks every family of routines oracle.h declares on
// small deterministic inputs chosen to reach the corners where indices are formed: grids cut by the per-cell cap, both
// binning modes, dense exports, origins on the far bounds, degenerate directions, shards whose tiles overhang the frame,
// the step limit, moving boxes, bounces, the BVH with fewer primitives than a leaf holds, empty scenes.
// It checks nothing but a few invariants; the sanitizers are the assertion. Exit code 0 = clean.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "oracle.h"

static uint32_t lcg(uint32_t& s) { return s = s * 1664525u + 1013904223u; }
static float frand(uint32_t& s, float lo, float hi) { return lo + (hi - lo) * (float)(lcg(s) >> 8) / 16777216.0f; }

static int failures = 0;
#define EXPECT(cond)                                                      \
    do {                                                                  \
        if (!(cond)) {                                                    \
            std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                   \
        }                                                                 \
    } while (0)

struct Built {
    orc_grid* g = nullptr;
    std::vector<uint32_t> cell_start, indices;
    orc_scene s{};
};

static void build(Built& b, const std::vector<BoxData>& boxes, const std::vector<TriangleData>& tris, float cell, uint32_t cap, int mode) {
    b.g = orc_grid_build(boxes.data(), (uint32_t)boxes.size(), tris.data(), (uint32_t)tris.size(), cell, cap, mode);
    uint64_t n_cells = 0, n_refs = 0, dropped = 0;
    GridMetadata meta;
    orc_grid_info(b.g, &meta, &n_cells, &n_refs, &dropped);
    b.cell_start.assign(n_cells + 1, 0);
    b.indices.assign(n_refs + 1, 0);
    orc_grid_copy_csr(b.g, b.cell_start.data(), b.indices.data());
    EXPECT(b.cell_start[n_cells] == n_refs);
    std::memset(&b.s, 0, sizeof(b.s));
    b.s.boxes = boxes.data();
    b.s.nb = mode == RT_GRID_TRIANGLES ? 0u : (uint32_t)boxes.size();
    b.s.tris = tris.data();
    b.s.nt = (uint32_t)tris.size();
    b.s.meta = meta;
    b.s.cell_start = b.cell_start.data();
    b.s.indices = b.indices.data();
    b.s.n_cells = n_cells;
}

static std::vector<TriangleData> soup(uint32_t n, uint32_t seed, float extent, float size) {
    std::vector<TriangleData> t(n);
    std::memset(t.data(), 0, n * sizeof(TriangleData));
    for (uint32_t i = 0; i < n; ++i) {
        const float c[3] = {frand(seed, -extent, extent), frand(seed, -extent, extent), frand(seed, -extent, extent)};
        for (int k = 0; k < 3; ++k) {
            t[i].v0[k] = c[k] + frand(seed, -size, size);
            t[i].v1[k] = c[k] + frand(seed, -size, size);
            t[i].v2[k] = c[k] + frand(seed, -size, size);
        }
        if (i % 17 == 3) std::memcpy(&t[i], &t[i - 1], sizeof(TriangleData));  // exact duplicates
        if (i % 29 == 5) std::memcpy(t[i].v2, t[i].v1, sizeof(t[i].v1));        // degenerate
    }
    return t;
}

static std::vector<BoxData> some_boxes(uint32_t n, uint32_t seed) {
    std::vector<BoxData> b(n);
    std::memset(b.data(), 0, n * sizeof(BoxData));
    for (uint32_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) {
            const float c0 = frand(seed, -30, 30), c1 = i + 3 >= n ? c0 + frand(seed, -5, 5) : c0, h = frand(seed, 0.5f, 4.0f);
            b[i].center0[k] = c0, b[i].center1[k] = c1, b[i].half_size[k] = h;
            b[i].min[k] = std::fmin(c0, c1) - h - 0.1f, b[i].max[k] = std::fmax(c0, c1) + h + 0.1f;
        }
    }
    return b;
}

int main() {
    rt_params p;
    std::memset(&p, 0, sizeof(p));
    p.finest_cell_size = 16.0f, p.max_objects_per_cell = 8192, p.max_dda_steps = 200, p.lod_cull = 1, p.moving_box_count = 3;
    p.fov = 0.785398f, p.light_direction[0] = 0.5f, p.light_direction[1] = -1.0f, p.light_direction[2] = 0.3f, p.shadow_offset = 0.001f;
    p.skip_duplicate_triangles = 1;

    // primitive routines
    {
        const float o[3] = {0, 0, 0}, d[3] = {1, 0, 0}, mn[3] = {5, -1, -1}, mx[3] = {10, 1, 1};
        EXPECT(std::fabs(orc_intersect_aabb(o, d, mn, mx, 0) - 5.0f) < 0.01f);
        float v[3];
        for (uint32_t s = 0; s < 100; ++s) orc_random_unit_vector(s, v);
        std::vector<float> bmn(3 * 1000), bmx(3 * 1000), t(100 * 1000), dirs(300), zeros(300, 0.0f);
        orc_particle_aabbs(1000, 0.5f, bmn.data(), bmx.data());
        for (uint32_t s = 0; s < 100; ++s) orc_random_unit_vector(s, &dirs[3 * s]);
        orc_intersect_aabb_pairs(zeros.data(), dirs.data(), 100, bmn.data(), bmx.data(), 1000, 0, t.data());
        orc_intersect_aabb_pairs(zeros.data(), dirs.data(), 100, bmn.data(), bmx.data(), 1000, 1, t.data());
    }

    for (int round = 0; round < 6; ++round) {
        const bool with_boxes = round % 2 == 1;
        const int mode = round == 4 ? RT_GRID_TRIANGLES : RT_GRID_HIERARCHICAL;
        const uint32_t cap = round == 2 ? 3u : round == 4 ? 256u : 8192u;
        const float cell = round == 3 ? 2.5f : 16.0f;
        std::vector<TriangleData> tris = round == 5 ? std::vector<TriangleData>() : soup(400 + 300 * round, 7u + round, 40.0f, round == 2 ? 12.0f : 3.0f);
        std::vector<BoxData> boxes = with_boxes || round == 5 ? some_boxes(round == 5 ? 2 : 9, 99u + round) : std::vector<BoxData>();
        Built b;
        build(b, boxes, tris, cell, cap, mode);
        p.max_objects_per_cell = cap;
        p.max_dda_steps = round == 3 ? 17u : 200u;
        p.wgsl_compat = round & 1;
        // dense exports
        if (b.s.n_cells < 4000) {
            if (mode == RT_GRID_TRIANGLES) {
                std::vector<TriFineCellData> dense(b.s.n_cells);
                orc_grid_copy_dense_tri(b.g, dense.data());
            } else {
                std::vector<FineCellData> dense(b.s.n_cells);
                orc_grid_copy_dense(b.g, dense.data());
            }
        }
        std::vector<uint8_t> coarse(1 << 20);
        orc_grid_coarse_counts(b.g, coarse.data(), coarse.size());

        CameraUniform cam;
        const float pos[3] = {round == 1 ? 0.0f : 60.0f, 45.0f, 90.0f};
        orc_camera_to_uniform(pos, 3.5f, -0.4f, 0.37f * round, 50.0f, p.fov, round == 2, &cam);
        const uint32_t w = 77, h = 50;
        for (int sharded = 0; sharded < 3; ++sharded) {
            const uint32_t owner[5] = {0, 1, 1, 0, 1};
            rt_shard sh;
            std::memset(&sh, 0, sizeof(sh));
            sh.tile_w = 32, sh.tile_h = 16, sh.first = 1, sh.stride = sharded == 2 ? 5u : 3u;
            sh.owner = sharded == 2 ? owner : nullptr;
            const rt_shard* shp = sharded ? &sh : nullptr;
            const uint64_t n = orc_shard_pixels(w, h, shp);
            std::vector<rt_hit> prim(n + 1), shad(n + 1);
            std::vector<float> so(3 * n + 3), sd(3 * n + 3);
            rt_stats st;
            orc_trace_primary(&b.s, &p, &cam, w, h, shp, prim.data(), shad.data(), so.data(), sd.data(), &st, 2);
            EXPECT(st.rays >= st.primary_hits);
            std::vector<uint8_t> pic(4 * n + 4);
            rt_render_opts ro;
            std::memset(&ro, 0, sizeof(ro));
            ro.max_bounces = 1 + round, ro.shadows = round & 1, ro.debug_enabled = sharded == 0, ro.debug_pixel[0] = 5, ro.debug_pixel[1] = 7;
            orc_render(&b.s, &p, &cam, w, h, shp, nullptr, 0, &ro, pic.data(), &st, 2);
        }
        // ray lists: random, axis-aligned and zero directions, origins on the bounds and far outside
        {
            uint32_t seed = 1234u + round;
            const uint32_t n = 600;
            std::vector<float> o(3 * n), d(3 * n);
            for (uint32_t i = 0; i < n; ++i) {
                for (int k = 0; k < 3; ++k) {
                    o[3 * i + k] = i % 7 == 0 ? b.s.meta.bounds_max[k] : i % 11 == 0 ? b.s.meta.bounds_min[k] : frand(seed, -120, 120);
                    d[3 * i + k] = frand(seed, -1, 1);
                }
                if (i % 5 == 0) d[3 * i] = 0.0f;
                if (i % 13 == 0) d[3 * i + 1] = d[3 * i + 2] = 0.0f;
                const float len = std::sqrt(d[3 * i] * d[3 * i] + d[3 * i + 1] * d[3 * i + 1] + d[3 * i + 2] * d[3 * i + 2]);
                if (len > 0) for (int k = 0; k < 3; ++k) d[3 * i + k] /= len;
            }
            std::vector<rt_hit> out(n), brute(n);
            rt_stats st;
            orc_trace_rays(&b.s, &p, &cam, o.data(), d.data(), n, out.data(), 0, &st, 2);
            orc_trace_rays(&b.s, &p, &cam, o.data(), d.data(), n, out.data(), 1, &st, 1);
            orc_trace_rays_bruteforce(&b.s, &p, &cam, o.data(), d.data(), n, brute.data(), 2);
            RayDebugInfo info;
            orc_debug_pixel(&b.s, &p, &cam, w, h, 40, 20, &info);
            // the reference's CPU path on the same rays
            if (!tris.empty()) {
                orc_bvh* bvh = orc_bvh_build(tris.data(), (uint32_t)tris.size());
                uint64_t nodes, leaves, depth, prims;
                orc_bvh_stats(bvh, &nodes, &leaves, &depth, &prims);
                EXPECT(prims == tris.size());
                std::vector<float> t(n);
                orc_bvh_trace(bvh, tris.data(), o.data(), d.data(), n, t.data(), 2);
                orc_linear_scan(tris.data(), (uint32_t)tris.size(), o.data(), d.data(), n, t.data(), 2);
                orc_bvh_free(bvh);
            }
        }
        orc_grid_free(b.g);
    }
    // generators, tiny BVHs, pyramid
    {
        std::vector<TriangleData> t(1500);
        orc_generate_test_triangles(t.size(), t.data());
        std::vector<float> o(3 * 500), d(3 * 500), out(500);
        orc_generate_test_rays(500, o.data(), d.data());
        for (uint32_t n : {1u, 3u, 5u, 1500u}) {
            orc_bvh* bvh = orc_bvh_build(t.data(), n);
            orc_bvh_trace(bvh, t.data(), o.data(), d.data(), 500, out.data(), 1);
            orc_bvh_free(bvh);
        }
        TriangleData pyr[6];
        orc_pyramid_triangles(pyr);
        const uint32_t all[6] = {0, 1, 2, 3, 4, 5};
        float out6[6];
        EXPECT(orc_batch_intersect_triangles(&o[0], &d[0], pyr, all, 6, out6) >= -1);
    }
    if (failures) return 1;
    std::puts("oracle sanitizer run: clean");
    return 0;
}
