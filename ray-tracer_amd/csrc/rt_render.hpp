// Shaded picture: the rest of the reference's compute shader around trace_ray -
// sky colour and lighting (src/raytracer_unified.wgsl:590-621), the reflection-bounce loop, the debug-pixel
// highlight and the texture store of main (:655-720), box colours (:230-233) and the untextured part of the
// triangle material (:280-303, :345-364).
//
// The alpha MASK / BLEND rejection inside the triangle test (:303-316) happens in the trace: for materials with a
// constant alpha it is folded into the reference records when materials are set (k_emit_records in rt_build.hip),
// for a textured alpha trace_ray<TEX> samples per hit. TEX here: materials name textures, so the winning hit's
// material is evaluated at its interpolated uv (base-colour texture, normal map, emissive texture: rt_texture.hpp).
//
// Same numerics contract as the trace: the reference's f32 expression trees, no contraction, host-evaluated
// transcendental constants. The 8-bit store is round-to-nearest-even of clamp(c, 0, 1) * 255 (NaN -> 0).
#pragma once

#include "rt_trace.hpp"

namespace rt {

struct DevShade {
    const float4* boxes;      // BoxData as six float4: [2] = (color.rgb, reflectivity)
    const uint32_t* tri_mat;  // material index per triangle: u32(TriangleData::material_id)
    const float4* materials;  // MaterialData as four float4: (base_color) (emissive, texture) (metallic, roughness, ..) (..)
    uint32_t n_materials;
    uint32_t max_bounces;
    uint32_t shadows;         // 1: a hit whose shadow ray is occluded gets no diffuse light (not in the shader)
    uint32_t debug_enabled, debug_x, debug_y;
    uint32_t rgb8;              // 1: 3-byte pixels (no alpha byte)
    uint32_t box_frame_stride;  // float4 records between the per-frame box records of consecutive frames
    size_t frame_pixels;        // output records per frame
};


__device__ inline uint32_t unorm8(float c) {
    float x = fminf(fmaxf(c, 0.0f), 1.0f);  // fmaxf(NaN, 0) = 0
    return (uint32_t)rintf(x * 255.0f);
}

// raytracer_unified.wgsl:419-428; WGSL's % on f32 is the truncated remainder (fmodf)
__device__ inline bool near_grid_boundary(const DevScene& s, V3 pos, float threshold) {
    const V3 rel = pos - v3(s.bmin[0], s.bmin[1], s.bmin[2]);
    const float cs = s.cell_size;
    const V3 local = v3(fmodf(rel.x, cs) / cs, fmodf(rel.y, cs) / cs, fmodf(rel.z, cs) / cs);
    const float dx = fminf(local.x, 1.0f - local.x), dy = fminf(local.y, 1.0f - local.y), dz = fminf(local.z, 1.0f - local.z);
    return dx < threshold || dy < threshold || dz < threshold;
}

#ifndef RT_RENDER_WAVES
#define RT_RENDER_WAVES 1
#endif
#ifndef RT_RENDER_TEX_SINGLE_WAVES
#define RT_RENDER_TEX_SINGLE_WAVES 6  // waves per SIMD the textured single-bounce instantiation is held to
#endif
// SINGLE: max_bounces == 1 (direct lighting only): no bounce loop, so the ray origin stays the camera position in
// scalar registers and nothing is carried from one traversal to the next.
template <bool WGSL, bool BOXES, bool TEX = false, bool WIDE = false, bool SINGLE = false>
__global__ __launch_bounds__(256, SINGLE ? (TEX ? RT_RENDER_TEX_SINGLE_WAVES : 8) : RT_RENDER_WAVES) void k_render(DevScene s, DevFrames fs, ShardMap m, DevShade sh, uint32_t* __restrict__ out,
                                                unsigned long long* stats, const uint32_t* __restrict__ cull_masks, uint32_t cull_words) {
    const DevFrame& f = fs.f[blockIdx.y];
    const float show_grid = fs.show_grid[blockIdx.y];
    s.box_frame += (size_t)blockIdx.y * sh.box_frame_stride;
    const size_t frame_first = (size_t)blockIdx.y * sh.frame_pixels;
    const uint32_t lane = threadIdx.x & 63, w4 = threadIdx.x >> 6;
    const uint32_t blk_x = (m.sub_x + 1) / 2, blk_y = (m.sub_y + 1) / 2;
    const uint32_t blk_per_tile = blk_x * blk_y;
    const uint32_t lt = blockIdx.x / blk_per_tile, blk = blockIdx.x % blk_per_tile;
    if (lt >= m.n_local) return;
    const uint32_t sub_x = (blk % blk_x) * 2 + (w4 & 1), sub_y = (blk / blk_x) * 2 + (w4 >> 1);
    const uint32_t lx = sub_x * 8 + (lane & 7), ly = sub_y * 8 + (lane >> 3);
    const uint32_t tile = shard_tile(m, lt);
    const uint32_t px = (tile % m.tiles_x) * m.tile_w + lx, py = (tile / m.tiles_x) * m.tile_h + ly;
    const bool in_tile = lx < m.tile_w && ly < m.tile_h;
    const bool live = in_tile && px < f.width && py < f.height;
    // A clear bit in the frame's block mask (k_block_mask in rt_cull.hpp): no camera ray of this wave's 8x8 pixel block
    // meets an occupied cell, so every pixel of it shows the sky. The same in every lane of the wave.
    bool may_hit = true;
    if (cull_masks) {
        const uint32_t bx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((tile % m.tiles_x) * m.tile_w >> 3) + sub_x));
        const uint32_t by = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((tile / m.tiles_x) * m.tile_h >> 3) + sub_y));
        const uint32_t blocks_x = (f.width + 7u) >> 3, blocks_y = (f.height + 7u) >> 3;
        if (bx < blocks_x && by < blocks_y) {
            const uint32_t word = cull_masks[(size_t)blockIdx.y * cull_words + (size_t)by * ((blocks_x + 31u) >> 5) + (bx >> 5)];
            may_hit = ((word >> (bx & 31u)) & 1u) != 0u;
        }
    }

    uint32_t rgba = 0;
    Counters ctr{};
    // rays, primary hits, shadow rays, shadow hits of this pixel: at most 8 bounces, so a byte each in one register
    uint32_t counts = 0;
    if (live) {
        V3 o = v3(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]);
        V3 d = primary_dir(f, px, py);
        V3 acc = v3(0.0f, 0.0f, 0.0f);
        float mult = 1.0f;
        const V3 to_light = f.shadow.d;  // -normalize(LIGHT_DIRECTION)
        for (uint32_t bounce = 0; bounce < (SINGLE ? 1u : sh.max_bounces); ++bounce) {
            const RayDir rd = make_ray_dir<WGSL>(d, s.cell_size, s.px, s.pxy);
            HitRec h;
            counts += 1u;
            if (!(may_hit || bounce != 0) || !trace_ray<WGSL, false, 0, BOXES, TEX, WIDE>(s, o, rd, h, ctr)) {
                // sky (:591-596)
                const float t = (d.y + 1.0f) * 0.5f;
                acc = acc + mix(v3(0.3f, 0.5f, 0.7f), v3(0.5f, 0.7f, 1.0f), t) * mult;
                break;
            }
            if (bounce == 0) counts += 1u << 8;
            // Only what the shadow ray needs is evaluated before it (position, normal); colour, emissive and
            // reflectivity are fetched afterwards, so they do not occupy registers during the second traversal.
            const V3 pos = o + d * h.t;
            V3 n = hit_normal(s, h, pos);
            const bool is_box = BOXES && h.ref == MISS;
            uint32_t mat = MISS;
            F2 uv{0.0f, 0.0f};
            if (!is_box) {
                mat = sh.tri_mat[h.id - s.nb];
                if (mat >= sh.n_materials) mat = MISS;
                if (TEX && mat != MISS) {
                    const float4 c0 = s.refs[3 * (size_t)h.ref], c1 = s.refs[3 * (size_t)h.ref + 1],
                                 c2 = s.refs[3 * (size_t)h.ref + 2];
                    float bu, bv;
                    hit_barycentrics(o, d, v3(c0.x, c0.y, c0.z), v3(c1.x, c1.y, c1.z), v3(c2.x, c2.y, c2.z), bu, bv);
                    uv = hit_uv(s.tex, h.id - s.nb, bu, bv);
                    n = mapped_normal(s.tex, mat, uv, n);
                }
            }
            // lighting (:598-603)
            float diffuse = fmaxf(dot(n, to_light), 0.0f);
            if (sh.shadows) {
                V3 nf = n;
                if (dot(nf, d) > 0.0f) nf = neg(nf);
                HitRec oh;
                counts += 0x10001u;
                if (trace_ray<WGSL, true, 0, BOXES, TEX, WIDE>(s, pos + nf * f.shadow_offset, f.shadow, oh, ctr)) {
                    counts += 1u << 24;
                    diffuse = 0.0f;
                }
            }
            V3 color = v3(0.7f, 0.7f, 0.7f), emissive = v3(0.0f, 0.0f, 0.0f);
            float refl = 0.0f;
            if (is_box) {
                const float4 c = sh.boxes[6 * h.id + 2];
                color = v3(c.x, c.y, c.z);
                refl = c.w;
            } else if (mat != MISS) {
                const float4 m0 = sh.materials[4 * mat], m1 = sh.materials[4 * mat + 1], m2 = sh.materials[4 * mat + 2];
                color = v3(m0.x, m0.y, m0.z);
                emissive = v3(m1.x, m1.y, m1.z);
                refl = m2.x;  // metallic (:361)
                if (TEX) {
                    const float4 bc = base_color(s.tex, mat, uv);
                    color = v3(bc.x, bc.y, bc.z);
                    emissive = emissive_color(s.tex, mat, uv);
                }
            }
            V3 lit = color * (0.3f + diffuse * 0.7f) + emissive;
            if (show_grid > 0.5f && near_grid_boundary(s, pos, 0.02f)) lit = mix(lit, v3(0.0f, 1.0f, 0.0f), 0.6f);
            // bounce loop (:670-688)
            acc = acc + (lit * (1.0f - refl)) * mult;
            if (SINGLE || refl < 0.01f) break;
            mult *= refl;
            d = d - n * (2.0f * dot(n, d));  // reflect(e1, e2) = e1 - 2 * dot(e2, e1) * e2
            o = pos + n * 0.001f;
        }
        // debug pixel (:693-716)
        if (sh.debug_enabled) {
            const int dx = (int)px - (int)sh.debug_x, dy = (int)py - (int)sh.debug_y;
            if (dx == 0 && dy == 0)
                acc = v3(1.0f, 1.0f, 0.0f);
            else if (dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1)
                acc = mix(acc, v3(1.0f, 1.0f, 0.0f), 0.5f);
        }
        rgba = unorm8(acc.x) | (unorm8(acc.y) << 8) | (unorm8(acc.z) << 16) | 0xFF000000u;
    }
    if (in_tile) {
        const size_t oi = frame_first + (size_t)lt * m.tile_w * m.tile_h + (size_t)ly * m.tile_w + lx;
        if (sh.rgb8) {
            uint8_t* o3 = reinterpret_cast<uint8_t*>(out) + 3 * oi;
            o3[0] = (uint8_t)rgba, o3[1] = (uint8_t)(rgba >> 8), o3[2] = (uint8_t)(rgba >> 16);
        } else {
            out[oi] = rgba;
        }
    }
    if (stats) flush_stats(stats, counts & 255u, (counts >> 8) & 255u, (counts >> 16) & 255u, counts >> 24, ctr);
}

// k_render for pictures with reflection bounces of triangle-only scenes. Camera
// rays and their shadow rays are traced as k_render traces them: neighbouring pixels' rays share their cells. From
// the first reflection on a wave holds a few scattered rays in unrelated cells, and a ray inside dense geometry scans
// long lists alone while the other lanes wait; those traversals go through trace_ray_repack, which all 64 lanes
// enter - with or without a ray of their own - and share the rays' candidate lists among them. The bounce loop is
// therefore the same in every lane of the wave; `going` says whether this lane's path is still followed. Per lane the
// arithmetic is k_render's, operation for operation.
// ALPHA: some material's alpha is a texture's, so the traversals sample it per hit (trace_ray<TEX>).
#ifndef RT_RENDER_DEAL_WAVES
#define RT_RENDER_DEAL_WAVES 4  // waves per SIMD the kernel is held to
#endif
template <bool WGSL, bool TEX, bool ALPHA = false>
__global__ __launch_bounds__(256, RT_RENDER_DEAL_WAVES) void k_render_deal(DevScene s, DevFrames fs, ShardMap m, DevShade sh, uint32_t* __restrict__ out,
                                                     const uint32_t* __restrict__ cull_masks, uint32_t cull_words) {
    __shared__ RepackLds repack[4];
    const DevFrame& f = fs.f[blockIdx.y];
    const float show_grid = fs.show_grid[blockIdx.y];
    const size_t frame_first = (size_t)blockIdx.y * sh.frame_pixels;
    const uint32_t lane = threadIdx.x & 63, w4 = threadIdx.x >> 6;
    const uint32_t blk_x = (m.sub_x + 1) / 2, blk_y = (m.sub_y + 1) / 2;
    const uint32_t blk_per_tile = blk_x * blk_y;
    const uint32_t lt = blockIdx.x / blk_per_tile, blk = blockIdx.x % blk_per_tile;
    if (lt >= m.n_local) return;
    const uint32_t sub_x = (blk % blk_x) * 2 + (w4 & 1), sub_y = (blk / blk_x) * 2 + (w4 >> 1);
    const uint32_t lx = sub_x * 8 + (lane & 7), ly = sub_y * 8 + (lane >> 3);
    const uint32_t tile = shard_tile(m, lt);
    const uint32_t px = (tile % m.tiles_x) * m.tile_w + lx, py = (tile / m.tiles_x) * m.tile_h + ly;
    const bool in_tile = lx < m.tile_w && ly < m.tile_h;
    const bool live = in_tile && px < f.width && py < f.height;
    bool may_hit = true;  // the frame's block mask, as in k_render
    if (cull_masks) {
        const uint32_t bx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((tile % m.tiles_x) * m.tile_w >> 3) + sub_x));
        const uint32_t by = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((tile / m.tiles_x) * m.tile_h >> 3) + sub_y));
        const uint32_t blocks_x = (f.width + 7u) >> 3, blocks_y = (f.height + 7u) >> 3;
        if (bx < blocks_x && by < blocks_y) {
            const uint32_t word = cull_masks[(size_t)blockIdx.y * cull_words + (size_t)by * ((blocks_x + 31u) >> 5) + (bx >> 5)];
            may_hit = ((word >> (bx & 31u)) & 1u) != 0u;
        }
    }
    RepackLds& L = repack[w4 & 3];
    Counters ctr{};
    constexpr uint32_t FIRST_DEALT = 1u;  // camera rays the plain way, every reflection dealt
    V3 o = v3(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]);
    V3 d = primary_dir(f, live ? px : 0u, live ? py : 0u);
    V3 acc = v3(0.0f, 0.0f, 0.0f);
    float mult = 1.0f;
    const V3 to_light = f.shadow.d;  // -normalize(LIGHT_DIRECTION)
    bool going = live;
    for (uint32_t bounce = 0; bounce < sh.max_bounces; ++bounce) {
        if (__builtin_amdgcn_ballot_w64(going) == 0ull) break;  // the same in every lane
        const RayDir rd = make_ray_dir<WGSL>(d, s.cell_size, s.px, s.pxy);
        HitRec h{1e10f, MISS, MISS};
        bool hit = false;
        if (bounce < FIRST_DEALT) {
            if (going && (may_hit || bounce != 0)) hit = trace_ray<WGSL, false, 0, false, ALPHA, false>(s, o, rd, h, ctr);
        } else {
            hit = trace_ray_repack<WGSL, false, 1, ALPHA>(s, o, rd, going, h, L);
        }
        if (going && !hit) {  // sky (:591-596)
            const float t = (d.y + 1.0f) * 0.5f;
            acc = acc + mix(v3(0.3f, 0.5f, 0.7f), v3(0.5f, 0.7f, 1.0f), t) * mult;
            going = false;
        }
        V3 pos = o, n = d;
        uint32_t mat = MISS;
        F2 uv{0.0f, 0.0f};
        if (going) {
            pos = o + d * h.t;
            n = hit_normal(s, h, pos);
            mat = sh.tri_mat[h.id - s.nb];
            if (mat >= sh.n_materials) mat = MISS;
            if (TEX && mat != MISS) {
                const float4 c0 = s.refs[3 * (size_t)h.ref], c1 = s.refs[3 * (size_t)h.ref + 1], c2 = s.refs[3 * (size_t)h.ref + 2];
                float bu, bv;
                hit_barycentrics(o, d, v3(c0.x, c0.y, c0.z), v3(c1.x, c1.y, c1.z), v3(c2.x, c2.y, c2.z), bu, bv);
                uv = hit_uv(s.tex, h.id - s.nb, bu, bv);
                n = mapped_normal(s.tex, mat, uv, n);
            }
        }
        // lighting (:598-603)
        float diffuse = fmaxf(dot(n, to_light), 0.0f);
        if (sh.shadows) {  // the same in every lane
            V3 nf = n;
            if (dot(nf, d) > 0.0f) nf = neg(nf);
            const V3 so = pos + nf * f.shadow_offset;
            HitRec oh;
            bool occluded = false;
            if (bounce < FIRST_DEALT) {
                if (going) occluded = trace_ray<WGSL, true, 0, false, ALPHA, false>(s, so, f.shadow, oh, ctr);
            } else {
                occluded = trace_ray_repack<WGSL, true, 1, ALPHA>(s, so, f.shadow, going, oh, L);
            }
            if (occluded) diffuse = 0.0f;
        }
        if (going) {
            V3 color = v3(0.7f, 0.7f, 0.7f), emissive = v3(0.0f, 0.0f, 0.0f);
            float refl = 0.0f;
            if (mat != MISS) {
                const float4 m0 = sh.materials[4 * mat], m1 = sh.materials[4 * mat + 1], m2 = sh.materials[4 * mat + 2];
                color = v3(m0.x, m0.y, m0.z);
                emissive = v3(m1.x, m1.y, m1.z);
                refl = m2.x;  // metallic (:361)
                if (TEX) {
                    const float4 bc = base_color(s.tex, mat, uv);
                    color = v3(bc.x, bc.y, bc.z);
                    emissive = emissive_color(s.tex, mat, uv);
                }
            }
            V3 lit = color * (0.3f + diffuse * 0.7f) + emissive;
            if (show_grid > 0.5f && near_grid_boundary(s, pos, 0.02f)) lit = mix(lit, v3(0.0f, 1.0f, 0.0f), 0.6f);
            // bounce loop (:670-688)
            acc = acc + (lit * (1.0f - refl)) * mult;
            if (refl < 0.01f) {
                going = false;
            } else {
                mult *= refl;
                d = d - n * (2.0f * dot(n, d));  // reflect(e1, e2) = e1 - 2 * dot(e2, e1) * e2
                o = pos + n * 0.001f;
            }
        }
    }
    uint32_t rgba = 0;
    if (live) {
        if (sh.debug_enabled) {  // debug pixel (:693-716)
            const int dx = (int)px - (int)sh.debug_x, dy = (int)py - (int)sh.debug_y;
            if (dx == 0 && dy == 0)
                acc = v3(1.0f, 1.0f, 0.0f);
            else if (dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1)
                acc = mix(acc, v3(1.0f, 1.0f, 0.0f), 0.5f);
        }
        rgba = unorm8(acc.x) | (unorm8(acc.y) << 8) | (unorm8(acc.z) << 16) | 0xFF000000u;
    }
    if (in_tile) {
        const size_t oi = frame_first + (size_t)lt * m.tile_w * m.tile_h + (size_t)ly * m.tile_w + lx;
        if (sh.rgb8) {
            uint8_t* o3 = reinterpret_cast<uint8_t*>(out) + 3 * oi;
            o3[0] = (uint8_t)rgba, o3[1] = (uint8_t)(rgba >> 8), o3[2] = (uint8_t)(rgba >> 16);
        } else {
            out[oi] = rgba;
        }
    }
}

}  // namespace rt
