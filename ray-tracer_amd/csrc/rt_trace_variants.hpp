// DEVELOPMENT ONLY (built with -DRT_EXPERIMENTS into librt_mi355_exp.so, never shipped):
// alternative shapes of the traversal loop, timed against each other in one process by tools/variants_bench.py.
//
// VARIANT v = launch shape * 10 + body:
//   body 0      the shipped trace_ray
//   body 1..9   trace_ray_x<body> below (whatever is being tried); body 2 = trace_ray without LDS staging
//   0x          four waves per workgroup (shipped)      1x  one wave per workgroup
//   2x          __launch_bounds__(256, 8)               3x  per-wave start/end timestamps into g_wave_t
//   6x          at most 80 scalar registers (eight 256-thread workgroups per CU are admitted up to 80; the shipped kernel
//               holds 93, which admits seven)
//   4x          persistent waves: 2048 workgroups (every wave slot of the chip once), each wave takes the next 8x8
//               pixel block from an atomic counter until none is left
//   5x          the same, blocks handed out in the order of g_order (e.g. most expensive first)
#pragma once
#include "rt_trace.hpp"

namespace rt {

// experiment slot: by default the shipped function
template <int BODY, bool ANY>
__device__ inline bool trace_ray_x(const DevScene& s, V3 o, const RayDir& rd, HitRec& out) {
    Counters ctr{};
    return trace_ray<false, ANY, 0, false, false, false, BODY>(s, o, rd, out, ctr);
}


// Bodies 6 / 7 / 8: trace_ray_repack (rt_trace.hpp) with MODE 0 / 1 / 2.
template <int VARIANT>
struct VarCfg {
    static constexpr int body = VARIANT % 10;
    static constexpr bool stamps = VARIANT >= 30 && VARIANT < 40;
    static constexpr int threads = (VARIANT >= 10 && VARIANT < 20) ? 64 : 256;
    static constexpr int min_waves = (VARIANT >= 20 && VARIANT < 30) ? 8 : 1;
};

__device__ uint32_t g_queue[8 * 32];  // next block of the persistent variants: one counter (in a 128-byte line of its own) per stream that
                                     // launches them, zeroed before every launch
__device__ uint32_t g_order[65536];  // block handed out at each queue position (variants 5x)

template <int VARIANT>
__device__ inline void trace_block(const DevScene& s, const DevFrame& f, const ShardMap& m, uint2* __restrict__ out_p,
                                   uint2* __restrict__ out_s, uint32_t gw);

template <int VARIANT>
__global__ __launch_bounds__(VarCfg<VARIANT>::threads, VarCfg<VARIANT>::min_waves) void k_trace_primary_var(
    DevScene s, DevFrame f, ShardMap m, uint2* __restrict__ out_p, uint2* __restrict__ out_s) {
    const uint32_t gw = VarCfg<VARIANT>::threads == 64 ? blockIdx.x : blockIdx.x * 4 + (threadIdx.x >> 6);  // global wave
    trace_block<VARIANT>(s, f, m, out_p, out_s, gw);
}
#define RT_SGPR80(N)                                                                                                     \
    template <>                                                                                                          \
    __global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_trace_primary_var<N>(                  \
        DevScene s, DevFrame f, ShardMap m, uint2* __restrict__ out_p, uint2* __restrict__ out_s) {                      \
        trace_block<N>(s, f, m, out_p, out_s, blockIdx.x * 4 + (threadIdx.x >> 6));                                      \
    }

// Persistent waves: every wave reaches the exit test (chunk >= n_chunks) after finitely many blocks. A wave takes
// CHUNK queue positions per atomic (same-address device-scope atomics complete at about one per 15 ns on this chip:
// one atomic per 8x8 block made the launch four times slower than the plain grid). Chunk c holds positions
// c, c + n_chunks, c + 2 n_chunks, ...: with a cost-sorted order every wave starts with an expensive block and ends
// with cheap ones.
template <int VARIANT>
__global__ __launch_bounds__(256) void k_trace_primary_persist(DevScene s, DevFrame f, ShardMap m, uint2* __restrict__ out_p,
                                                               uint2* __restrict__ out_s, uint32_t n_waves, uint32_t* __restrict__ queue) {
    constexpr uint32_t CHUNK = (VARIANT % 10) ? (VARIANT % 10) : 8;
    const uint32_t n_chunks = (n_waves + CHUNK - 1) / CHUNK;
    for (;;) {
        uint32_t c = 0;
        if ((threadIdx.x & 63) == 0) c = atomicAdd(queue, 1u);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c >= n_chunks) break;
        for (uint32_t j = 0; j < CHUNK; ++j) {
            const uint32_t q = c + j * n_chunks;
            if (q >= n_waves) break;
            const uint32_t gw = VARIANT >= 50 ? g_order[q] : q;
            if (gw < n_waves) trace_block<(VARIANT / 10) * 10>(s, f, m, out_p, out_s, gw);
        }
    }
}

RT_SGPR80(60)

// Variant 80: every 8x8 pixel block by a workgroup of four waves, each wave a 4x4 quadrant with four lanes per ray
// (trace_ray<X = 5>). The whole frame this way measures what the mode costs where it is NOT needed; a small frame over
// the model's densest cells measures what it gains where it is.
__global__ __launch_bounds__(256) void k_trace_primary_quad_all(DevScene s, DevFrame f, ShardMap m, uint2* __restrict__ out_p,
                                                                uint2* __restrict__ out_s) {
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t gw = blockIdx.x;  // the 8x8 block, numbered as the waves of the shipped kernel
    const uint32_t w4 = gw & 3, b16 = gw >> 2;
    const uint32_t blk_x = (m.sub_x + 1) / 2, blk_y = (m.sub_y + 1) / 2;
    const uint32_t blk_per_tile = blk_x * blk_y;
    const uint32_t lt = b16 / blk_per_tile, blk = b16 % blk_per_tile;
    if (lt >= m.n_local) return;
    const uint32_t sub_x = (blk % blk_x) * 2 + (w4 & 1), sub_y = (blk / blk_x) * 2 + (w4 >> 1);
    const uint32_t r = lane >> 2;
    const uint32_t lx = sub_x * 8 + (w & 1) * 4 + (r & 3), ly = sub_y * 8 + (w >> 1) * 4 + (r >> 2);
    const uint32_t tile = shard_tile(m, lt);
    const uint32_t px = (tile % m.tiles_x) * m.tile_w + lx, py = (tile / m.tiles_x) * m.tile_h + ly;
    const bool in_tile = lx < m.tile_w && ly < m.tile_h;
    const bool live = in_tile && px < f.width && py < f.height;
    uint2 ph = make_uint2(MISS, 0u), sh = make_uint2(MISS, 0u);
    Counters ctr{};
    if (live) {
        const V3 o = v3(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]);
        const V3 d = primary_dir(f, px, py);
        HitRec h;
        const RayDir rd = make_ray_dir<false>(d, s.cell_size, s.px, s.pxy);
        if (trace_ray<false, false, 0, false, false, false, TRACE_QUAD>(s, o, rd, h, ctr, 4u)) {
            ph = make_uint2(h.id, __float_as_uint(h.t));
            const V3 pos = o + d * h.t;
            V3 n = hit_normal(s, h, pos);
            if (dot(n, d) > 0.0f) n = neg(n);
            const V3 so = pos + n * f.shadow_offset;
            HitRec oh;
            if (trace_ray<false, true, 0, false, false, false, TRACE_QUAD>(s, so, f.shadow, oh, ctr, 4u)) sh = make_uint2(oh.id, __float_as_uint(oh.t));
        }
    }
    if (in_tile && (lane & 3) == 0) {
        const size_t oi = (size_t)lt * m.tile_w * m.tile_h + (size_t)ly * m.tile_w + lx;
        out_p[oi] = ph;
        out_s[oi] = sh;
    }
}

template <int VARIANT>
__device__ inline void trace_block(const DevScene& s, const DevFrame& f, const ShardMap& m, uint2* __restrict__ out_p,
                                   uint2* __restrict__ out_s, uint32_t gw) {
    constexpr int BODY = VarCfg<VARIANT>::body;
    constexpr bool STAMPS = VarCfg<VARIANT>::stamps;
    const unsigned long long t_start = STAMPS ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t w4 = gw & 3, b16 = gw >> 2;
    const uint32_t blk_x = (m.sub_x + 1) / 2, blk_y = (m.sub_y + 1) / 2;
    const uint32_t blk_per_tile = blk_x * blk_y;
    const uint32_t lt = b16 / blk_per_tile, blk = b16 % blk_per_tile;
    if (lt >= m.n_local) return;
    const uint32_t sub_x = (blk % blk_x) * 2 + (w4 & 1), sub_y = (blk / blk_x) * 2 + (w4 >> 1);
    const uint32_t lx = sub_x * 8 + (lane & 7), ly = sub_y * 8 + (lane >> 3);
    const uint32_t tile = shard_tile(m, lt);
    const uint32_t px = (tile % m.tiles_x) * m.tile_w + lx, py = (tile / m.tiles_x) * m.tile_h + ly;
    const bool in_tile = lx < m.tile_w && ly < m.tile_h;
    const bool live = in_tile && px < f.width && py < f.height;
    uint2 ph = make_uint2(MISS, 0u), sh = make_uint2(MISS, 0u);
    Counters ctr{};
    auto trace = [&](V3 o, const RayDir& rd, HitRec& h, auto any) -> bool {
        constexpr bool ANY = decltype(any)::value;
        if constexpr (BODY == 0) return trace_ray<false, ANY, 0, false>(s, o, rd, h, ctr);
        else return trace_ray_x<BODY, ANY>(s, o, rd, h);
    };
    if constexpr (BODY >= 6 && BODY <= 8) {  // every lane of the wave takes part, with or without a ray of its own
        constexpr int MODE = BODY - 6;
        __shared__ RepackLds repack[4];
        RepackLds& L = repack[(threadIdx.x >> 6) & 3];
        const V3 o = v3(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]);
        const V3 d = primary_dir(f, live ? px : 0u, live ? py : 0u);
        HitRec h;
        const RayDir rd = make_ray_dir<false>(d, s.cell_size, s.px, s.pxy);
        const bool hit = trace_ray_repack<false, false, MODE>(s, o, rd, live, h, L);
        V3 so = o;
        if (hit) {
            ph = make_uint2(h.id, __float_as_uint(h.t));
            const V3 pos = o + d * h.t;
            V3 n = hit_normal(s, h, pos);
            if (dot(n, d) > 0.0f) n = neg(n);
            so = pos + n * f.shadow_offset;
        }
        HitRec oh;
        if (trace_ray_repack<false, true, MODE>(s, so, f.shadow, hit, oh, L)) sh = make_uint2(oh.id, __float_as_uint(oh.t));
    } else if (live) {
        const V3 o = v3(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]);
        const V3 d = primary_dir(f, px, py);
        HitRec h;
        const RayDir rd = make_ray_dir<false>(d, s.cell_size, s.px, s.pxy);
        if (trace(o, rd, h, std::false_type{})) {
            ph = make_uint2(h.id, __float_as_uint(h.t));
            const V3 pos = o + d * h.t;
            V3 n = hit_normal(s, h, pos);
            if (dot(n, d) > 0.0f) n = neg(n);
            const V3 so = pos + n * f.shadow_offset;
            HitRec oh;
            if (trace(so, f.shadow, oh, std::true_type{})) sh = make_uint2(oh.id, __float_as_uint(oh.t));
        }
    }
    if (in_tile) {
        const size_t oi = (size_t)lt * m.tile_w * m.tile_h + (size_t)ly * m.tile_w + lx;
        out_p[oi] = ph;
        out_s[oi] = sh;
    }
    if (STAMPS && lane == 0 && gw < 65536) {
        g_wave_t[2 * gw] = t_start;
        g_wave_t[2 * gw + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace rt
