// The reference's other acceleration structure, the one its CPU benchmark times: a binned-SAH bounding volume
// hierarchy over the triangles (BVHNode::build, src/core/bvh.rs:44-232) walked by the example's traversal
// (examples/massive_benchmark.rs:226-270,311-320): every node whose box the ray meets is opened, both children
// always, no ordering by distance, no pruning by the closest hit so far; leaves run Moller-Trumbore with the CPU
// routine's constants; the answer is the smallest t, or nothing.
//
// Build: the reference's algorithm as it stands (12 buckets, leaves of up to 4, partition in place), because its
// tree decides which boxes a degenerate ray meets - on the GPU, level by level (rt_build_bvh, below), and on the
// host as the recursion the reference runs (rt_build_bvh_host: the checker of the GPU builder, node for node).
// Layout on the device: nodes in pre-order with a "skip" index (the node that follows the whole subtree), so the
// both-children walk needs no stack - box missed or leaf done: go to skip, else go to the next node; leaf triangles
// as the trace's 48-byte records, in leaf order.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <vector>

#include "rt_internal.hpp"
#include "rt_trace.hpp"

using namespace rt;

namespace {

constexpr size_t MAX_LEAF_SIZE = 4;  // src/core/bvh.rs:5
constexpr int SAH_BUCKETS = 12;      // src/core/bvh.rs:8

struct HBox {
    V3 mn, mx;
};
inline HBox hunion(const HBox& a, const HBox& b) { return HBox{vmin(a.mn, b.mn), vmax(a.mx, b.mx)}; }
inline float harea(const HBox& b) {  // AABB::surface_area, src/math/aabb.rs:24-28
    const V3 d = b.mx - b.mn;
    return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
}
inline float comp(V3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
inline size_t f2usize_sat(float x) {  // Rust `as usize`
    if (!(x > 0.0f)) return 0;
    if (x >= 18446744073709551616.0f) return SIZE_MAX;
    return (size_t)x;
}

struct Builder {
    const std::vector<HBox>& pb;  // primitive bounds
    std::vector<V3> pc;           // centroids
    std::vector<float4> nodes;    // two per node: (min, skip) (max, count)
    std::vector<uint32_t> leaf_first;
    std::vector<uint32_t> prims;  // triangle indices in leaf order
    rt_bvh_stats stats{};

    explicit Builder(const std::vector<HBox>& b) : pb(b) {
        pc.resize(pb.size());
        for (size_t i = 0; i < pb.size(); ++i) pc[i] = (pb[i].mn + pb[i].mx) * 0.5f;  // AABB::center
    }

    // src/core/bvh.rs:118-178
    void evaluate_axis(const uint32_t* idx, size_t n, const HBox& bounds, int axis, float* cost, float* pos) const {
        bool has[SAH_BUCKETS] = {};
        HBox bb[SAH_BUCKETS];
        size_t cnt[SAH_BUCKETS] = {};
        const float axis_extent = comp(bounds.mx - bounds.mn, axis);
        if (axis_extent < 1e-6f) {
            *cost = INFINITY;
            *pos = 0.0f;
            return;
        }
        for (size_t k = 0; k < n; ++k) {
            const uint32_t i = idx[k];
            const float offset = (comp(pc[i], axis) - comp(bounds.mn, axis)) / axis_extent;
            const size_t b = std::min<size_t>(f2usize_sat(offset * (float)SAH_BUCKETS), SAH_BUCKETS - 1);
            cnt[b]++;
            bb[b] = has[b] ? hunion(bb[b], pb[i]) : pb[i];
            has[b] = true;
        }
        float best_cost = INFINITY;
        int best_split = 0;
        for (int split = 1; split < SAH_BUCKETS; ++split) {
            bool la = false, ra = false;
            HBox lb{}, rb{};
            size_t lc = 0, rc = 0;
            for (int i = 0; i < split; ++i)
                if (has[i]) {
                    lb = la ? hunion(lb, bb[i]) : bb[i];
                    la = true;
                    lc += cnt[i];
                }
            for (int i = split; i < SAH_BUCKETS; ++i)
                if (has[i]) {
                    rb = ra ? hunion(rb, bb[i]) : bb[i];
                    ra = true;
                    rc += cnt[i];
                }
            if (la && ra) {
                const float c = 0.125f + 1.0f * (harea(lb) * (float)lc + harea(rb) * (float)rc);  // :203-209
                if (c < best_cost) {
                    best_cost = c;
                    best_split = split;
                }
            }
        }
        *cost = best_cost;
        *pos = comp(bounds.mn, axis) + ((float)best_split / (float)SAH_BUCKETS) * axis_extent;
    }

    // src/core/bvh.rs:49-92; returns the index of the node it made
    uint32_t build(uint32_t* idx, size_t n, size_t depth) {
        HBox bounds = pb[idx[0]];
        for (size_t k = 0; k < n; ++k) bounds = hunion(bounds, pb[idx[k]]);
        const uint32_t self = (uint32_t)leaf_first.size();
        nodes.push_back(make_float4(bounds.mn.x, bounds.mn.y, bounds.mn.z, 0.0f));
        nodes.push_back(make_float4(bounds.mx.x, bounds.mx.y, bounds.mx.z, 0.0f));
        leaf_first.push_back(0);
        auto leaf = [&]() {
            leaf_first[self] = (uint32_t)prims.size();
            prims.insert(prims.end(), idx, idx + n);
            nodes[2 * (size_t)self + 1].w = __builtin_bit_cast(float, (uint32_t)n);
            nodes[2 * (size_t)self].w = __builtin_bit_cast(float, self + 1u);
            stats.leaf_count++;
            stats.total_primitives += n;
            stats.max_depth = std::max<uint64_t>(stats.max_depth, depth);
            return self;
        };
        if (n <= MAX_LEAF_SIZE) return leaf();
        float best_cost = INFINITY, best_pos = 0.0f;
        int best_axis = 0;
        for (int axis = 0; axis < 3; ++axis) {
            float c, p;
            evaluate_axis(idx, n, bounds, axis, &c, &p);
            if (c < best_cost) {
                best_cost = c;
                best_axis = axis;
                best_pos = p;
            }
        }
        size_t left = 0, right = n;  // partition_primitives, :212-232
        while (left < right) {
            if (comp(pc[idx[left]], best_axis) < best_pos) {
                left++;
            } else {
                right--;
                std::swap(idx[left], idx[right]);
            }
        }
        if (left == 0 || left == n) return leaf();
        build(idx, left, depth + 1);
        build(idx + left, n - left, depth + 1);
        nodes[2 * (size_t)self].w = __builtin_bit_cast(float, (uint32_t)leaf_first.size());  // skip = first node after the subtree
        return self;
    }
};

// mark_copies: the spare word of record k says whether triangle k of the leaf order is, bit for bit, the triangle before
// it. A copy gives the same t as its original, and the walk keeps the smallest t only, so a leaf's loop passes over
// records marked so that follow another record of the SAME leaf (the loop never skips a leaf's first record). The
// reference's generator makes such scenes (src/core/benchmark.rs:102-105: 10 000 positions, repeated), and its builder
// cannot separate copies - they share a centroid -, so they end up side by side in one leaf.
__global__ void k_bvh_records(const uint32_t* __restrict__ prims, uint32_t n, const float* __restrict__ tri_v,
                              float4* __restrict__ recs, int mark_copies) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* v = tri_v + 9 * (size_t)prims[k];
    const V3 v0 = ld3(v), e1 = ld3(v + 3) - v0, e2 = ld3(v + 6) - v0;
    bool copy = mark_copies != 0 && k > 0;
    if (copy) {
        const uint32_t* a = reinterpret_cast<const uint32_t*>(v);
        const uint32_t* b = reinterpret_cast<const uint32_t*>(tri_v + 9 * (size_t)prims[k - 1]);
        for (int w = 0; w < 9; ++w) copy = copy && a[w] == b[w];
    }
    recs[3 * (size_t)k] = make_float4(v0.x, v0.y, v0.z, __uint_as_float(copy ? 1u : 0u));
    recs[3 * (size_t)k + 1] = make_float4(e1.x, e1.y, e1.z, 0.f);
    recs[3 * (size_t)k + 2] = make_float4(e2.x, e2.y, e2.z, 0.f);
}

// One lane per ray. examples/massive_benchmark.rs:311-320 for the box test (plain 1/dir, no epsilon: a zero
// component gives +-inf and 0*inf = NaN, which min/max then ignore - kept as is).
__global__ __launch_bounds__(256) void k_trace_bvh(const float4* __restrict__ nodes, const uint32_t* __restrict__ leaf_first,
                                                   uint32_t n_nodes, const float4* __restrict__ recs,
                                                   const float* __restrict__ origins, const float* __restrict__ dirs,
                                                   uint32_t n, float* __restrict__ out_t) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 o = ld3(origins + 3 * (size_t)i), d = ld3(dirs + 3 * (size_t)i);
    const V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const RayPk rp = make_ray_pk(o, d);
    float best = INFINITY;
    bool any = false;
    uint32_t node = 0;
    while (node < n_nodes) {
        const float4 a = nodes[2 * (size_t)node], b = nodes[2 * (size_t)node + 1];
        const V3 t1 = (v3(a.x, a.y, a.z) - o) * inv, t2 = (v3(b.x, b.y, b.z) - o) * inv;
        const V3 lo = vmin(t1, t2), hi = vmax(t1, t2);
        const float tmin = fmaxf(fmaxf(lo.x, lo.y), lo.z), tmax = fminf(fminf(hi.x, hi.y), hi.z);
        const uint32_t skip = __float_as_uint(a.w), count = __float_as_uint(b.w);
        if (!(tmax >= tmin && tmax >= 0.0f)) {
            node = skip;
            continue;
        }
        if (count) {
            const uint32_t first = leaf_first[node];
            for (uint32_t k = 0; k < count; ++k) {
                const size_t r = 3 * (size_t)(first + k);
                const float4 c0 = recs[r];
                if (k != 0 && __float_as_uint(c0.w) != 0u) continue;  // a copy of the record before it: the same t
                const float t = moller_trumbore_pk<false>(rp, c0, recs[r + 1], recs[r + 2]);
                if (t >= 0.0f && t < best) {  // the routine returns Some(t) only for t >= 1e-6
                    best = t;
                    any = true;
                }
            }
            node = skip;
        } else {
            node = node + 1;  // left child; the right one follows the left subtree
        }
    }
    out_t[i] = any ? best : -1.0f;
}

// The same walk with 32 lanes to a ray. The tree is cut into up to 32 subtrees (k_bvh_parts, once per build: the largest
// subtree whose root is an inner node is replaced by its two children until there are 32); a lane first meets the boxes
// of its subtree's ancestors - the example opens a node only if it met every box above it - and then walks its subtree
// alone. The ray's answer is the smallest t any lane found: the minimum does not care who found what, so the result is
// the one-lane walk's bit for bit, and a ray's dependent chain of node visits is a thirty-second as long.
constexpr uint32_t BVH_PARTS = 32, BVH_PART_DEPTH = 31;
struct BvhParts {
    uint32_t n;
    uint32_t first[BVH_PARTS], end[BVH_PARTS], n_above[BVH_PARTS];
    uint32_t above[BVH_PARTS][BVH_PART_DEPTH];  // the ancestors of a part's root, from the tree's root down
};

__global__ void k_bvh_parts(const float4* __restrict__ nodes, uint32_t n_nodes, BvhParts* __restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    BvhParts& p = *out;
    p.n = 1, p.first[0] = 0, p.end[0] = n_nodes, p.n_above[0] = 0;
    while (p.n < BVH_PARTS) {
        uint32_t pick = BVH_PARTS, size = 0;
        for (uint32_t k = 0; k < p.n; ++k) {
            const bool inner = __float_as_uint(nodes[2 * (size_t)p.first[k] + 1].w) == 0u;
            if (inner && p.n_above[k] < BVH_PART_DEPTH && p.end[k] - p.first[k] > size) pick = k, size = p.end[k] - p.first[k];
        }
        if (pick == BVH_PARTS) break;
        const uint32_t root = p.first[pick], left = root + 1, right = __float_as_uint(nodes[2 * (size_t)left].w), m = p.n++;
        for (uint32_t a = 0; a < p.n_above[pick]; ++a) p.above[m][a] = p.above[pick][a];
        p.above[pick][p.n_above[pick]] = p.above[m][p.n_above[pick]] = root;
        p.n_above[m] = ++p.n_above[pick];
        p.first[m] = right, p.end[m] = p.end[pick];
        p.first[pick] = left, p.end[pick] = right;
    }
}

__global__ __launch_bounds__(256) void k_trace_bvh_parts(const float4* __restrict__ nodes, const uint32_t* __restrict__ leaf_first,
                                                         const BvhParts* __restrict__ parts, const float4* __restrict__ recs,
                                                         const float* __restrict__ origins, const float* __restrict__ dirs,
                                                         uint32_t n, float* __restrict__ out_t) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = (uint32_t)(t / BVH_PARTS), part = (uint32_t)(t % BVH_PARTS);
    float best = INFINITY;
    if (i < n && part < parts->n) {
        const V3 o = ld3(origins + 3 * (size_t)i), d = ld3(dirs + 3 * (size_t)i);
        const V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        auto meets = [&](const float4& a, const float4& b) {
            const V3 t1 = (v3(a.x, a.y, a.z) - o) * inv, t2 = (v3(b.x, b.y, b.z) - o) * inv;
            const V3 lo = vmin(t1, t2), hi = vmax(t1, t2);
            const float tmin = fmaxf(fmaxf(lo.x, lo.y), lo.z), tmax = fminf(fminf(hi.x, hi.y), hi.z);
            return tmax >= tmin && tmax >= 0.0f;
        };
        bool open = true;
        for (uint32_t a = 0; a < parts->n_above[part] && open; ++a) {
            const uint32_t up = parts->above[part][a];
            open = meets(nodes[2 * (size_t)up], nodes[2 * (size_t)up + 1]);
        }
        if (open) {
            const RayPk rp = make_ray_pk(o, d);
            uint32_t node = parts->first[part];
            const uint32_t end = parts->end[part];
            while (node < end) {
                const float4 a = nodes[2 * (size_t)node], b = nodes[2 * (size_t)node + 1];
                const uint32_t skip = __float_as_uint(a.w), count = __float_as_uint(b.w);
                if (!meets(a, b)) {
                    node = skip;
                    continue;
                }
                if (count) {
                    const uint32_t first = leaf_first[node];
                    for (uint32_t k = 0; k < count; ++k) {
                        const size_t r = 3 * (size_t)(first + k);
                        const float4 c0 = recs[r];
                        if (k != 0 && __float_as_uint(c0.w) != 0u) continue;  // a copy of the record before it: the same t
                        const float tt = moller_trumbore_pk<false>(rp, c0, recs[r + 1], recs[r + 2]);
                        if (tt >= 0.0f && tt < best) best = tt;
                    }
                    node = skip;
                } else {
                    node = node + 1;
                }
            }
        }
    }
    for (int step = BVH_PARTS / 2; step > 0; step >>= 1) best = fminf(best, __shfl_xor(best, step, BVH_PARTS));
    if (i < n && part == 0) out_t[i] = best < INFINITY ? best : -1.0f;
}


// ---- the same tree, built on the GPU ----------------------------------------------------------------------------
// Level by level: every node of a level is a segment [lo, hi) of the index array. Per level, one pass each over the
// positions or the nodes: bounds (min / max are order-independent: atomics on order-preserving integer keys, a wave
// whose lanes share a node reduces first), leaf test, bucket counts and bounds per axis (src/core/bvh.rs:135-147; a
// workgroup whose 256 positions share a node bins in LDS first), the sweep (:149-178, one lane per node), and the
// reference's in-place partition (:212-232) as a permutation in closed form.
//
// The partition loop examines idx[left]; a small element (centroid < split) advances `left`, a big one is swapped
// with idx[--right], which brings an unexamined element from the back to `left`. With S = number of small elements,
// every position below S is examined from the front with its original content, every position above S is pulled
// from the back (position S itself: from the front if its element is big, else from the back), and:
//   a front small stays;
//   a front big with m front bigs before it goes to the slot just below the m-th small counted from the back (m = 0:
//     the last slot): that is where `right` stands when it is examined;
//   a back big at r goes to r - 1 (it is pulled to `left`, examined, and swapped to the new `right`);
//   the m-th small counted from the back goes to the position of the m-th front big (it ends the chain of pulls that
//     big started).
// Checked against the loop itself for every 0/1 array of up to 11 elements and 20 000 random ones
// (tests/test_bvh_partition_rule.py), and node for node against the host recursion on the GPU.
// the finished tree (bvh_nodes, bvh_n_nodes) cut into parts for k_trace_bvh_parts; waits for the stream
static int cut_into_parts(rt_ctx* c) {
    RT_HIP(c, c->bvh_parts.reserve(sizeof(BvhParts)));
    hipLaunchKernelGGL(k_bvh_parts, dim3(1), dim3(64), 0, c->stream, c->bvh_nodes.as<float4>(), c->bvh_n_nodes, c->bvh_parts.as<BvhParts>());
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t BIN_WORDS = 3 * SAH_BUCKETS * 7;  // per axis and bucket: count, min.xyz, max.xyz (keys)

__device__ inline uint32_t fkey(float f) {  // order-preserving: a < b  <=>  fkey(a) < fkey(b)
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ inline float funkey(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

struct GNodes {         // one entry per node, numbered in order of creation (a level's nodes are consecutive)
    uint32_t *lo, *hi;  // segment of the index array
    uint32_t* left;     // first child (the second is left + 1), NONE for a leaf
    uint32_t* depth;
    uint32_t* bkey;     // 6 keys per node: min.xyz, max.xyz
    uint32_t* axis;     // split axis; NONE: the node is a leaf by size and no split was looked for
    float* pos;         // split position
    uint32_t* size;     // nodes in the subtree
    uint32_t* pre;      // pre-order number
};

__global__ void kb_prims(const float* __restrict__ tri_v, uint32_t n, float4* __restrict__ pmin, float4* __restrict__ pmax,
                         float4* __restrict__ pc, uint32_t* __restrict__ idx, uint32_t* __restrict__ seg) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* v = tri_v + 9 * (size_t)k;
    const V3 p0 = ld3(v), p1 = ld3(v + 3), p2 = ld3(v + 6);
    const V3 mn = vmin(vmin(p0, p1), p2), mx = vmax(vmax(p0, p1), p2);  // TriangleData::bounds, src/types.rs:235-244
    const V3 c = (mn + mx) * 0.5f;                                      // AABB::center
    pmin[k] = make_float4(mn.x, mn.y, mn.z, 0.f);
    pmax[k] = make_float4(mx.x, mx.y, mx.z, 0.f);
    pc[k] = make_float4(c.x, c.y, c.z, 0.f);
    idx[k] = k;
    seg[k] = 0;
}

__global__ void kb_root(GNodes g, uint32_t n, uint32_t* __restrict__ counters) {  // counters: see rt_build_bvh
    g.lo[0] = 0, g.hi[0] = n, g.depth[0] = 0, g.left[0] = NONE, g.axis[0] = NONE;
    for (int j = 0; j < 3; ++j) g.bkey[j] = 0xFFFFFFFFu, g.bkey[3 + j] = 0u;
    counters[0] = 1;  // nodes made so far
    counters[4] = 1;  // levels so far
    counters[8] = 0, counters[9] = 1;  // lvl: the root's level
    counters[16] = 0;                  // level_first[0]
}

__device__ inline uint32_t wave_min_u32(uint32_t x) {
    for (int o = 32; o > 0; o >>= 1) x = min(x, (uint32_t)__shfl_xor((int)x, o, 64));
    return x;
}
__device__ inline uint32_t wave_max_u32(uint32_t x) {
    for (int o = 32; o > 0; o >>= 1) x = max(x, (uint32_t)__shfl_xor((int)x, o, 64));
    return x;
}

// bounds of every node of the level: union of its members' bounds (:55-58)
__global__ __launch_bounds__(256) void kb_bounds(GNodes g, const uint32_t* __restrict__ seg, const uint32_t* __restrict__ idx, uint32_t n,
                                                 const float4* __restrict__ pmin, const float4* __restrict__ pmax) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t node = k < n ? seg[k] : NONE;
    uint32_t key[6];
    if (node != NONE) {
        const float4 a = pmin[idx[k]], b = pmax[idx[k]];
        key[0] = fkey(a.x), key[1] = fkey(a.y), key[2] = fkey(a.z), key[3] = fkey(b.x), key[4] = fkey(b.y), key[5] = fkey(b.z);
    } else {
        key[0] = key[1] = key[2] = 0xFFFFFFFFu, key[3] = key[4] = key[5] = 0u;
    }
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)node);
    if (__builtin_amdgcn_ballot_w64(node != first) == 0ull) {  // the whole wave in one node (or in none)
        if (first == NONE) return;
        for (int j = 0; j < 3; ++j) key[j] = wave_min_u32(key[j]), key[3 + j] = wave_max_u32(key[3 + j]);
        if ((threadIdx.x & 63) == 0)
            for (int j = 0; j < 3; ++j) atomicMin(&g.bkey[6 * (size_t)first + j], key[j]), atomicMax(&g.bkey[6 * (size_t)first + 3 + j], key[3 + j]);
    } else if (node != NONE) {
        for (int j = 0; j < 3; ++j) atomicMin(&g.bkey[6 * (size_t)node + j], key[j]), atomicMax(&g.bkey[6 * (size_t)node + 3 + j], key[3 + j]);
    }
}

// leaf by size (:61), else the node's buckets are cleared; bins are addressed by lo / 5 (a node that is split holds
// at least five positions, so no two of a level share a slot)
__global__ void kb_node_init(GNodes g, const uint32_t* __restrict__ lvl, uint32_t* __restrict__ bins) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= lvl[1]) return;
    const uint32_t node = lvl[0] + j;
    const uint32_t n = g.hi[node] - g.lo[node];
    if (n <= MAX_LEAF_SIZE) {
        g.axis[node] = NONE;
        return;
    }
    g.axis[node] = 0;
    uint32_t* b = bins + (size_t)(g.lo[node] / 5u) * BIN_WORDS;
    for (uint32_t w = 0; w < 3 * SAH_BUCKETS; ++w) {
        b[7 * w] = 0u;
        b[7 * w + 1] = b[7 * w + 2] = b[7 * w + 3] = 0xFFFFFFFFu;
        b[7 * w + 4] = b[7 * w + 5] = b[7 * w + 6] = 0u;
    }
}

__device__ inline uint32_t bucket_of(float centroid, float bmin, float extent) {  // :138-139
    const float offset = (centroid - bmin) / extent;
    const float x = offset * (float)SAH_BUCKETS;
    uint32_t b = 0;  // Rust's `as usize`: NaN and negatives -> 0, large -> saturates
    if (x > 0.0f) b = x >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)x;
    return b < (uint32_t)SAH_BUCKETS - 1u ? b : (uint32_t)SAH_BUCKETS - 1u;
}

__global__ __launch_bounds__(256) void kb_bin(GNodes g, const uint32_t* __restrict__ seg, const uint32_t* __restrict__ idx, uint32_t n,
                                              const float4* __restrict__ pmin, const float4* __restrict__ pmax,
                                              const float4* __restrict__ pc, uint32_t* __restrict__ bins) {
    __shared__ uint32_t lbin[BIN_WORDS];
    __shared__ uint32_t block_node[2];
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t node = k < n ? seg[k] : NONE;
    if (node != NONE && g.axis[node] == NONE) node = NONE;  // a leaf by size
    // does the whole workgroup sit in one node? (its first and last position do, and segments are contiguous)
    if (threadIdx.x == 0) block_node[0] = node;
    if (threadIdx.x == blockDim.x - 1) block_node[1] = node;
    for (uint32_t w = threadIdx.x; w < BIN_WORDS; w += blockDim.x) {
        const uint32_t f = w % 7u;
        lbin[w] = f == 0u ? 0u : (f <= 3u ? 0xFFFFFFFFu : 0u);
    }
    __syncthreads();
    const bool shared_node = block_node[0] != NONE && block_node[0] == block_node[1];
    if (node != NONE) {
        const uint32_t t = idx[k];
        const float4 a = pmin[t], b = pmax[t], c = pc[t];
        const uint32_t ka[6] = {fkey(a.x), fkey(a.y), fkey(a.z), fkey(b.x), fkey(b.y), fkey(b.z)};
        const float cen[3] = {c.x, c.y, c.z};
        uint32_t* dst = shared_node ? lbin : bins + (size_t)(g.lo[node] / 5u) * BIN_WORDS;
        for (int axis = 0; axis < 3; ++axis) {
            const float bmin = funkey(g.bkey[6 * (size_t)node + axis]), bmax = funkey(g.bkey[6 * (size_t)node + 3 + axis]);
            const float extent = bmax - bmin;
            if (extent < 1e-6f) continue;  // :131-133: this axis is not looked at
            uint32_t* w = dst + 7u * ((uint32_t)axis * SAH_BUCKETS + bucket_of(cen[axis], bmin, extent));
            atomicAdd(w, 1u);
            for (int j = 0; j < 3; ++j) atomicMin(w + 1 + j, ka[j]), atomicMax(w + 4 + j, ka[3 + j]);
        }
    }
    __syncthreads();
    if (shared_node) {
        uint32_t* dst = bins + (size_t)(g.lo[block_node[0]] / 5u) * BIN_WORDS;
        for (uint32_t w = threadIdx.x; w < BIN_WORDS; w += blockDim.x) {
            const uint32_t f = w % 7u, v = lbin[w];
            if (f == 0u) {
                if (v) atomicAdd(dst + w, v);
            } else if (f <= 3u) {
                if (v != 0xFFFFFFFFu) atomicMin(dst + w, v);
            } else if (v != 0u) {
                atomicMax(dst + w, v);
            }
        }
    }
}

__device__ inline float key_area(const uint32_t* mn, const uint32_t* mx) {  // AABB::surface_area, src/math/aabb.rs:24-28
    const float dx = funkey(mx[0]) - funkey(mn[0]), dy = funkey(mx[1]) - funkey(mn[1]), dz = funkey(mx[2]) - funkey(mn[2]);
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}

// find_best_split (:95-178): one lane per node
__global__ void kb_split(GNodes g, const uint32_t* __restrict__ lvl, const uint32_t* __restrict__ bins) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= lvl[1]) return;
    const uint32_t node = lvl[0] + j;
    if (g.axis[node] == NONE) return;
    const uint32_t* b = bins + (size_t)(g.lo[node] / 5u) * BIN_WORDS;
    float best_cost = INFINITY, best_pos = 0.0f;
    uint32_t best_axis = 0;
    for (uint32_t axis = 0; axis < 3; ++axis) {
        const float bmin = funkey(g.bkey[6 * (size_t)node + axis]), bmax = funkey(g.bkey[6 * (size_t)node + 3 + axis]);
        const float extent = bmax - bmin;
        float cost = INFINITY, pos = 0.0f;
        if (!(extent < 1e-6f)) {
            const uint32_t* ab = b + 7u * axis * SAH_BUCKETS;
            // unions and counts of buckets [0, i] from the left and [i, 12) from the right (min / max of keys: any
            // order of taking the unions gives the reference's boxes)
            uint32_t lbox[SAH_BUCKETS][6], rbox[SAH_BUCKETS][6], lcnt[SAH_BUCKETS], rcnt[SAH_BUCKETS];
            for (int i = 0; i < SAH_BUCKETS; ++i) {
                const uint32_t* w = ab + 7 * i;
                for (int q = 0; q < 3; ++q) {
                    const uint32_t pmn = i ? lbox[i - 1][q] : 0xFFFFFFFFu, pmx = i ? lbox[i - 1][3 + q] : 0u;
                    lbox[i][q] = w[0] ? min(pmn, w[1 + q]) : pmn;
                    lbox[i][3 + q] = w[0] ? max(pmx, w[4 + q]) : pmx;
                }
                lcnt[i] = (i ? lcnt[i - 1] : 0u) + w[0];
            }
            for (int i = SAH_BUCKETS - 1; i >= 0; --i) {
                const uint32_t* w = ab + 7 * i;
                for (int q = 0; q < 3; ++q) {
                    const uint32_t pmn = i + 1 < SAH_BUCKETS ? rbox[i + 1][q] : 0xFFFFFFFFu, pmx = i + 1 < SAH_BUCKETS ? rbox[i + 1][3 + q] : 0u;
                    rbox[i][q] = w[0] ? min(pmn, w[1 + q]) : pmn;
                    rbox[i][3 + q] = w[0] ? max(pmx, w[4 + q]) : pmx;
                }
                rcnt[i] = (i + 1 < SAH_BUCKETS ? rcnt[i + 1] : 0u) + w[0];
            }
            float axis_best = INFINITY;
            int best_split = 0;
            for (int split = 1; split < SAH_BUCKETS; ++split) {
                const uint32_t lc = lcnt[split - 1], rc = rcnt[split];
                if (lc && rc) {
                    const float cst = 0.125f + 1.0f * (key_area(lbox[split - 1], lbox[split - 1] + 3) * (float)lc +
                                                       key_area(rbox[split], rbox[split] + 3) * (float)rc);  // :203-209
                    if (cst < axis_best) axis_best = cst, best_split = split;
                }
            }
            cost = axis_best;
            pos = bmin + ((float)best_split / (float)SAH_BUCKETS) * extent;
        }
        if (cost < best_cost) best_cost = cost, best_axis = axis, best_pos = pos;
    }
    g.axis[node] = best_axis;
    g.pos[node] = best_pos;
}

// big[k] = 1: the partition loop's `else` branch for this element (:223)
__global__ void kb_flags(GNodes g, const uint32_t* __restrict__ seg, const uint32_t* __restrict__ idx, uint32_t n,
                         const float4* __restrict__ pc, uint32_t* __restrict__ big) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n) return;
    uint32_t f = 0;
    if (k < n) {
        const uint32_t node = seg[k];
        if (node != NONE && g.axis[node] != NONE) {
            const float4 c = pc[idx[k]];
            const uint32_t axis = g.axis[node];
            const float x = axis == 0 ? c.x : (axis == 1 ? c.y : c.z);
            f = x < g.pos[node] ? 0u : 1u;
        }
    }
    big[k] = f;  // entry n closes the scan
}

struct Part {  // what a position needs of its node's partition
    uint32_t lo, hi, S, nfront;
};
__device__ inline Part part_of(const GNodes& g, uint32_t node, const uint32_t* __restrict__ pb, const uint32_t* __restrict__ big) {
    Part p;
    p.lo = g.lo[node], p.hi = g.hi[node];
    p.S = (p.hi - p.lo) - (pb[p.hi] - pb[p.lo]);
    p.nfront = p.S + ((p.lo + p.S < p.hi && big[p.lo + p.S]) ? 1u : 0u);
    return p;
}

// where the front bigs and the back smalls are, by their rank (see the rule above)
__global__ void kb_scatter(GNodes g, const uint32_t* __restrict__ seg, uint32_t n, const uint32_t* __restrict__ big,
                           const uint32_t* __restrict__ pb, uint32_t* __restrict__ front_big_pos, uint32_t* __restrict__ back_small_pos) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t node = seg[k];
    if (node == NONE || g.axis[node] == NONE) return;
    const Part p = part_of(g, node, pb, big);
    if (k - p.lo < p.nfront) {
        if (big[k]) front_big_pos[p.lo + (pb[k] - pb[p.lo])] = k;
    } else if (!big[k]) {
        const uint32_t m = (p.hi - 1u - k) - (pb[p.hi] - pb[k + 1u]) + 1u;  // 1-based, counted from the back
        back_small_pos[p.lo + m] = k;
    }
}

__global__ void kb_move(GNodes g, const uint32_t* __restrict__ seg, uint32_t n, const uint32_t* __restrict__ big,
                        const uint32_t* __restrict__ pb, const uint32_t* __restrict__ front_big_pos,
                        const uint32_t* __restrict__ back_small_pos, const uint32_t* __restrict__ idx_in, uint32_t* __restrict__ idx_out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t node = seg[k];
    uint32_t dest = k;
    if (node != NONE && g.axis[node] != NONE) {
        const Part p = part_of(g, node, pb, big);
        if (k - p.lo < p.nfront) {
            if (big[k]) {
                const uint32_t m = pb[k] - pb[p.lo];
                dest = m == 0u ? p.hi - 1u : back_small_pos[p.lo + m] - 1u;
            }
        } else if (big[k]) {
            dest = k - 1u;
        } else {
            const uint32_t m = (p.hi - 1u - k) - (pb[p.hi] - pb[k + 1u]) + 1u;
            dest = front_big_pos[p.lo + m - 1u];
        }
    }
    idx_out[dest] = idx_in[k];
}

// :74-86: a partition that leaves one side empty makes a leaf; else two children, appended to the node arrays
__global__ void kb_children(GNodes g, const uint32_t* __restrict__ lvl, const uint32_t* __restrict__ big, const uint32_t* __restrict__ pb,
                            uint32_t* __restrict__ counters, uint32_t max_nodes) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= lvl[1]) return;
    const uint32_t node = lvl[0] + j;
    g.left[node] = NONE;
    if (g.axis[node] == NONE) return;
    const Part p = part_of(g, node, pb, big);
    if (p.S == 0u || p.S == p.hi - p.lo) return;
    const uint32_t c = atomicAdd(&counters[0], 2u);
    if (c + 2u > max_nodes) {  // cannot happen (a tree over n primitives has at most 2 n - 1 nodes); never write past the arrays
        counters[1] = 1u;
        return;
    }
    g.left[node] = c;
    for (uint32_t q = 0; q < 2; ++q) {
        g.lo[c + q] = q ? p.lo + p.S : p.lo;
        g.hi[c + q] = q ? p.hi : p.lo + p.S;
        g.depth[c + q] = g.depth[node] + 1u;
        g.left[c + q] = NONE;
        g.axis[c + q] = NONE;
        for (int w = 0; w < 3; ++w) g.bkey[6 * (size_t)(c + q) + w] = 0xFFFFFFFFu, g.bkey[6 * (size_t)(c + q) + 3 + w] = 0u;
    }
}

__global__ void kb_seg(GNodes g, uint32_t* __restrict__ seg, uint32_t n) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t node = seg[k];
    if (node == NONE) return;
    const uint32_t left = g.left[node];
    seg[k] = left == NONE ? NONE : (k < g.hi[left] ? left : left + 1u);
}

// The level's bookkeeping stays on the device, so the host need not wait for every level: lvl = (first node, node count)
// of the level being built, level_first[l] = first node of level l, counters[4] = levels so far.
constexpr uint32_t MAX_LEVELS = 1u << 16;
__global__ void kb_advance(uint32_t* __restrict__ lvl, uint32_t* __restrict__ counters, uint32_t* __restrict__ level_first) {
    const uint32_t first = lvl[0] + lvl[1], count = counters[0] - first;
    lvl[0] = first, lvl[1] = count;
    if (count) {
        if (counters[4] < MAX_LEVELS) level_first[counters[4]] = first;
        counters[4]++;
    }
}

__global__ void kb_sizes(GNodes g, uint32_t first, uint32_t count) {  // levels from the deepest up
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const uint32_t node = first + j, left = g.left[node];
    g.size[node] = left == NONE ? 1u : 1u + g.size[left] + g.size[left + 1u];
}

__global__ void kb_preorder(GNodes g, uint32_t first, uint32_t count) {  // levels from the root down
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const uint32_t node = first + j, left = g.left[node];
    if (node == 0u) g.pre[0] = 0u;
    if (left != NONE) {
        g.pre[left] = g.pre[node] + 1u;
        g.pre[left + 1u] = g.pre[node] + 1u + g.size[left];
    }
}

// nodes in pre-order as the traversal reads them; counters: [2] leaves, [3] deepest leaf
__global__ void kb_emit(GNodes g, uint32_t n_nodes, float4* __restrict__ nodes, uint32_t* __restrict__ leaf_first,
                        uint32_t* __restrict__ counters) {
    const uint32_t node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= n_nodes) return;
    const uint32_t at = g.pre[node];
    const bool leaf = g.left[node] == NONE;
    const uint32_t* k = g.bkey + 6 * (size_t)node;
    nodes[2 * (size_t)at] = make_float4(funkey(k[0]), funkey(k[1]), funkey(k[2]), __uint_as_float(at + g.size[node]));
    nodes[2 * (size_t)at + 1] = make_float4(funkey(k[3]), funkey(k[4]), funkey(k[5]), __uint_as_float(leaf ? g.hi[node] - g.lo[node] : 0u));
    leaf_first[at] = leaf ? g.lo[node] : 0u;
    if (leaf) {
        atomicAdd(&counters[2], 1u);
        atomicMax(&counters[3], g.depth[node]);
    }
}

}  // namespace

extern "C" int rt_build_bvh(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    c->err.clear();
    RT_HIP(c, hipSetDevice(c->device));
    c->has_bvh = false;
    const uint32_t n = c->nt;
    if (!n) return fail(c, RT_ERR_STATE, "rt_build_bvh: the scene has no triangles");
    if (n > 0x3FFFFFFFu) return fail(c, RT_ERR_LIMIT, "rt_build_bvh: %u triangles (the node arrays hold 2 n entries)", n);
    hipStream_t st = c->stream;
    const uint32_t max_nodes = 2u * n;
    // working memory: one allocation, carved up; kept by the context for the next build (rt_destroy releases it)
    DevBuf& work = c->bvh_work;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    const size_t o_pmin = carve((size_t)n * 16), o_pmax = carve((size_t)n * 16), o_pc = carve((size_t)n * 16);
    const size_t o_idx0 = carve((size_t)n * 4), o_idx1 = carve((size_t)n * 4), o_seg = carve((size_t)n * 4);
    const size_t o_big = carve(((size_t)n + 1) * 4), o_pb = carve(((size_t)n + 1) * 4);
    const size_t o_fbp = carve(((size_t)n + 2) * 4), o_bsp = carve(((size_t)n + 2) * 4);
    const size_t o_bins = carve(((size_t)n / 5 + 1) * BIN_WORDS * 4);
    const size_t o_cnt = carve((16 + (size_t)MAX_LEVELS) * 4);  // [0] nodes, [1] overflow, [2] leaves, [3] depth, [4] levels, [8..9] lvl, [16..] level_first
    size_t o_node[9];
    const size_t node_words[9] = {1, 1, 1, 1, 6, 1, 1, 1, 1};
    for (int i = 0; i < 9; ++i) o_node[i] = carve((size_t)max_nodes * 4 * node_words[i]);
    size_t scan_bytes = 0;
    RT_HIP(c, rocprim::exclusive_scan(nullptr, scan_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), st));
    const size_t o_scan = carve(scan_bytes + 256);
    RT_HIP(c, work.reserve(off));
    uint8_t* base = work.as<uint8_t>();
    auto fail_free = [&](int rc) { return rc; };
    float4 *pmin = (float4*)(base + o_pmin), *pmax = (float4*)(base + o_pmax), *pc = (float4*)(base + o_pc);
    uint32_t *idx_a = (uint32_t*)(base + o_idx0), *idx_b = (uint32_t*)(base + o_idx1), *seg = (uint32_t*)(base + o_seg);
    uint32_t *big = (uint32_t*)(base + o_big), *pb = (uint32_t*)(base + o_pb), *fbp = (uint32_t*)(base + o_fbp), *bsp = (uint32_t*)(base + o_bsp);
    uint32_t *bins = (uint32_t*)(base + o_bins), *counters = (uint32_t*)(base + o_cnt);
    GNodes g;
    g.lo = (uint32_t*)(base + o_node[0]), g.hi = (uint32_t*)(base + o_node[1]), g.left = (uint32_t*)(base + o_node[2]);
    g.depth = (uint32_t*)(base + o_node[3]), g.bkey = (uint32_t*)(base + o_node[4]), g.axis = (uint32_t*)(base + o_node[5]);
    g.pos = (float*)(base + o_node[6]), g.size = (uint32_t*)(base + o_node[7]), g.pre = (uint32_t*)(base + o_node[8]);
    const dim3 per_prim((n + 255) / 256), per_prim1((n + 1 + 255) / 256), tb(256);
    if (hipMemsetAsync(counters, 0, 64, st) != hipSuccess) return fail_free(fail(c, RT_ERR_HIP, "rt_build_bvh: memset"));
    uint32_t *lvl = counters + 8, *level_first_dev = counters + 16;
    hipLaunchKernelGGL(kb_prims, per_prim, tb, 0, st, c->tri_v.as<float>(), n, pmin, pmax, pc, idx_a, seg);
    hipLaunchKernelGGL(kb_root, dim3(1), dim3(1), 0, st, g, n, counters);
    // Levels are queued without waiting for their sizes (per-node kernels are launched over the most nodes level l can
    // have, min(2^l, n), and read the real count on the device); every fourth level the host looks whether the last one
    // made any node. Levels queued past the last one find nothing to do.
    uint32_t made = 0;
    std::vector<uint32_t> level_first;
    for (uint32_t l = 0;; ++l) {
        const uint32_t most = l < 31 ? std::min<uint64_t>(1ull << l, n) : n;
        const dim3 per_node((most + 255) / 256);
        hipLaunchKernelGGL(kb_bounds, per_prim, tb, 0, st, g, seg, idx_a, n, pmin, pmax);
        hipLaunchKernelGGL(kb_node_init, per_node, tb, 0, st, g, lvl, bins);
        hipLaunchKernelGGL(kb_bin, per_prim, tb, 0, st, g, seg, idx_a, n, pmin, pmax, pc, bins);
        hipLaunchKernelGGL(kb_split, per_node, tb, 0, st, g, lvl, bins);
        hipLaunchKernelGGL(kb_flags, per_prim1, tb, 0, st, g, seg, idx_a, n, pc, big);
        size_t sb = scan_bytes;
        if (rocprim::exclusive_scan(base + o_scan, sb, big, pb, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), st) != hipSuccess)
            return fail_free(fail(c, RT_ERR_HIP, "rt_build_bvh: scan"));
        hipLaunchKernelGGL(kb_scatter, per_prim, tb, 0, st, g, seg, n, big, pb, fbp, bsp);
        hipLaunchKernelGGL(kb_move, per_prim, tb, 0, st, g, seg, n, big, pb, fbp, bsp, idx_a, idx_b);
        hipLaunchKernelGGL(kb_children, per_node, tb, 0, st, g, lvl, big, pb, counters, max_nodes);
        hipLaunchKernelGGL(kb_seg, per_prim, tb, 0, st, g, seg, n);
        hipLaunchKernelGGL(kb_advance, dim3(1), dim3(1), 0, st, lvl, counters, level_first_dev);
        std::swap(idx_a, idx_b);
        if ((l & 3u) != 3u) continue;
        uint32_t h[10];
        if (hipMemcpyAsync(h, counters, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return fail_free(fail(c, RT_ERR_HIP, "rt_build_bvh: %s", hipGetErrorString(hipGetLastError())));
        if (h[1] || h[0] > max_nodes) return fail_free(fail(c, RT_ERR_LIMIT, "rt_build_bvh: more nodes than 2 n"));
        if (h[4] > MAX_LEVELS) return fail_free(fail(c, RT_ERR_LIMIT, "rt_build_bvh: a tree deeper than %u levels", MAX_LEVELS));
        if (h[9] == 0u) {  // the level after the last queued one is empty: done
            made = h[0];
            level_first.resize(h[4]);
            if (hipMemcpyAsync(level_first.data(), level_first_dev, (size_t)h[4] * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess)
                return fail_free(fail(c, RT_ERR_HIP, "rt_build_bvh: copy"));
            break;
        }
    }
    const size_t levels = level_first.size();
    for (size_t l = levels; l-- > 0;) {
        const uint32_t f = level_first[l], cnt = (l + 1 < levels ? level_first[l + 1] : made) - f;
        hipLaunchKernelGGL(kb_sizes, dim3((cnt + 255) / 256), tb, 0, st, g, f, cnt);
    }
    for (size_t l = 0; l < levels; ++l) {
        const uint32_t f = level_first[l], cnt = (l + 1 < levels ? level_first[l + 1] : made) - f;
        hipLaunchKernelGGL(kb_preorder, dim3((cnt + 255) / 256), tb, 0, st, g, f, cnt);
    }
    if (c->bvh_nodes.reserve((size_t)made * 2 * sizeof(float4)) != hipSuccess || c->bvh_leaf_first.reserve((size_t)made * sizeof(uint32_t)) != hipSuccess ||
        c->bvh_prims.reserve((size_t)n * sizeof(uint32_t) + 16) != hipSuccess || c->bvh_recs.reserve((size_t)n * 3 * sizeof(float4) + 16) != hipSuccess)
        return fail_free(fail(c, RT_ERR_HIP, "rt_build_bvh: out of device memory"));
    hipLaunchKernelGGL(kb_emit, dim3((made + 255) / 256), tb, 0, st, g, made, c->bvh_nodes.as<float4>(), c->bvh_leaf_first.as<uint32_t>(), counters);
    uint32_t h[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(c->bvh_prims.p, idx_a, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail_free(fail(c, RT_ERR_HIP, "rt_build_bvh: copy"));
    hipLaunchKernelGGL(k_bvh_records, per_prim, tb, 0, st, c->bvh_prims.as<uint32_t>(), n, c->tri_v.as<float>(), c->bvh_recs.as<float4>(),
                       (int)c->params.skip_duplicate_triangles);
    if (hipMemcpyAsync(h, counters, 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess ||
        hipGetLastError() != hipSuccess)
        return fail_free(fail(c, RT_ERR_HIP, "rt_build_bvh: a kernel failed"));
    c->bvh_stats = rt_bvh_stats{made, h[2], h[3], n};
    c->bvh_n_nodes = made;
    if (int r = cut_into_parts(c)) return fail_free(r);
    c->has_bvh = true;
    return RT_OK;
}

// The tree as the traversal reads it, for callers that want to look at it (and for the test that compares the two
// builders): nodes = 8 floats per node in pre-order (min.xyz, skip index bits, max.xyz, leaf count bits), leaf_first
// and prims as described at the top. Host memory; sizes from rt_get_bvh_stats (node_count, total_primitives).
extern "C" int rt_read_bvh(rt_ctx* c, float* nodes, uint32_t* leaf_first, uint32_t* prims) {
    if (!c) return RT_ERR_INVALID;
    c->err.clear();
    RT_HIP(c, hipSetDevice(c->device));
    if (!c->has_bvh) return fail(c, RT_ERR_STATE, "rt_read_bvh: no BVH (call rt_build_bvh)");
    const size_t nn = c->bvh_n_nodes, np = (size_t)c->bvh_stats.total_primitives;
    if (nodes) RT_HIP(c, hipMemcpyAsync(nodes, c->bvh_nodes.p, nn * 2 * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    if (leaf_first) RT_HIP(c, hipMemcpyAsync(leaf_first, c->bvh_leaf_first.p, nn * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (prims) RT_HIP(c, hipMemcpyAsync(prims, c->bvh_prims.p, np * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

extern "C" int rt_build_bvh_host(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    c->err.clear();
    RT_HIP(c, hipSetDevice(c->device));
    c->has_bvh = false;
    if (!c->nt) return fail(c, RT_ERR_STATE, "rt_build_bvh_host: the scene has no triangles");
    std::vector<float> v((size_t)c->nt * 9);
    RT_HIP(c, hipMemcpyAsync(v.data(), c->tri_v.p, v.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    std::vector<HBox> pb(c->nt);
    for (uint32_t i = 0; i < c->nt; ++i) {  // TriangleData::bounds, src/types.rs:235-244
        const V3 p0 = ld3(&v[9 * (size_t)i]), p1 = ld3(&v[9 * (size_t)i + 3]), p2 = ld3(&v[9 * (size_t)i + 6]);
        pb[i] = HBox{vmin(vmin(p0, p1), p2), vmax(vmax(p0, p1), p2)};
    }
    Builder b(pb);
    std::vector<uint32_t> idx(c->nt);
    for (uint32_t i = 0; i < c->nt; ++i) idx[i] = i;
    b.build(idx.data(), idx.size(), 0);
    b.stats.node_count = b.leaf_first.size();
    const size_t nn = b.leaf_first.size(), np = b.prims.size();
    RT_HIP(c, c->bvh_nodes.reserve(nn * 2 * sizeof(float4)));
    RT_HIP(c, c->bvh_leaf_first.reserve(nn * sizeof(uint32_t)));
    RT_HIP(c, c->bvh_prims.reserve(np * sizeof(uint32_t) + 16));
    RT_HIP(c, c->bvh_recs.reserve(np * 3 * sizeof(float4) + 16));
    RT_HIP(c, hipMemcpyAsync(c->bvh_nodes.p, b.nodes.data(), nn * 2 * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipMemcpyAsync(c->bvh_leaf_first.p, b.leaf_first.data(), nn * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipMemcpyAsync(c->bvh_prims.p, b.prims.data(), np * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_bvh_records, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, c->stream, c->bvh_prims.as<uint32_t>(),
                       (uint32_t)np, c->tri_v.as<float>(), c->bvh_recs.as<float4>(), (int)c->params.skip_duplicate_triangles);
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipStreamSynchronize(c->stream));
    c->bvh_stats = b.stats;
    c->bvh_n_nodes = (uint32_t)nn;
    if (int r = cut_into_parts(c)) return r;
    c->has_bvh = true;
    return RT_OK;
}

extern "C" int rt_get_bvh_stats(rt_ctx* c, rt_bvh_stats* out) {
    if (!c) return RT_ERR_INVALID;
    c->err.clear();
    if (!c->has_bvh) return fail(c, RT_ERR_STATE, "rt_get_bvh_stats: no BVH (call rt_build_bvh)");
    if (!out) return fail(c, RT_ERR_INVALID, "rt_get_bvh_stats: NULL output");
    *out = c->bvh_stats;
    return RT_OK;
}

extern "C" int rt_trace_rays_bvh(rt_ctx* c, const float* origins, const float* dirs, uint32_t n, float* out_t) {
    if (!c) return RT_ERR_INVALID;
    c->err.clear();
    RT_HIP(c, hipSetDevice(c->device));
    if (!c->has_bvh) return fail(c, RT_ERR_STATE, "rt_trace_rays_bvh: no BVH (call rt_build_bvh after the scene or its vertices change)");
    if (!n) return RT_OK;
    if (!origins || !dirs || !out_t) return fail(c, RT_ERR_INVALID, "rt_trace_rays_bvh: NULL argument");
    const bool dev = is_device_ptr(out_t);
    if (dev != is_device_ptr(origins) || dev != is_device_ptr(dirs))
        return fail(c, RT_ERR_INVALID, "rt_trace_rays_bvh: arguments must be all host or all device memory");
    const float *d_o = origins, *d_d = dirs;
    float* d_out = out_t;
    const size_t rb = (size_t)n * 12;
    if (!dev) {
        RT_HIP(c, c->in_o.reserve(rb));
        RT_HIP(c, c->in_d.reserve(rb));
        RT_HIP(c, c->out_p.reserve((size_t)n * sizeof(float)));
        RT_HIP(c, hipMemcpyAsync(c->in_o.p, origins, rb, hipMemcpyHostToDevice, c->stream));
        RT_HIP(c, hipMemcpyAsync(c->in_d.p, dirs, rb, hipMemcpyHostToDevice, c->stream));
        d_o = c->in_o.as<float>(), d_d = c->in_d.as<float>(), d_out = c->out_p.as<float>();
    }
    // 32 lanes to a ray while that does not oversubscribe the chip (1 563 waves of 100 000 one-lane rays leave it idle behind
    // their dependent chains: 3.5 -> 1.5 ms; 2 M rays fill it one lane each, and the parts' repeated ancestor tests and
    // uneven sizes cost them 4.3 -> 16 ms). RT_BVH_LANES=1 / 32 (development, tests): that walk whatever the count.
    const char* lanes = getenv("RT_BVH_LANES");
    const bool one_lane = lanes ? atoi(lanes) == 1 : n > 262144u;
    if (one_lane)
        hipLaunchKernelGGL(k_trace_bvh, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->bvh_nodes.as<float4>(),
                           c->bvh_leaf_first.as<uint32_t>(), c->bvh_n_nodes, c->bvh_recs.as<float4>(), d_o, d_d, n, d_out);
    else
        hipLaunchKernelGGL(k_trace_bvh_parts, dim3((uint32_t)(((size_t)n * BVH_PARTS + 255) / 256)), dim3(256), 0, c->stream,
                           c->bvh_nodes.as<float4>(), c->bvh_leaf_first.as<uint32_t>(), c->bvh_parts.as<BvhParts>(),
                           c->bvh_recs.as<float4>(), d_o, d_d, n, d_out);
    RT_HIP(c, hipGetLastError());
    if (!dev) {
        RT_HIP(c, hipMemcpyAsync(out_t, d_out, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        RT_HIP(c, hipStreamSynchronize(c->stream));
    }
    return RT_OK;
}
