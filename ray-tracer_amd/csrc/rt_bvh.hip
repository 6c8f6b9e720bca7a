// The reference's other acceleration structure, the one its CPU benchmark times: a binned-SAH bounding volume
// hierarchy over the triangles (BVHNode::build, src/core/bvh.rs:44-232) walked by the example's traversal
// (examples/massive_benchmark.rs:226-270,311-320): every node whose box the ray meets is opened, both children
// always, no ordering by distance, no pruning by the closest hit so far; leaves run Moller-Trumbore with the CPU
// routine's constants; the answer is the smallest t, or nothing.
//
// Build: on the host, the reference's algorithm as it stands (12 buckets, leaves of up to 4, partition in place),
// because its tree decides which boxes a degenerate ray meets. Layout on the device: nodes in pre-order with a
// "skip" index (the node that follows the whole subtree), so the both-children walk needs no stack - box missed or
// leaf done: go to skip, else go to the next node; leaf triangles as the trace's 48-byte records, in leaf order.
#include <hip/hip_runtime.h>

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

__global__ void k_bvh_records(const uint32_t* __restrict__ prims, uint32_t n, const float* __restrict__ tri_v,
                              float4* __restrict__ recs) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* v = tri_v + 9 * (size_t)prims[k];
    const V3 v0 = ld3(v), e1 = ld3(v + 3) - v0, e2 = ld3(v + 6) - v0;
    recs[3 * (size_t)k] = make_float4(v0.x, v0.y, v0.z, 0.f);
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
                const float t = moller_trumbore_pk<false>(rp, recs[r], recs[r + 1], recs[r + 2]);
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

}  // namespace

extern "C" int rt_build_bvh(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    c->err.clear();
    RT_HIP(c, hipSetDevice(c->device));
    c->has_bvh = false;
    if (!c->nt) return fail(c, RT_ERR_STATE, "rt_build_bvh: the scene has no triangles");
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
                       (uint32_t)np, c->tri_v.as<float>(), c->bvh_recs.as<float4>());
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipStreamSynchronize(c->stream));
    c->bvh_stats = b.stats;
    c->bvh_n_nodes = (uint32_t)nn;
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
    hipLaunchKernelGGL(k_trace_bvh, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->bvh_nodes.as<float4>(),
                       c->bvh_leaf_first.as<uint32_t>(), c->bvh_n_nodes, c->bvh_recs.as<float4>(), d_o, d_d, n, d_out);
    RT_HIP(c, hipGetLastError());
    if (!dev) {
        RT_HIP(c, hipMemcpyAsync(out_t, d_out, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        RT_HIP(c, hipStreamSynchronize(c->stream));
    }
    return RT_OK;
}
