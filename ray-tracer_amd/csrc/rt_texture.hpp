// Material evaluation at a triangle hit: the texture-dependent part of the shader's intersect_triangle
// (src/raytracer_unified.wgsl:279-364) - interpolated uv, base colour, the alpha MASK / BLEND rejection, normal
// map, emissive - against the texture array the renderer builds (src/renderer.rs:196-248: all textures as layers
// of one Rgba8Unorm array of the largest width and height, each written to the top-left corner of its layer) and
// its sampler (:251-259: repeat addressing, linear filter, mip level 0).
//
// Filtering is evaluated in f32: texel / 255, texel centres at integer + 0.5, weights = the fractional parts,
// horizontal blends first: (c00*(1-fx) + c10*fx)*(1-fy) + (c01*(1-fx) + c11*fx)*fy. GPU texture units use
// fixed-point weights, so a wgpu run of the shader can differ from this in the last bits of a colour.
#pragma once

#include "rt_math.hpp"

namespace rt {

struct DevTex {
    const uchar4* texels;      // [layer][y][x]
    uint32_t w, h, layers;     // layers == 0: the renderer's 1x1 white dummy
    const float* tri_uv;       // 6 floats per triangle: uv0, uv1, uv2
    const uint32_t* tri_mat;   // material index per triangle
    const float4* materials;   // MaterialData as four float4
    uint32_t n_materials;
};

struct F2 {
    float x, y;
};

__device__ inline uint32_t wrap_texel(float f, uint32_t n) {
    const int i = f2i_sat(f);
    if ((n & (n - 1u)) == 0u) return (uint32_t)i & (n - 1u);  // a power of two (the same in every lane): the same residue
    const int m = i % (int)n;
    return (uint32_t)(m < 0 ? m + (int)n : m);
}

// (float)a / 255.0f, correctly rounded, for a = 0 .. 255 in three instructions instead of a division: one Newton step
// on a * RN(1/255). Equal to the division for every one of the 256 inputs (tests/test_unorm8_exact.py checks it in
// exact arithmetic).
constexpr uint32_t RCP_255_BITS = 0x3B808081u;
__device__ inline float unorm8_to_float(unsigned char a) {
    const float fa = (float)a, r = __uint_as_float(RCP_255_BITS);
    const float q = fa * r;
    return fmaf(fmaf(-q, 255.0f, fa), r, q);
}

__device__ inline float4 sample_texture(const DevTex& tx, F2 uv, int index) {
    if (tx.layers == 0) return make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    const uint32_t layer = index < 0 ? 0u : ((uint32_t)index < tx.layers ? (uint32_t)index : tx.layers - 1u);
    const float x = uv.x * (float)tx.w - 0.5f, y = uv.y * (float)tx.h - 0.5f;
    const float x0 = floorf(x), y0 = floorf(y);
    const float fx = x - x0, fy = y - y0;
    const uint32_t ix0 = wrap_texel(x0, tx.w), ix1 = wrap_texel(x0 + 1.0f, tx.w);
    const uint32_t iy0 = wrap_texel(y0, tx.h), iy1 = wrap_texel(y0 + 1.0f, tx.h);
    const uchar4* base = tx.texels + (size_t)layer * tx.w * tx.h;
    const uchar4 c00 = base[(size_t)iy0 * tx.w + ix0], c10 = base[(size_t)iy0 * tx.w + ix1];
    const uchar4 c01 = base[(size_t)iy1 * tx.w + ix0], c11 = base[(size_t)iy1 * tx.w + ix1];
    auto blend = [&](unsigned char a, unsigned char b, unsigned char c, unsigned char d) {
        const float top = unorm8_to_float(a) * (1.0f - fx) + unorm8_to_float(b) * fx;
        const float bot = unorm8_to_float(c) * (1.0f - fx) + unorm8_to_float(d) * fx;
        return top * (1.0f - fy) + bot * fy;
    };
    return make_float4(blend(c00.x, c10.x, c01.x, c11.x), blend(c00.y, c10.y, c01.y, c11.y),
                       blend(c00.z, c10.z, c01.z, c11.z), blend(c00.w, c10.w, c01.w, c11.w));
}

// Barycentric u, v of the hit (the values the triangle test computes on its way to t, :252-263).
__device__ inline void hit_barycentrics(V3 o, V3 d, V3 v0, V3 e1, V3 e2, float& u, float& v) {
    const V3 h = cross(d, e2);
    const float a = dot(e1, h);
    const float f = 1.0f / a;
    const V3 s = o - v0;
    u = f * dot(s, h);
    const V3 q = cross(s, e1);
    v = f * dot(d, q);
}

// uv = uv0*w + uv1*u + uv2*v with w = 1 - u - v (:283-285)
__device__ inline F2 hit_uv(const DevTex& tx, uint32_t tri, float u, float v) {
    const float* p = tx.tri_uv + 6 * (size_t)tri;
    const float w = 1.0f - u - v;
    return F2{(p[0] * w + p[2] * u) + p[4] * v, (p[1] * w + p[3] * u) + p[5] * v};
}

// Base colour with alpha (:288-301) of material `mat` at `uv`.
__device__ inline float4 base_color(const DevTex& tx, uint32_t mat, F2 uv) {
    const float4 m0 = tx.materials[4 * mat], m1 = tx.materials[4 * mat + 1];
    const int texture = (int)__float_as_uint(m1.w);
    if (texture < 0) return m0;
    const float4 t = sample_texture(tx, uv, texture);
    return make_float4(t.x * m0.x, t.y * m0.y, t.z * m0.z, t.w * m0.w);
}

// The alpha rules (:303-316): true = this hit is discarded.
__device__ inline bool alpha_discards(const DevTex& tx, uint32_t mat, float alpha) {
    const float4 m3 = tx.materials[4 * mat + 3];
    const uint32_t mode = __float_as_uint(m3.x);
    if (mode == 1u) return alpha < m3.y;
    if (mode == 2u) return alpha < 0.01f;
    return false;
}

// Normal map (:319-348): the sampled tangent-space normal in a frame built from the geometric normal and a
// fixed up vector. gn = normalize(cross(edge1, edge2)), the value the hit already carries.
__device__ inline V3 mapped_normal(const DevTex& tx, uint32_t mat, F2 uv, V3 gn) {
    const float4 m2 = tx.materials[4 * mat + 2];
    const int texture = (int)__float_as_uint(m2.z);
    if (texture < 0) return gn;
    const float4 s = sample_texture(tx, uv, texture);
    const V3 tn = normalize(v3(s.x * 2.0f - 1.0f, s.y * 2.0f - 1.0f, s.z * 2.0f - 1.0f));
    const V3 up = fabsf(gn.y) > 0.9f ? v3(0.0f, 0.0f, 1.0f) : v3(0.0f, 1.0f, 0.0f);
    const V3 tangent = normalize(cross(up, gn));
    const V3 bitangent = cross(gn, tangent);
    return normalize((tangent * tn.x + bitangent * tn.y) + gn * tn.z);
}

// Emissive factor times the emissive texture (:350-362).
__device__ inline V3 emissive_color(const DevTex& tx, uint32_t mat, F2 uv) {
    const float4 m1 = tx.materials[4 * mat + 1], m2 = tx.materials[4 * mat + 2];
    const int texture = (int)__float_as_uint(m2.w);
    V3 e = v3(m1.x, m1.y, m1.z);
    if (texture < 0) return e;
    const float4 s = sample_texture(tx, uv, texture);
    return v3(e.x * s.x, e.y * s.y, e.z * s.z);
}

}  // namespace rt
