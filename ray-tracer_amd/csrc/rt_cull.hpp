// Pixel blocks whose primary rays provably meet no occupied grid cell: a per-frame mask, one bit per 8x8 pixel block,
// made from the occupied cells of the grid before the trace kernel runs. A wave whose block's bit is clear writes its 64
// miss records and exits without traversing. No counterpart in the reference, whose every ray walks the grid
// (src/raytracer_unified.wgsl:520-588); the answers are the same bit for bit, because a ray can only report a hit from
// a candidate list, and only occupied cells have one.
//
// Why a clear bit is safe. The traversal being replaced is a float DDA: per axis the k-th boundary crossing happens at
// the float time T(k) = (((t0 + dt) + dt) + ...), which differs from the real crossing time of the same ray by at most
// (k + 8) * 2^-24 * T; at any moment the DDA's cell index on an axis is therefore the true ray's index at a time at most
// that far away, i.e. at a position at most eps = (k + 8) * 2^-24 * T away along that axis (|d| = 1). So every cell the
// DDA visits lies within eps (per axis) of the real ray o + t * d, t > 0, where d is the float direction the kernel
// computes for the pixel. The host only makes a mask when eps is below a twentieth of a cell for the whole grid
// (cull_frame), and the kernel below grows every occupied cell's box by a quarter of a cell plus 0.002 (the entry point's
// absolute offsets, :467-489) before projecting it. The projection itself is done in double precision on the exact
// pinhole model the pixel directions come from, and the pixel rectangle is padded by one pixel on every side (the float
// direction of a pixel differs from the model's by far less than a hundredth of a pixel). Hence: a pixel outside every
// rectangle has a ray that stays farther than eps from every occupied cell, so the DDA visits none.
// The one path that does not follow the ray - a start cell outside the grid, whose wrapped index names unrelated cells
// (rt_trace.hpp) - needs the origin on a far bound of the grid; the host makes no mask for such an origin.
#pragma once

#include "rt_trace.hpp"

namespace rt {

struct CullFrame {
    double o[3];
    double r0[3], r1[3], r2[3];  // rows of the inverse of [fwd | right * aspect * fov_scale | -up * fov_scale]:
                                 // (lambda, lambda * ndc_x, lambda * ndc_y) = rows * (Q - o) for a point Q on a pixel's ray
    double tol;                  // |p| or |q| below this at a crossing of the camera plane: sign unknown
    double near_r;               // camera this close (per axis) to a grown box: the box may be seen through any pixel
    double margin;               // growth of every occupied cell's box
    uint32_t width, height, blocks_x, blocks_y, words_per_row;
};

// One wave per workgroup, 512 grid cells per wave (a wave of this kernel has to find room on a GPU that the previous
// frame's trace on another stream keeps full: a single wave of at most 64 registers fits wherever a trace wave ends).
// The wave gathers the rectangles of its occupied cells in a copy of the mask in LDS and adds the words it touched to the
// frame's mask at the end, so it never waits for global memory between rectangles. Dynamic LDS: one word per mask word.
//
// The pixel a point Q is seen through: (lambda, p, q) = rows * (Q - o), ndc = (p, q) / lambda, lambda > 0 in front of the
// camera plane. lambda, p and q are evaluated in double precision; the quotient in float (a relative error of 1e-7 is a
// thousandth of a pixel where it matters, inside the frame).
__device__ inline void cull_project(const CullFrame& cf, double qx, double qy, double qz, double& lam, double& p, double& q) {
    const double vx = qx - cf.o[0], vy = qy - cf.o[1], vz = qz - cf.o[2];
    lam = cf.r0[0] * vx + cf.r0[1] * vy + cf.r0[2] * vz;
    p = cf.r1[0] * vx + cf.r1[1] * vy + cf.r1[2] * vz;
    q = cf.r2[0] * vx + cf.r2[1] * vy + cf.r2[2] * vz;
}

// Edges of a box that crosses the camera plane: the visible part of such an edge runs off to infinity in the direction
// of (p, q) at the crossing. Rare (cells level with the camera), so a plain loop.
__device__ inline void cull_crossing_edges(const CullFrame& cf, const double* lo, const double* hi, float& amin, float& amax,
                                                 float& bmn, float& bmx) {
    const float INF = __builtin_huge_valf();
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
#pragma unroll 1
        for (int bit = 1; bit < 8; bit <<= 1) {
            if (k & bit) continue;
            const int j = k | bit;
            double l0, p0, q0, l1, p1, q1;
            cull_project(cf, (k & 1) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 4) ? hi[2] : lo[2], l0, p0, q0);
            cull_project(cf, (j & 1) ? hi[0] : lo[0], (j & 2) ? hi[1] : lo[1], (j & 4) ? hi[2] : lo[2], l1, p1, q1);
            if ((l0 > 0.0) == (l1 > 0.0)) continue;
            const double s = l0 / (l0 - l1);
            const double pc = p0 + s * (p1 - p0), qc = q0 + s * (q1 - q0);
            if (!(pc < -cf.tol)) amax = INF;
            if (!(pc > cf.tol)) amin = -INF;
            if (!(qc < -cf.tol)) bmx = INF;
            if (!(qc > cf.tol)) bmn = -INF;
        }
    }
}

constexpr uint32_t CULL_CELLS_PER_WAVE = 512;

__device__ inline void block_mask_wave(const uint2* __restrict__ cells, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t px,
                                       uint32_t pxy, float bminx, float bminy, float bminz, float cs, const CullFrame& cf,
                                       uint32_t* __restrict__ mask) {
    extern __shared__ uint32_t lmask[];
    __shared__ uint32_t occupied[CULL_CELLS_PER_WAVE];
    // the frame's trace waits for this kernel, and it shares its SIMDs with the previous frame's trace waves: go first
    __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = threadIdx.x;
    const uint32_t words = cf.blocks_y * cf.words_per_row;
    for (uint32_t k = lane; k < words; k += 64u) lmask[k] = 0u;
    // the occupied ones among this wave's cells, packed (one cell in seven is occupied in the bundled models: projecting
    // them where they lie would leave most lanes idle)
    const uint32_t n = gx * gy * gz, base = blockIdx.x * CULL_CELLS_PER_WAVE;
    uint32_t n_occ = 0;
#pragma unroll
    for (uint32_t c = 0; c < CULL_CELLS_PER_WAVE / 64u; ++c) {
        const uint32_t i = base + c * 64u + lane;
        bool occ = false;
        if (i < n) {
            const uint32_t x = i % gx, y = (i / gx) % gy, z = i / (gx * gy);
            occ = cells[(x + 1u) + (y + 1u) * px + (z + 1u) * pxy].y != 0u;
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(occ);
        if (occ) occupied[n_occ + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = i;
        n_occ += (uint32_t)__popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    auto word_bits = [](uint32_t w, uint32_t x0, uint32_t x1) -> uint32_t {
        const uint32_t first = w == (x0 >> 5) ? (x0 & 31u) : 0u, last = w == (x1 >> 5) ? (x1 & 31u) : 31u;
        return (last == 31u ? 0xFFFFFFFFu : ((1u << (last + 1u)) - 1u)) & ~((1u << first) - 1u);
    };
    // the last large rectangle this wave wrote: cells next to the camera all cover the frame, once is enough
    uint32_t bx0 = 1, bx1 = 0, by0 = 1, by1 = 0;
    for (uint32_t batch = 0; batch < n_occ; batch += 64u) {
        uint32_t X0 = 0, X1 = 0, Y0 = 0, Y1 = 0;
        bool valid = false;
        if (batch + lane < n_occ) {
            const uint32_t i = occupied[batch + lane];
            const uint32_t x = i % gx, y = (i / gx) % gy, z = i / (gx * gy);
            const double c = (double)cs;
            const double lo[3] = {(double)bminx + x * c - cf.margin, (double)bminy + y * c - cf.margin, (double)bminz + z * c - cf.margin};
            const double hi[3] = {(double)bminx + (x + 1u) * c + cf.margin, (double)bminy + (y + 1u) * c + cf.margin,
                                  (double)bminz + (z + 1u) * c + cf.margin};
            double dist = 0.0;
            for (int k = 0; k < 3; ++k) dist = fmax(dist, fmax(lo[k] - cf.o[k], cf.o[k] - hi[k]));
            const float INF = __builtin_huge_valf();
            float amin = INF, amax = -INF, bmn = INF, bmx = -INF;
            bool any_front = false;
            if (!(dist > cf.near_r)) {  // the camera is in (or on) the grown box, or something is not a number
                amin = bmn = -INF;
                amax = bmx = INF;
                any_front = true;
            } else {
                // lambda is linear in the corner: its range over the box tells whether the box crosses the camera plane
                double lmin = 0.0, lmax = 0.0;
                for (int k = 0; k < 3; ++k) {
                    const double u = cf.r0[k] * (lo[k] - cf.o[k]), v = cf.r0[k] * (hi[k] - cf.o[k]);
                    lmin += fmin(u, v), lmax += fmax(u, v);
                }
                if (lmax > 0.0) {
#pragma unroll 2
                    for (int k = 0; k < 8; ++k) {
                        double lam, p, q;
                        cull_project(cf, (k & 1) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 4) ? hi[2] : lo[2], lam, p, q);
                        if (lam > 0.0) {  // in front of the camera plane: the pixel it is seen through
                            any_front = true;
                            const float inv = 1.0f / (float)lam;
                            float a = (float)p * inv, b = (float)q * inv;
                            // a corner a hair in front of the plane: lambda rounds to 0 or the quotient overflows
                            if (!(a == a)) a = p < 0.0 ? -INF : INF;
                            if (!(b == b)) b = q < 0.0 ? -INF : INF;
                            amin = fminf(amin, a), amax = fmaxf(amax, a);
                            bmn = fminf(bmn, b), bmx = fmaxf(bmx, b);
                        }
                    }
                    // lmin is a sum of rounded products: take a box as crossing unless it is clearly in front
                    if (!(lmin > 1e-9 * (lmax - lmin))) cull_crossing_edges(cf, lo, hi, amin, amax, bmn, bmx);
                }
            }
            if (any_front) {
                // ndc -> pixel centres: ndc_x = 2 (px + 0.5) / W - 1 (raytracer_unified.wgsl:634-640); one pixel of padding
                const float W = (float)cf.width, H = (float)cf.height;
                const float x0 = (amin + 1.0f) * 0.5f * W - 0.5f - 1.0f, x1 = (amax + 1.0f) * 0.5f * W - 0.5f + 1.0f;
                const float y0 = (bmn + 1.0f) * 0.5f * H - 0.5f - 1.0f, y1 = (bmx + 1.0f) * 0.5f * H - 0.5f + 1.0f;
                // written so that a NaN anywhere keeps the rectangle (and makes it the whole frame)
                if (!(x1 < 0.0f) && !(x0 > W - 1.0f) && !(y1 < 0.0f) && !(y0 > H - 1.0f)) {
                    const float cx0 = x0 > 0.0f ? x0 : 0.0f, cy0 = y0 > 0.0f ? y0 : 0.0f;
                    const float cx1 = x1 < W - 1.0f ? x1 : W - 1.0f, cy1 = y1 < H - 1.0f ? y1 : H - 1.0f;
                    X0 = (uint32_t)floorf(cx0) >> 3, X1 = (uint32_t)ceilf(cx1) >> 3;
                    Y0 = (uint32_t)floorf(cy0) >> 3, Y1 = (uint32_t)ceilf(cy1) >> 3;
                    if (X1 >= cf.blocks_x) X1 = cf.blocks_x - 1;
                    if (Y1 >= cf.blocks_y) Y1 = cf.blocks_y - 1;
                    valid = X0 <= X1 && Y0 <= Y1;
                }
            }
        }
        // fill. A rectangle of a few words (a cell far from the camera) is written by its own lane, all lanes at once;
        // a large one (a cell close to the camera can cover the frame) by the whole wave, a lane per row.
        const bool small = valid && ((X1 >> 5) - (X0 >> 5) + 1u) * (Y1 - Y0 + 1u) <= 32u;
        if (small) {
            for (uint32_t row = Y0; row <= Y1; ++row)
                for (uint32_t w = X0 >> 5; w <= (X1 >> 5); ++w) atomicOr(&lmask[row * cf.words_per_row + w], word_bits(w, X0, X1));
        }
        unsigned long long todo = __builtin_amdgcn_ballot_w64(valid && !small);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t x0 = (uint32_t)__builtin_amdgcn_readlane((int)X0, src), x1 = (uint32_t)__builtin_amdgcn_readlane((int)X1, src);
            const uint32_t y0 = (uint32_t)__builtin_amdgcn_readlane((int)Y0, src), y1 = (uint32_t)__builtin_amdgcn_readlane((int)Y1, src);
            if (x0 >= bx0 && x1 <= bx1 && y0 >= by0 && y1 <= by1) continue;  // inside the one written last
            for (uint32_t row = y0 + lane; row <= y1; row += 64u)
                for (uint32_t w = x0 >> 5; w <= (x1 >> 5); ++w) atomicOr(&lmask[row * cf.words_per_row + w], word_bits(w, x0, x1));
            bx0 = x0, bx1 = x1, by0 = y0, by1 = y1;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t k = lane; k < words; k += 64u) {
        const uint32_t v = lmask[k];
        if (v) atomicOr(mask + k, v);
    }
}

__global__ __launch_bounds__(64, 8) void k_block_mask(const uint2* __restrict__ cells, uint32_t gx, uint32_t gy, uint32_t gz,
                                                      uint32_t px, uint32_t pxy, float bminx, float bminy, float bminz, float cs,
                                                      CullFrame cf, uint32_t* __restrict__ mask) {
    block_mask_wave(cells, gx, gy, gz, px, pxy, bminx, bminy, bminz, cs, cf, mask);
}

// The masks of the frames of one rt_render_frames launch (blockIdx.y picks the frame), `words` mask words apart.
constexpr uint32_t CULL_MAX_FRAMES = 16;
struct CullFrames {
    CullFrame f[CULL_MAX_FRAMES];
};
__global__ __launch_bounds__(64, 8) void k_block_masks(const uint2* __restrict__ cells, uint32_t gx, uint32_t gy, uint32_t gz,
                                                       uint32_t px, uint32_t pxy, float bminx, float bminy, float bminz, float cs,
                                                       CullFrames cfs, uint32_t* __restrict__ masks, uint32_t words) {
    block_mask_wave(cells, gx, gy, gz, px, pxy, bminx, bminy, bminz, cs, cfs.f[blockIdx.y], masks + (size_t)blockIdx.y * words);
}

}  // namespace rt
