// What does a wave pay for 64 scattered 48-byte records? Two ways of fetching them, timed on tables that do and do not
// fit the caches:
//   A  every lane fetches its own record: three 16-byte loads per lane, each instruction touching 64 different
//      cache lines, the same 64 lines three times over;
//   B  three neighbouring lanes fetch one record between them (lane l of instruction k takes piece (64k + l) % 3 of
//      record (64k + l) / 3): three instructions as well, but each touches only ~22 records' lines; the pieces reach
//      the lane that needs them through LDS (ds_write_b128 / three ds_read_b128).
//   C  B without the LDS round trip (every lane sums what it loaded): the memory side of B alone.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/gather_probe tools/gather_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

__device__ inline float sum4(float4 a) { return (a.x + a.y) + (a.z + a.w); }

template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ recs, const uint32_t* __restrict__ idx, uint32_t rounds,
                                                float* __restrict__ out) {
    __shared__ float4 stage[4][192];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const size_t wave_global = (size_t)blockIdx.x * 4u + wave;
    float acc = 0.f;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t mine = idx[(wave_global * rounds + r) * 64u + lane];
        if (MODE == 0) {
            const float4 a = recs[3 * (size_t)mine], b = recs[3 * (size_t)mine + 1], c = recs[3 * (size_t)mine + 2];
            acc += sum4(a) + sum4(b) + sum4(c);
        } else {
            float4 piece[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t q = 64u * k + lane, pair = q / 3u, part = q - 3u * pair;
                const uint32_t rec = (uint32_t)__shfl((int)mine, (int)pair, 64);
                piece[k] = recs[3 * (size_t)rec + part];
            }
            if (MODE == 2) {
                acc += sum4(piece[0]) + sum4(piece[1]) + sum4(piece[2]);
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) stage[wave][64 * k + lane] = piece[k];
                __builtin_amdgcn_wave_barrier();
                const float4 a = stage[wave][3 * lane], b = stage[wave][3 * lane + 1], c = stage[wave][3 * lane + 2];
                __builtin_amdgcn_wave_barrier();
                acc += sum4(a) + sum4(b) + sum4(c);
            }
        }
    }
    out[wave_global * 64u + lane] = acc;
}

int main(int argc, char** argv) {
    const uint32_t rounds = 64, blocks = 8192;  // 8192 workgroups x 4 waves x 64 rounds x 64 lanes = 134 M records
    const size_t n_idx = (size_t)blocks * 4 * rounds * 64;
    float* out;
    uint32_t* idx;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMalloc(&idx, n_idx * 4));
    std::vector<uint32_t> h(n_idx);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const uint32_t run = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u;  // consecutive lanes take `run` consecutive records
    printf("runs of %u consecutive records\n", run);
    for (size_t table_mb : {2, 48, 768, 3840}) {
        const size_t n_rec = table_mb * 1024 * 1024 / 48;
        float4* recs;
        CHECK(hipMalloc(&recs, n_rec * 48));
        CHECK(hipMemset(recs, 0, n_rec * 48));
        uint64_t s = 88172645463325252ull;
        for (size_t i = 0; i < n_idx; ++i) {
            if (i % run == 0) s ^= s << 13, s ^= s >> 7, s ^= s << 17;
            h[i] = (uint32_t)(s % (n_rec - run) + i % run);
        }
        CHECK(hipMemcpy(idx, h.data(), n_idx * 4, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) k_gather<0><<<blocks, 256>>>(recs, idx, rounds, out);
                if (mode == 1) k_gather<1><<<blocks, 256>>>(recs, idx, rounds, out);
                if (mode == 2) k_gather<2><<<blocks, 256>>>(recs, idx, rounds, out);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("table %5zu MB  mode %c  %8.3f ms  %7.2f G records/s  %7.1f GB/s of records\n", table_mb, "ABC"[mode], best,
                   n_idx / best * 1e-6, n_idx * 48.0 / best * 1e-6);
            fflush(stdout);
        }
        CHECK(hipFree(recs));
    }
    return 0;
}
