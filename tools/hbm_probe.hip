// Development tool: what a plain device copy reaches on this GPU, over the shapes a copy kernel can take
// (grid size, loads in flight per lane, cache hints, read-only and write-only forms). Build:
//   hipcc --offload-arch=gfx950 -O3 -o tools/hbm_probe tools/hbm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef uint32_t vec4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const vec4* __restrict__ src, vec4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        vec4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = NT ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (NT) __builtin_nontemporal_store(v[k], dst + i + k * stride);
            else dst[i + k * stride] = v[k];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// each workgroup owns one contiguous chunk
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_chunk(const vec4* __restrict__ src, vec4* __restrict__ dst, size_t n) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t b = per * blockIdx.x, e = b + per < n ? b + per : n;
    size_t i = b + threadIdx.x;
    for (; i + (U - 1) * 256 < e; i += U * 256) {
        vec4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = NT ? __builtin_nontemporal_load(src + i + k * 256) : src[i + k * 256];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (NT) __builtin_nontemporal_store(v[k], dst + i + k * 256);
            else dst[i + k * 256] = v[k];
        }
    }
    for (; i < e; i += 256) dst[i] = src[i];
}

template <int U>
__global__ __launch_bounds__(256) void k_read(const vec4* __restrict__ src, uint32_t* sink, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    vec4 acc = {0, 0, 0, 0};
    for (; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
        for (int k = 0; k < U; ++k) acc ^= __builtin_nontemporal_load(src + i + k * stride);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345677u) sink[0] = 1;
}

__global__ __launch_bounds__(256) void k_write(vec4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    const vec4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) __builtin_nontemporal_store(v, dst + i);
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 2048) << 20;
    const size_t n = bytes / 16;
    vec4 *a, *b;
    uint32_t* sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double moved, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipGetLastError());
        printf("%-44s %8.1f GB/s\n", name, moved * 5 / (ms * 1e-3) / 1e9);
        fflush(stdout);
    };
    const int grids[] = {1024, 2048, 4096, 8192, 16384, 65536};
    char name[128];
    for (int g : grids) {
        snprintf(name, sizeof name, "copy strided U=8 nt grid=%d", g);
        timeit(name, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<8, true>), dim3(g), dim3(256), 0, 0, a, b, n); });
        snprintf(name, sizeof name, "copy strided U=8 plain grid=%d", g);
        timeit(name, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<8, false>), dim3(g), dim3(256), 0, 0, a, b, n); });
        snprintf(name, sizeof name, "copy strided U=4 nt grid=%d", g);
        timeit(name, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<4, true>), dim3(g), dim3(256), 0, 0, a, b, n); });
        snprintf(name, sizeof name, "copy strided U=2 nt grid=%d", g);
        timeit(name, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<2, true>), dim3(g), dim3(256), 0, 0, a, b, n); });
        snprintf(name, sizeof name, "copy chunked U=8 nt grid=%d", g);
        timeit(name, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy_chunk<8, true>), dim3(g), dim3(256), 0, 0, a, b, n); });
        snprintf(name, sizeof name, "copy chunked U=4 plain grid=%d", g);
        timeit(name, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy_chunk<4, false>), dim3(g), dim3(256), 0, 0, a, b, n); });
    }
    {
        const size_t full = (n + 255) / 256;
        timeit("copy one element per lane (grid = n/256) nt", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<1, true>), dim3((uint32_t)full), dim3(256), 0, 0, a, b, n); });
        timeit("copy one element per lane (grid = n/256)", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<1, false>), dim3((uint32_t)full), dim3(256), 0, 0, a, b, n); });
    }
    timeit("hipMemcpyAsync device to device", 2.0 * bytes, [&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
    for (int g : {2048, 8192}) {
        snprintf(name, sizeof name, "read only U=8 grid=%d", g);
        timeit(name, 1.0 * bytes, [&] { hipLaunchKernelGGL((k_read<8>), dim3(g), dim3(256), 0, 0, a, sink, n); });
        snprintf(name, sizeof name, "write only grid=%d", g);
        timeit(name, 1.0 * bytes, [&] { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, b, n); });
    }
    timeit("hipMemsetAsync", 1.0 * bytes, [&] { CK(hipMemsetAsync(b, 0, bytes, 0)); });
    return 0;
}
