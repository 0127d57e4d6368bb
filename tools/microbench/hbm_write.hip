// Micro-benchmark: what does MI355X sustain for a pure streaming WRITE of the input GEMM's `gi` size (214 MB), for a
// read of the same size, and for the 3:1 write:read mix of that kernel?  (bounds the weight-stationary GEMM from below)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <class V>
__global__ __launch_bounds__(512) void wr(V *dst, size_t n, unsigned seed) {
    V v;
    for (int i = 0; i < (int) (sizeof(V) / 4); ++i) v[i] = seed + threadIdx.x + i;
    for (size_t i = (size_t) blockIdx.x * 512 + threadIdx.x; i < n; i += (size_t) gridDim.x * 512) dst[i] = v;
}
template <class V>
__global__ __launch_bounds__(512) void rd(const V *src, size_t n, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t) blockIdx.x * 512 + threadIdx.x; i < n; i += (size_t) gridDim.x * 512) acc += src[i][0];
    if (acc == 0x12345678) sink[0] = acc;
}
template <class V>
__global__ __launch_bounds__(512) void mix(V *dst, const V *src, size_t n, unsigned *sink) {  // 3 writes : 1 read
    unsigned acc = 0;
    for (size_t i = (size_t) blockIdx.x * 512 + threadIdx.x; i < n; i += (size_t) gridDim.x * 512) {
        V v = src[i / 3];
        acc += v[0];
        dst[i] = v;
    }
    if (acc == 0x12345678) sink[0] = acc;
}

int main() {
    const size_t bytes = (size_t) 214 << 20;
    void *a, *b;
    unsigned *sink;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMalloc(&sink, 4);
    hipMemset(a, 1, bytes);
    hipMemset(b, 2, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch, double gb) {
        for (int i = 0; i < 20; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int n = 200;
        for (int i = 0; i < n; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %.1f us  %.2f TB/s\n", name, ms * 1e3 / n, gb / (ms / n));
    };
    for (int grid : {256, 512, 1024, 2048}) {
        printf("grid %d\n", grid);
        time("write 8 B/lane", [&] { hipLaunchKernelGGL(wr<u32x2>, dim3(grid), dim3(512), 0, 0, (u32x2 *) a, bytes / 8, 1u); }, bytes / 1e9);
        time("write 16 B/lane", [&] { hipLaunchKernelGGL(wr<u32x4>, dim3(grid), dim3(512), 0, 0, (u32x4 *) a, bytes / 16, 1u); }, bytes / 1e9);
        time("read 16 B/lane", [&] { hipLaunchKernelGGL(rd<u32x4>, dim3(grid), dim3(512), 0, 0, (const u32x4 *) b, bytes / 16, sink); }, bytes / 1e9);
        time("write 214 MB + read 71 MB", [&] { hipLaunchKernelGGL(mix<u32x4>, dim3(grid), dim3(512), 0, 0, (u32x4 *) a, (const u32x4 *) b, bytes / 16, sink); },
             bytes * (4.0 / 3.0) / 1e9);
    }
    return 0;
}
