// Micro-benchmark: issue rate of the bf16 MFMA shapes on gfx950, two waves per SIMD, 4 independent accumulators each.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512, 2) void k(int iters, float *out, long long *cyc) {
    const int lane = threadIdx.x & 63;
    bf16x8 a8, b8;
    s16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16) (float) (lane + i); b8[i] = (__bf16) (float) (lane * 2 + i); }
    for (int i = 0; i < 4; ++i) { a4[i] = (short) (0x3f80 + lane + i); b4[i] = (short) (0x3f80 + 2 * lane + i); }
    f32x4 c[4] = {};
    f32x16 d[2] = {};
    __builtin_amdgcn_s_barrier();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (KIND == 0) c[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c[j & 3], 0, 0, 0);
            if (KIND == 1) c[j & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c[j & 3], 0, 0, 0);
            if (KIND == 2) d[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, d[j & 1], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 3 && threadIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3] + d[0][0] + d[1][5];
}

int main() {
    float *out;
    long long *cyc, h;
    hipMalloc(&out, 64 * 512 * 4);
    hipMalloc(&cyc, 8);
    const int iters = 5000;
    const char *names[] = {"16x16x32_bf16", "16x16x16_bf16_1k", "32x32x16_bf16"};
    for (int kind = 0; kind < 3; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(8), dim3(512), 0, 0, iters, out, cyc);
            if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(8), dim3(512), 0, 0, iters, out, cyc);
            if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(8), dim3(512), 0, 0, iters, out, cyc);
            hipDeviceSynchronize();
        }
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);

        printf("%-18s %.1f cycles per MFMA (first wave of the SIMD, which is served first)\n", names[kind], (double) h / iters / 16);
    }
    return 0;
}
