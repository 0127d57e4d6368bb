// Micro-benchmark: dependent-accumulator latency of v_mfma_f32_16x16x32_bf16 on gfx950.  One wave per SIMD issues 24 MFMAs
// per iteration round-robin over C independent accumulators.  Only the C = 1 line is meaningful (back-to-back dependent
// MFMAs: 32 cycles): for C > 1 hipcc copies the accumulators in and out of the asm block every iteration and the loop
// drains the matrix pipe each time -- the issue rate with independent accumulators (16-17.5 cycles) is measured in coexec.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MF(c) "v_mfma_f32_16x16x32_bf16 %" #c ", %6, %7, %" #c "\n"
#define OUTS "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5)

__global__ __launch_bounds__(256, 1) void k(int chains, int iters, float *out, long long *cyc) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16) (float) (lane + i); b[i] = (__bf16) (float) (lane * 2 + i); }
    f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (chains == 1)
            asm volatile(MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0)
                             MF(0) MF(0) MF(0) MF(0) MF(0) MF(0)
                         : OUTS : "v"(a), "v"(b));
        else if (chains == 2)
            asm volatile(MF(0) MF(1) MF(0) MF(1) MF(0) MF(1) MF(0) MF(1) MF(0) MF(1) MF(0) MF(1) MF(0) MF(1) MF(0) MF(1) MF(0) MF(1)
                             MF(0) MF(1) MF(0) MF(1) MF(0) MF(1)
                         : OUTS : "v"(a), "v"(b));
        else if (chains == 3)
            asm volatile(MF(0) MF(1) MF(2) MF(0) MF(1) MF(2) MF(0) MF(1) MF(2) MF(0) MF(1) MF(2) MF(0) MF(1) MF(2) MF(0) MF(1) MF(2)
                             MF(0) MF(1) MF(2) MF(0) MF(1) MF(2)
                         : OUTS : "v"(a), "v"(b));
        else if (chains == 4)
            asm volatile(MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) MF(0) MF(1)
                             MF(2) MF(3) MF(0) MF(1) MF(2) MF(3)
                         : OUTS : "v"(a), "v"(b));
        else
            asm volatile(MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(0) MF(1) MF(2) MF(3) MF(4) MF(5)
                             MF(0) MF(1) MF(2) MF(3) MF(4) MF(5)
                         : OUTS : "v"(a), "v"(b));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 3 && threadIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1];
}

int main() {
    float *out;
    long long *cyc, h;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 8);
    const int iters = 10000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
    for (int chains : {1, 2, 3, 4, 6, 4, 1}) {
        hipLaunchKernelGGL(k, dim3(8), dim3(256), 0, 0, chains, 100, out, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(8), dim3(256), 0, 0, chains, iters, out, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("chains %d: %.1f ticks per MFMA, %.2f ns per MFMA (%.2f GHz)\n", chains, (double) h / iters / 24, ms * 1e6 / iters / 24,
               (double) h / (ms * 1e6));
    }
    return 0;
}
