// Micro-benchmark: does the VGPR placement of the operands change the issue rate of v_mfma_f32_16x16x32_bf16 on gfx950?
// One wave per SIMD, 24 MFMAs per iteration over 4 independent accumulators, operand registers fixed by hand.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define S(x) #x
#define MF(c, a, b) "v_mfma_f32_16x16x32_bf16 v[" S(c) "], v[" S(a) "], v[" S(b) "], v[" S(c) "]\n"
#define BODY(A, B, C0, C1, C2, C3) \
    MF(C0, A, B) MF(C1, A, B) MF(C2, A, B) MF(C3, A, B) MF(C0, A, B) MF(C1, A, B) MF(C2, A, B) MF(C3, A, B) MF(C0, A, B) MF(C1, A, B) \
    MF(C2, A, B) MF(C3, A, B) MF(C0, A, B) MF(C1, A, B) MF(C2, A, B) MF(C3, A, B) MF(C0, A, B) MF(C1, A, B) MF(C2, A, B) MF(C3, A, B) \
    MF(C0, A, B) MF(C1, A, B) MF(C2, A, B) MF(C3, A, B)
#define CLOB "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v32", "v33", "v34", "v35", "v36", "v37", \
    "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51"

__global__ __launch_bounds__(256, 1) void k(int variant, int iters, long long *cyc) {
    asm volatile("v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n"
                 "v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n"
                 "v_mov_b32 v26, 1.0\n v_mov_b32 v27, 1.0\n" ::: CLOB);
    for (int r = 32; r < 52; ++r) asm volatile("" ::: CLOB);
    asm volatile("v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n"
                 "v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n"
                 "v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n"
                 "v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n" ::: CLOB);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (variant == 0) asm volatile(BODY(16:19, 20:23, 32:35, 36:39, 40:43, 44:47) ::: CLOB);       // A, B, C all start in bank 0
        else if (variant == 1) asm volatile(BODY(16:19, 22:25, 32:35, 36:39, 40:43, 44:47) ::: CLOB);  // B starts in bank 2
        else if (variant == 2) asm volatile(BODY(16:19, 20:23, 34:37, 38:41, 42:45, 46:49) ::: CLOB);  // C starts in bank 2
        else if (variant == 3) asm volatile(BODY(16:19, 22:25, 34:37, 38:41, 42:45, 46:49) ::: CLOB);  // B and C in bank 2
        else if (variant == 4) asm volatile(BODY(18:21, 22:25, 32:35, 36:39, 40:43, 44:47) ::: CLOB);  // A, B bank 2, C bank 0
        else asm volatile(BODY(16:19, 16:19, 32:35, 36:39, 40:43, 44:47) ::: CLOB);                    // A == B
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 3 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    long long *cyc, h;
    hipMalloc(&cyc, 8);
    const int iters = 10000;
    const char *names[] = {"A b0, B b0, C b0", "A b0, B b2, C b0", "A b0, B b0, C b2", "A b0, B b2, C b2", "A b2, B b2, C b0", "A == B, C b0"};
    for (int v = 0; v < 6; ++v) {
        hipLaunchKernelGGL(k, dim3(8), dim3(256), 0, 0, v, 100, cyc);
        hipLaunchKernelGGL(k, dim3(8), dim3(256), 0, 0, v, iters, cyc);
        hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-18s %.1f cycles per MFMA\n", names[v], (double) h / iters / 24);
    }
    return 0;
}
