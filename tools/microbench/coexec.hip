// Micro-benchmark: when do MFMA and VALU work overlap on one gfx950 SIMD?  (inline asm bodies, s_memtime cycles)
// grid workgroups x 8 waves (two per SIMD; waves w and w+4 share a SIMD).  Work per wave and iteration by role:
//   M  16 v_mfma_f32_16x16x32_bf16, accumulators in VGPRs (4 chains)     A  same, accumulators in AGPRs
//   V  64 v_fma_f32 (8 chains)      T  48 v_fma_f32 + 16 v_exp_f32       P  64 v_pk_fma_f32 (8 chains)
//   X  16 x (mfma ; 4 v_fma) interleaved in ONE wave                      -  exit at once
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MFMA(c) "v_mfma_f32_16x16x32_bf16 %" #c ", %4, %5, %" #c "\n"
#define M4 MFMA(0) MFMA(1) MFMA(2) MFMA(3)
#define FMA4(a, b, c, d)                                                                \
    "v_fma_f32 %" #a ", %" #a ", %8, %9\n v_fma_f32 %" #b ", %" #b ", %8, %9\n"         \
    "v_fma_f32 %" #c ", %" #c ", %8, %9\n v_fma_f32 %" #d ", %" #d ", %8, %9\n"
#define EXP4(a, b, c, d)                                                                \
    "v_exp_f32 %" #a ", %" #a "\n v_exp_f32 %" #b ", %" #b "\n v_exp_f32 %" #c ", %" #c "\n v_exp_f32 %" #d ", %" #d "\n"
#define PK4(a, b, c, d)                                                                         \
    "v_pk_fma_f32 %" #a ", %" #a ", %8, %9\n v_pk_fma_f32 %" #b ", %" #b ", %8, %9\n"           \
    "v_pk_fma_f32 %" #c ", %" #c ", %8, %9\n v_pk_fma_f32 %" #d ", %" #d ", %8, %9\n"
#define XOUT "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
// X: operands 0-3 acc, 4/5 a/b, 6-13 x, 14/15 constants
#define XF4(a, b, c, d)                                                                 \
    "v_fma_f32 %" #a ", %" #a ", %14, %15\n v_fma_f32 %" #b ", %" #b ", %14, %15\n"     \
    "v_fma_f32 %" #c ", %" #c ", %14, %15\n v_fma_f32 %" #d ", %" #d ", %14, %15\n"

__global__ __launch_bounds__(512, 2) void k(const char *roles, int iters, float *out, long long *cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const char role = roles[wave];
    if (role == '-') return;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16) (float) (lane + i); b[i] = (__bf16) (float) (lane * 2 + i); }
    const float ka = 0.999f, kb = 0.001f;
    float res = 0;
    __builtin_amdgcn_s_barrier();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 'M') {
        f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int it = 0; it < iters; ++it) asm volatile(M4 M4 M4 M4 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        res = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (role == 'A') {
        f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int it = 0; it < iters; ++it) asm volatile(M4 M4 M4 M4 : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
        res = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (role == 'V' || role == 'T') {
        float x0 = lane * .001f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
        if (role == 'V')
            for (int it = 0; it < iters; ++it)
                asm volatile(FMA4(0, 1, 2, 3) FMA4(4, 5, 6, 7) FMA4(0, 1, 2, 3) FMA4(4, 5, 6, 7) FMA4(0, 1, 2, 3) FMA4(4, 5, 6, 7)
                                 FMA4(0, 1, 2, 3) FMA4(4, 5, 6, 7) FMA4(0, 1, 2, 3) FMA4(4, 5, 6, 7) FMA4(0, 1, 2, 3)
                                     FMA4(4, 5, 6, 7) FMA4(0, 1, 2, 3) FMA4(4, 5, 6, 7) FMA4(0, 1, 2, 3) FMA4(4, 5, 6, 7)
                             : XOUT
                             : "v"(ka), "v"(kb));
        else
            for (int it = 0; it < iters; ++it)
                asm volatile(FMA4(0, 1, 2, 3) FMA4(4, 5, 6, 7) EXP4(0, 1, 2, 3) FMA4(4, 5, 6, 7) FMA4(0, 1, 2, 3) EXP4(4, 5, 6, 7)
                                 FMA4(0, 1, 2, 3) FMA4(4, 5, 6, 7) EXP4(0, 1, 2, 3) FMA4(4, 5, 6, 7) FMA4(0, 1, 2, 3)
                                     EXP4(4, 5, 6, 7) FMA4(0, 1, 2, 3) FMA4(4, 5, 6, 7) FMA4(0, 1, 2, 3) FMA4(4, 5, 6, 7)
                             : XOUT
                             : "v"(ka), "v"(kb));
        res = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    } else if (role == 'P') {
        f32x2 x0 = {lane * .001f, 1.f}, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
        const f32x2 pa = {ka, ka}, pb = {kb, kb};
        for (int it = 0; it < iters; ++it)
            asm volatile(PK4(0, 1, 2, 3) PK4(4, 5, 6, 7) PK4(0, 1, 2, 3) PK4(4, 5, 6, 7) PK4(0, 1, 2, 3) PK4(4, 5, 6, 7)
                             PK4(0, 1, 2, 3) PK4(4, 5, 6, 7) PK4(0, 1, 2, 3) PK4(4, 5, 6, 7) PK4(0, 1, 2, 3) PK4(4, 5, 6, 7)
                                 PK4(0, 1, 2, 3) PK4(4, 5, 6, 7) PK4(0, 1, 2, 3) PK4(4, 5, 6, 7)
                         : XOUT
                         : "v"(pa), "v"(pb));
        res = x0[0] + x1[1] + x2[0] + x3[1] + x4[0] + x5[1] + x6[0] + x7[1];
    } else if (role == 'X') {
        f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        float x0 = lane * .001f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
        for (int it = 0; it < iters; ++it)
            asm volatile(MFMA(0) XF4(6, 7, 8, 9) MFMA(1) XF4(10, 11, 12, 13) MFMA(2) XF4(6, 7, 8, 9) MFMA(3) XF4(10, 11, 12, 13)
                             MFMA(0) XF4(6, 7, 8, 9) MFMA(1) XF4(10, 11, 12, 13) MFMA(2) XF4(6, 7, 8, 9) MFMA(3)
                                 XF4(10, 11, 12, 13) MFMA(0) XF4(6, 7, 8, 9) MFMA(1) XF4(10, 11, 12, 13) MFMA(2)
                                     XF4(6, 7, 8, 9) MFMA(3) XF4(10, 11, 12, 13) MFMA(0) XF4(6, 7, 8, 9) MFMA(1)
                                         XF4(10, 11, 12, 13) MFMA(2) XF4(6, 7, 8, 9) MFMA(3) XF4(10, 11, 12, 13)
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a), "+v"(b), XOUT
                         : "v"(ka), "v"(kb));
        res = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 5 && lane == 0) cyc[wave] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

int main() {
    float *out;
    long long *cyc, h[8];
    char *roles;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 64);
    hipMalloc(&roles, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    const char *cases[] = {"MMMM----", "AAAA----", "----VVVV", "----TTTT", "PPPP----", "MMMMVVVV", "MMMMTTTT", "XXXX----",
                           "XXXXXXXX", "MMMMMMMM", "AAAAAAAA", "VVVVVVVV", "TTTTTTTT", "PPPPPPPP", "MMMMPPPP", "AAAAPPPP"};
    for (int grid : {8, 256})
        for (const char *cs : cases) {
            hipMemcpy(roles, cs, 8, hipMemcpyHostToDevice);
            hipMemset(cyc, 0, 64);
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, roles, 100, out, cyc);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, roles, iters, out, cyc);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
            printf("grid %3d %s  %.3f ms  %.1f ns/iter   cycles/iter:", grid, cs, ms, ms * 1e6 / iters);
            for (int w = 0; w < 8; ++w) printf(" %.0f", (double) h[w] / iters);
            printf("\n");
        }
    return 0;
}
