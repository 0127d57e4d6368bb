// What does v_mfma_f32_16x16x32_bf16 compute, bit for bit?  (round 5: the bf16 configuration's oracle restates the GEMMs as
// k-ascending fmaf chains -- true for v_mfma_f32_16x16x4_f32, never checked for the bf16 instruction.)
// Discovery tests on ONE output element (row 0, col 0; every other product 0), then a random validation of the model they imply:
//   1. grouping: C = 2^24 (ulp 2), products p_i = p_j = 1: both in one rounding group -> 2^24 + 2, else each add ties back to 2^24
//   2. order:    C = 2^24, p_i = -2^24, p_j = 1: j before i -> 0, i before j -> 1 (only meaningful across groups)
//   3. rounding: C = 2^24, p_i = 3 -> RNE 2^24 + 4, truncation 2^24 + 2;  p_i = 1 -> tie: RNE 2^24
//   4. width:    C = 2^24, p_i = 1, p_j = 2^-s (same group): exact 2^24 + 1 + 2^-s rounds UP to 2^24 + 2 if the small term survives
//   5. transcendentals: v_rcp_f32 separable in (mantissa, exponent)?  rcp + 2 fma == IEEE 1/d on [1, 2^64)?  v_exp_f32(y) ==
//      ldexp(v_exp_f32(frac), floor) ?
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// case layout: A[16][32] bf16 (row-major), B[32][16] bf16 (k-major), C[16][16] f32, D[16][16] f32
__global__ void mfma_case(const uint16_t *A, const uint16_t *B, const float *C, float *D) {
    const int l = threadIdx.x, cs = blockIdx.x;
    A += (size_t) cs * 512; B += (size_t) cs * 512; C += (size_t) cs * 256; D += (size_t) cs * 256;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        const int k = (l >> 4) * 8 + i;
        a[i] = __builtin_bit_cast(__bf16, A[(l & 15) * 32 + k]);
        b[i] = __builtin_bit_cast(__bf16, B[k * 16 + (l & 15)]);
    }
    f32x4 c;
    for (int i = 0; i < 4; ++i) c[i] = C[((l >> 4) * 4 + i) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[((l >> 4) * 4 + i) * 16 + (l & 15)] = c[i];
}

static uint16_t bf(float x) {  // exact for the values used in discovery; RNE otherwise
    uint32_t u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t) (u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t) h << 16; float f; memcpy(&f, &u, 4); return f; }

struct Cases {
    std::vector<uint16_t> A, B; std::vector<float> C, D; int n = 0;
    int add() { A.resize((n + 1) * 512, 0); B.resize((n + 1) * 512, 0); C.resize((n + 1) * 256, 0.f); return n++; }
    void set(int cs, int k, float a, float b) { A[cs * 512 + 0 * 32 + k] = bf(a); B[cs * 512 + k * 16 + 0] = bf(b); }
    void run() {
        uint16_t *dA, *dB; float *dC, *dD;
        CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, C.size() * 4)); CK(hipMalloc(&dD, C.size() * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(mfma_case, dim3(n), dim3(64), 0, 0, dA, dB, dC, dD);
        D.resize(C.size());
        CK(hipMemcpy(D.data(), dD, C.size() * 4, hipMemcpyDeviceToHost));
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dD));
    }
    float d00(int cs) const { return D[cs * 256]; }
};

// ---- candidate models of one output element: products p[32] (exact in fp32), accumulator c
static float model_chain(const float *a, const float *b, float c) { for (int k = 0; k < 32; ++k) c = fmaf(a[k], b[k], c); return c; }
// groups of `g` consecutive k, each group: one RNE rounding of the exact (c + sum of the group's products)
static float model_group(const float *a, const float *b, float c, int g) {
    for (int k0 = 0; k0 < 32; k0 += g) {
        long double s = c;
        for (int k = k0; k < k0 + g; ++k) s += (long double) a[k] * (long double) b[k];
        c = (float) s;  // (long double -> float: one rounding; exact sum as long as the spread stays under 64 bits)
    }
    return c;
}
// the products of a group summed exactly and rounded to fp32 FIRST, then added to c (two roundings per group)
static float model_group2(const float *a, const float *b, float c, int g) {
    for (int k0 = 0; k0 < 32; k0 += g) {
        long double s = 0;
        for (int k = k0; k < k0 + g; ++k) s += (long double) a[k] * (long double) b[k];
        c = c + (float) s;
    }
    return c;
}

__global__ void trans_probe(unsigned long long *out) {
    // out[0]: rcp not separable, out[1]: rcp + 2 fma != IEEE 1/d on [1, 2^64), out[2]: same on [1, 2^127), out[3]: exp2 not separable (|y| >= 1),
    // out[4]: exp2(y) != 1 for |y| < 2^-30 (sampled), out[5..]: spare
    const unsigned long long gid = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x, stride = (unsigned long long) gridDim.x * blockDim.x;
    unsigned long long n0 = 0, n1 = 0, n2 = 0, n3 = 0, n4 = 0, n5 = 0;
    for (unsigned long long m = gid; m < (1ull << 23); m += stride) {
        const float base = __builtin_bit_cast(float, 0x3f800000u | (unsigned) m);  // [1, 2)
        const float rb = __builtin_amdgcn_rcpf(base);
        for (int e = 0; e < 127; ++e) {
            const float d = __builtin_bit_cast(float, ((unsigned) (127 + e) << 23) | (unsigned) m);
            const float r = __builtin_amdgcn_rcpf(d);
            if (e < 126 && r != ldexpf(rb, -e)) ++n0;
            const float er = __builtin_fmaf(-d, r, 1.0f), r1 = __builtin_fmaf(r, er, r);
            const float q = 1.0f / d;
            if (r1 != q) { if (e < 64) ++n1; ++n2; }
        }
        // exp2: y = n + f with f = m 2^-23 in [0, 1): exact for |y| < 2^... (y in [1, 2): ulp 2^-23; larger |y| drop low bits of m)
        const float f = (float) m * 0x1p-23f;
        const float ef = __builtin_amdgcn_exp2f(f);
        for (int n = -20; n <= 20; ++n) {
            const float y = (float) n + f;
            if (y - (float) n != f) continue;  // f not representable next to n
            if (__builtin_amdgcn_exp2f(y) != ldexpf(ef, n)) ++n3;
        }
        const float tiny = __builtin_bit_cast(float, ((unsigned) (127 - 31 - (int) (m % 90)) << 23) | (unsigned) m);
        if (__builtin_amdgcn_exp2f(tiny) != 1.0f) ++n4;
        if (__builtin_amdgcn_exp2f(-tiny) != 1.0f) ++n5;
    }
    atomicAdd(&out[0], n0); atomicAdd(&out[1], n1); atomicAdd(&out[2], n2); atomicAdd(&out[3], n3); atomicAdd(&out[4], n4); atomicAdd(&out[5], n5);
}

int main() {
    const float big = 16777216.0f;  // 2^24
    Cases q;
    // 1. grouping matrix
    std::vector<int> grp_case(32 * 32, -1);
    for (int i = 0; i < 32; ++i)
        for (int j = i + 1; j < 32; ++j) {
            const int cs = q.add(); grp_case[i * 32 + j] = cs;
            q.C[cs * 256] = big; q.set(cs, i, 1.0f, 1.0f); q.set(cs, j, 1.0f, 1.0f);
        }
    // 2. order matrix
    std::vector<int> ord_case(32 * 32, -1);
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) if (i != j) {
            const int cs = q.add(); ord_case[i * 32 + j] = cs;
            q.C[cs * 256] = big; q.set(cs, i, -4096.0f, 4096.0f); q.set(cs, j, 1.0f, 1.0f);
        }
    // 3. rounding
    int r3[32], r1[32];
    for (int i = 0; i < 32; ++i) { int cs = q.add(); r3[i] = cs; q.C[cs * 256] = big; q.set(cs, i, 3.0f, 1.0f); cs = q.add(); r1[i] = cs; q.C[cs * 256] = big; q.set(cs, i, 1.0f, 1.0f); }
    // 4. width: partner in the same lane group (k, k+1) and across lane groups (k, k+8)
    int w_same[48], w_cross[48], w_c[48];
    for (int s = 1; s <= 48; ++s) {
        int cs = q.add(); w_same[s - 1] = cs; q.C[cs * 256] = big; q.set(cs, 0, 1.0f, 1.0f); q.set(cs, 1, ldexpf(1.0f, -(s / 2)), ldexpf(1.0f, -(s - s / 2)));
        cs = q.add(); w_cross[s - 1] = cs; q.C[cs * 256] = big; q.set(cs, 0, 1.0f, 1.0f); q.set(cs, 8, ldexpf(1.0f, -(s / 2)), ldexpf(1.0f, -(s - s / 2)));
        // the accumulator as the small term: C = 2^-s, products 2^24 (exact 4096 x 4096) and 1: exact 2^24 + 1 + 2^-s
        cs = q.add(); w_c[s - 1] = cs; q.C[cs * 256] = ldexpf(1.0f, -s); q.set(cs, 0, 4096.0f, 4096.0f); q.set(cs, 1, 1.0f, 1.0f);
    }
    q.run();
    printf("== 1. same rounding group (1 = the two unit products were added before any rounding)\n");
    for (int i = 0; i < 32; ++i) {
        printf("k=%2d: ", i);
        for (int j = 0; j < 32; ++j) {
            if (i == j) { printf("."); continue; }
            const int cs = grp_case[(i < j ? i : j) * 32 + (i < j ? j : i)];
            printf("%c", q.d00(cs) == big + 2 ? '1' : (q.d00(cs) == big ? '0' : '?'));
        }
        printf("\n");
    }
    printf("== 2. order (row i = -2^24 product, column j = unit product; value printed: result, 0 = j first (lost), 1 = i first)\n");
    for (int i = 0; i < 32; ++i) {
        printf("k=%2d: ", i);
        for (int j = 0; j < 32; ++j) { if (i == j) { printf("."); continue; } printf("%g", q.d00(ord_case[i * 32 + j])); }
        printf("\n");
    }
    printf("== 3. rounding: C = 2^24 + 3 ->"); for (int i = 0; i < 32; ++i) printf(" %g", q.d00(r3[i]) - big); printf("\n");
    printf("               C = 2^24 + 1 ->"); for (int i = 0; i < 32; ++i) printf(" %g", q.d00(r1[i]) - big); printf("\n");
    printf("== 4. width: 2^24 + 1 + 2^-s, result - 2^24 for s = 1..48\n   same lane group:");
    for (int s = 0; s < 48; ++s) printf(" %g", q.d00(w_same[s]) - big);
    printf("\n   across groups:  "); for (int s = 0; s < 48; ++s) printf(" %g", q.d00(w_cross[s]) - big);
    printf("\n   C the small one:"); for (int s = 0; s < 48; ++s) printf(" %g", q.d00(w_c[s]) - big);
    printf("\n");

    // ---- random validation of the candidate models
    srand(12345);
    Cases r;
    const int NR = 4096;
    for (int cs = 0; cs < NR; ++cs) {
        r.add();
        const int kind = cs % 4;
        for (int i = 0; i < 512; ++i) {
            float va, vb;
            auto rnd = [&]() { return (float) rand() / RAND_MAX * 2.0f - 1.0f; };
            if (kind == 0) { va = rnd(); vb = rnd(); }
            else if (kind == 1) { va = ldexpf(rnd(), rand() % 13 - 6); vb = ldexpf(rnd(), rand() % 13 - 6); }
            else if (kind == 2) { va = rnd() * 4; vb = rnd() * 0.1f; }
            else { va = ldexpf(rnd(), rand() % 25 - 12); vb = ldexpf(rnd(), rand() % 25 - 12); }
            r.A[cs * 512 + i] = bf(va); r.B[cs * 512 + i] = bf(vb);
        }
        for (int i = 0; i < 256; ++i) r.C[cs * 256 + i] = kind == 3 ? ldexpf((float) rand() / RAND_MAX - 0.5f, rand() % 25 - 12) : ((float) rand() / RAND_MAX - 0.5f) * 8;
    }
    r.run();
    const char *names[] = {"fmaf chain", "exact groups of 2", "of 4", "of 8", "of 16", "of 32", "rounded group sums of 4", "of 8", "of 16", "of 32"};
    long long miss[10][4] = {}, tot[4] = {};
    for (int cs = 0; cs < NR; ++cs)
        for (int row = 0; row < 16; ++row)
            for (int col = 0; col < 16; ++col) {
                float a[32], b[32];
                for (int k = 0; k < 32; ++k) { a[k] = bf2f(r.A[cs * 512 + row * 32 + k]); b[k] = bf2f(r.B[cs * 512 + k * 16 + col]); }
                const float c = r.C[cs * 256 + row * 16 + col], d = r.D[cs * 256 + row * 16 + col];
                float m[10] = {model_chain(a, b, c), model_group(a, b, c, 2), model_group(a, b, c, 4), model_group(a, b, c, 8), model_group(a, b, c, 16),
                               model_group(a, b, c, 32), model_group2(a, b, c, 4), model_group2(a, b, c, 8), model_group2(a, b, c, 16), model_group2(a, b, c, 32)};
                ++tot[cs % 4];
                for (int q2 = 0; q2 < 10; ++q2) if (memcmp(&m[q2], &d, 4)) ++miss[q2][cs % 4];
            }
    printf("== random validation (%d cases x 256 outputs; mismatches per input kind: uniform | wide exponents | mixed scale | very wide)\n", NR);
    for (int q2 = 0; q2 < 10; ++q2) printf("   %-26s %8lld %8lld %8lld %8lld  of %lld each\n", names[q2], miss[q2][0], miss[q2][1], miss[q2][2], miss[q2][3], tot[0]);
    // dump a few raw cases for offline modelling
    FILE *f = fopen("gpurun_out/mfma_probe_cases.bin", "wb");
    if (f) {
        const int ND = 256;
        fwrite(&ND, 4, 1, f);
        fwrite(r.A.data(), 2, (size_t) ND * 512, f); fwrite(r.B.data(), 2, (size_t) ND * 512, f);
        fwrite(r.C.data(), 4, (size_t) ND * 256, f); fwrite(r.D.data(), 4, (size_t) ND * 256, f);
        // the very-wide cases too
        for (int cs = 3; cs < 4 * ND; cs += 4) {
            fwrite(&r.A[cs * 512], 2, 512, f); fwrite(&r.B[cs * 512], 2, 512, f); fwrite(&r.C[cs * 256], 4, 256, f); fwrite(&r.D[cs * 256], 4, 256, f);
        }
        fclose(f);
    }

    unsigned long long *dout, hout[8] = {};
    CK(hipMalloc(&dout, 64)); CK(hipMemset(dout, 0, 64));
    hipLaunchKernelGGL(trans_probe, dim3(4096), dim3(256), 0, 0, dout);
    CK(hipMemcpy(hout, dout, 64, hipMemcpyDeviceToHost));
    printf("== 5. transcendentals: rcp not separable %llu | rcp+2fma != 1/d on [1,2^64) %llu, on [1,2^127) %llu (of %llu) | exp2 not separable %llu | exp2(tiny) != 1: %llu / %llu\n",
           hout[0], hout[1], hout[2], (1ull << 23) * 127, hout[3], hout[4], hout[5]);
    return 0;
}
