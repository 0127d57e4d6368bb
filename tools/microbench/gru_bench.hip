// gru_bench.hip -- developer harness: the recurrent GRU kernels of kns_gru.hip alone, at the bench shape (256 m-tiles x 64
// steps), timed with HIP events and compared bit for bit against the production kernel.  No Python, no torch: a run costs
// seconds.  Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DKNS_TIMING] -o build/gru_bench tools/microbench/gru_bench.hip
#include "../../koala_amd/csrc/kns_gru.hip"
namespace kns {
#include "gru_exp.hpp"
}

#include <stdio.h>
#include <string.h>

#include <vector>

using namespace kns;

static uint16_t h_bf16(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t) (u >> 16);
}
static uint32_t rng_state = 12345;
static float frand() {  // uniform [-1, 1)
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float) (int32_t) rng_state * (1.0f / 2147483648.0f);
}

// stands in for the input GEMM of the real sequence: rewrites the whole pre-activation buffer, last steps first, so that the
// recurrent kernel finds the memory-side cache in the state the engine leaves it in (first steps freshest)
__global__ __launch_bounds__(256) void refill_kernel(const uint4 *src, uint4 *dst, size_t n16) {
    const size_t nb = gridDim.x, b = nb - 1 - blockIdx.x;
    const size_t per = (n16 + nb - 1) / nb, lo = b * per, hi = lo + per < n16 ? lo + per : n16;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) dst[i] = src[i];
}

typedef void (*kern_t)(GruArgs);
struct Variant {
    const char *name;
    kern_t k;
    int threads;
};

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 64, mtiles = argc > 2 ? atoi(argv[2]) : 256, reps = argc > 3 ? atoi(argv[3]) : 30;
    const size_t n_gi = (size_t) T * mtiles * kGateTiles * 64 * 4;  // fp16 values
    const size_t n_w = (size_t) kGateTiles * 9 * 64 * 8, n_h = (size_t) mtiles * kUnitTiles * 256;
    const size_t n_hs = (size_t) T * mtiles * 9 * 64 * 8;  // bf16 values
    std::vector<_Float16> gi(n_gi);
    for (auto &v : gi) v = (_Float16) (frand() * 2.0f);
    std::vector<uint16_t> w(n_w);
    for (auto &v : w) v = h_bf16(frand() * 0.08f);
    std::vector<float> bias(kGateTiles * 16), h0(n_h);
    for (auto &v : bias) v = frand() * 0.1f;
    for (auto &v : h0) v = frand();
    void *d_gi, *d_w, *d_b, *d_h0, *d_h1, *d_hs;
    CHECK(hipMalloc(&d_gi, n_gi * 2));
    CHECK(hipMalloc(&d_w, n_w * 2));
    CHECK(hipMalloc(&d_b, bias.size() * 4));
    CHECK(hipMalloc(&d_h0, n_h * 4));
    CHECK(hipMalloc(&d_h1, n_h * 4));
    CHECK(hipMalloc(&d_hs, n_hs * 2));
    CHECK(hipMemcpy(d_gi, gi.data(), n_gi * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_w, w.data(), n_w * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_b, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_h0, h0.data(), n_h * 4, hipMemcpyHostToDevice));
    GruArgs g;
    g.gi = d_gi;
    g.whh = d_w;
    g.bhh = (const float *) d_b;
    g.hstate_in = (const float *) d_h0;
    g.hstate_out = (float *) d_h1;
    g.hseq = d_hs;
    g.T = T;
    g.mtiles = mtiles;
    g.precision = kBf16;

    std::vector<Variant> vs = {
        {"resident8 (production)", gru_resident8_kernel, 512},
        {"x0 (round-1 kernel)", gru_x_kernel<0>, 512},
        {"x8192 chain M0 + t16 merged, publish top", gru_x_kernel<8192>, 512},
        {"x12288 restructured", gru_x_kernel<8192 + 4096>, 512},
        {"x4096 round-1 + publish after G0", gru_x_kernel<4096>, 512},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<uint16_t> ref_hs(n_hs), hs(n_hs);
    std::vector<float> ref_h(n_h), hh(n_h);
    if (argc > 4) {  // power probe: loop ONE variant for ~argv[5] seconds (sample rocm-smi from outside)
        const int v = atoi(argv[4]);
        const double secs = argc > 5 ? atof(argv[5]) : 4.0;
        CHECK(hipEventRecord(e0, 0));
        double elapsed = 0;
        long launches = 0;
        while (elapsed < secs) {
            for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(vs[v].k, dim3(mtiles), dim3(vs[v].threads), 0, 0, g);
            launches += 500;
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            elapsed = ms * 1e-3;
        }
        printf("probe %s: %.1f us/launch over %.1f s\n", vs[v].name, elapsed * 1e6 / launches, elapsed);
        return 0;
    }
    // clock ramp
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(vs[0].k, dim3(mtiles), dim3(vs[0].threads), 0, 0, g);
    CHECK(hipDeviceSynchronize());
    const bool mix = getenv("GRU_BENCH_MIX") != nullptr;
    void *d_gi_src = nullptr;
    if (mix) {
        CHECK(hipMalloc(&d_gi_src, n_gi * 2));
        CHECK(hipMemcpy(d_gi_src, d_gi, n_gi * 2, hipMemcpyDeviceToDevice));
    }
    for (int pass = 0; pass < 2; ++pass)
        for (size_t v = 0; v < vs.size(); ++v) {
            if (mix) {  // every recurrent launch behind a refill of its input, timed alone
                CHECK(hipMemset(d_hs, 0, n_hs * 2));
                CHECK(hipMemset(d_h1, 0, n_h * 4));
                double total = 0;
                for (int i = 0; i < reps + 5; ++i) {
                    hipLaunchKernelGGL(refill_kernel, dim3(8192), dim3(256), 0, 0, (const uint4 *) d_gi_src, (uint4 *) d_gi, n_gi / 8);
                    CHECK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(vs[v].k, dim3(mtiles), dim3(vs[v].threads), 0, 0, g);
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                    float ms = 0;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (i >= 5) total += ms;
                }
                CHECK(hipMemcpy(hs.data(), d_hs, n_hs * 2, hipMemcpyDeviceToHost));
                if (v == 0) ref_hs = hs;
                size_t bad = 0;
                for (size_t i = 0; i < n_hs; ++i) bad += hs[i] != ref_hs[i];
                printf("pass %d  MIX %-34s %8.1f us/launch  mismatches hseq %zu\n", pass, vs[v].name, total * 1e3 / reps, bad);
                continue;
            }
            CHECK(hipMemset(d_hs, 0, n_hs * 2));
            CHECK(hipMemset(d_h1, 0, n_h * 4));
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(vs[v].k, dim3(mtiles), dim3(vs[v].threads), 0, 0, g);
            CHECK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(vs[v].k, dim3(mtiles), dim3(vs[v].threads), 0, 0, g);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            CHECK(hipGetLastError());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(hs.data(), d_hs, n_hs * 2, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(hh.data(), d_h1, n_h * 4, hipMemcpyDeviceToHost));
            if (v == 0) {
                ref_hs = hs;
                ref_h = hh;
            }
            size_t bad_hs = 0, bad_h = 0;
            for (size_t i = 0; i < n_hs; ++i) bad_hs += hs[i] != ref_hs[i];
            double maxd = 0;
            for (size_t i = 0; i < n_h; ++i) {
                bad_h += memcmp(&hh[i], &ref_h[i], 4) != 0;
                double d = fabs((double) hh[i] - ref_h[i]);
                if (d > maxd) maxd = d;
            }
            double cyc = 0;
#ifdef KNS_TIMING
            {
                unsigned long long t[8 * 16];
                read_timing(t);
                cyc = (double) (t[10] - t[9]) / 16.0;
                if (pass == 1) {
                    unsigned long long base = ~0ull;
                    for (int w8 = 0; w8 < 8; ++w8) base = t[w8 * 16] < base ? t[w8 * 16] : base;
                    for (int w8 = 0; w8 < 8; ++w8) {
                        printf("    wave %d:", w8);
                        for (int i = 0; i < 9; ++i) printf(" %6lld", (long long) (t[w8 * 16 + i] - base));
                        printf("\n");
                    }
                }
            }
#endif
            const double us = ms * 1e3 / reps;
            const double flop = 2.0 * 271 * 813 * 16.0 * mtiles * T;
            printf("pass %d  %-34s %8.1f us/launch  %6.3f us/step  %6.1f TFLOP/s (%.3f of 2500)  mismatches hseq %zu hstate %zu (max |d| %.3g)  %.0f cyc/step -> %.2f GHz\n",
                   pass, vs[v].name, us, us / T, flop / us * 1e-6, flop / us * 1e-6 / 2500.0, bad_hs, bad_h, maxd, cyc,
                   cyc / (us / T) * 1e-3);
        }
#ifdef KNS_TIMING
    {
        unsigned long long t[8 * 16];
        read_timing(t);
        unsigned long long base = ~0ull;
        for (int w8 = 0; w8 < 8; ++w8) base = t[w8 * 16] < base ? t[w8 * 16] : base;
        printf("stamps of the LAST launched variant, workgroup 0, step 5 (ticks from the earliest stamp 0)\n");
        for (int w8 = 0; w8 < 8; ++w8) {
            printf("wave %d:", w8);
            for (int i = 0; i < 9; ++i) printf(" %6lld", (long long) (t[w8 * 16 + i] - base));
            printf("\n");
        }
        printf("steady state: %.0f ticks per step\n", (double) (t[10] - t[9]) / 16.0);
    }
#endif
    return 0;
}
