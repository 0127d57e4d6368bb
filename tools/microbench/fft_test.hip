// fft_test.hip -- developer check of the four-frames-per-wave FFT-256 (kns_device.hpp) against a double-precision DFT,
// and of the row_negate / fft_partner lane maps.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o build/fft_test tools/microbench/fft_test.hip
#include "../../koala_amd/csrc/kns_device.hpp"

#include <math.h>
#include <stdio.h>

#include <vector>

using namespace kns;

__global__ void k_negate(float *out) {
    out[threadIdx.x] = lane_pair((float) fft_column(threadIdx.x));  // must be the column 16 - c (mod 16) except for columns 0 and 8
}

__global__ __launch_bounds__(64) void k_fft(const float2 *in, float2 *out, float2 *partner, const float2 *tw512) {
    __shared__ __attribute__((aligned(16))) char buf[kFftWaveBytes + kFftTwiddleBytes];
    const int lane = threadIdx.x, l = fft_column(lane), q = lane >> 4;
    fft_fill_twiddles(buf + kFftWaveBytes, tw512, lane, 64);
    __syncthreads();
    cpx v[16];
    for (int j = 0; j < 16; ++j) {
        float2 z = in[q * 256 + 16 * j + l];
        v[j] = cpx{z.x, z.y};
    }
    char *xw;
    const char *xr, *twl;
    fft_lane_bases(buf, buf + kFftWaveBytes, lane, &xw, &xr, &twl);
    fft256_rows(v, twl, xw, xr);
    cpx p[16];
    fft_partner(v, p, l);
    for (int k2 = 0; k2 < 16; ++k2) {
        out[q * 256 + l + 16 * k2] = float2{v[k2].x, v[k2].y};
        partner[q * 256 + l + 16 * k2] = float2{p[k2].x, p[k2].y};
    }
}

int main() {
    float *d;
    hipMalloc(&d, 64 * 4);
    hipLaunchKernelGGL(k_negate, dim3(1), dim3(64), 0, 0, d);
    float h[64];
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const int l = i & 15, c = (l & 1) ? (l == 1 ? 8 : 16 - (l >> 1)) : (l >> 1);
        if (c != 0 && c != 8) bad += (int) h[i] != 16 - c;
    }
    printf("column pairing: %s\n", bad ? "WRONG" : "ok");

    std::vector<float2> in(1024), out(1024), part(1024), tw(512);
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < 512; ++k) tw[k] = float2{(float) cos(2 * pi * k / 512), (float) -sin(2 * pi * k / 512)};
    unsigned s = 1;
    for (auto &z : in) {
        s = s * 1664525u + 1013904223u;
        z.x = (float) (int) s / 2147483648.0f;
        s = s * 1664525u + 1013904223u;
        z.y = (float) (int) s / 2147483648.0f;
    }
    float2 *di, *dout, *dp, *dtw;
    hipMalloc(&di, 8192);
    hipMalloc(&dout, 8192);
    hipMalloc(&dp, 8192);
    hipMalloc(&dtw, 4096);
    hipMemcpy(di, in.data(), 8192, hipMemcpyHostToDevice);
    hipMemcpy(dtw, tw.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_fft, dim3(1), dim3(64), 0, 0, di, dout, dp, dtw);
    hipMemcpy(out.data(), dout, 8192, hipMemcpyDeviceToHost);
    hipMemcpy(part.data(), dp, 8192, hipMemcpyDeviceToHost);
    double maxe = 0, maxp = 0;
    for (int f = 0; f < 4; ++f)
        for (int k = 0; k < 256; ++k) {
            double re = 0, im = 0;
            for (int n = 0; n < 256; ++n) {
                const double a = -2 * pi * k * n / 256;
                re += in[f * 256 + n].x * cos(a) - in[f * 256 + n].y * sin(a);
                im += in[f * 256 + n].x * sin(a) + in[f * 256 + n].y * cos(a);
            }
            maxe = fmax(maxe, fmax(fabs(re - out[f * 256 + k].x), fabs(im - out[f * 256 + k].y)));
            if (k) maxp = fmax(maxp, fmax(fabs(part[f * 256 + k].x - out[f * 256 + 256 - k].x), fabs(part[f * 256 + k].y - out[f * 256 + 256 - k].y)));
        }
    printf("fft256_rows: max abs error vs double DFT %.3g (inputs in [-1, 1): expect ~1e-5); partner map max mismatch %.3g\n", maxe, maxp);
    return bad || maxe > 1e-4 || maxp != 0;
}
