// What does the host link carry for the shapes the host-pointer path moves?  (round 5)
// Page-locked host buffers [4096 rows][64 frames x 512 B]; copies of a column block of `frames` frames: contiguous (a staging slot) and
// strided 2-D (the caller's matrix), host-to-device, device-to-host, and both directions at once on two streams.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t rows = 4096, T = 64, pitch = T * 512;
    char *h_in, *h_out, *d_in, *d_out;
    CK(hipHostMalloc((void **) &h_in, rows * pitch, hipHostMallocDefault));
    CK(hipHostMalloc((void **) &h_out, rows * pitch, hipHostMallocDefault));
    CK(hipMalloc((void **) &d_in, rows * pitch));
    CK(hipMalloc((void **) &d_out, rows * pitch));
    for (size_t i = 0; i < rows * pitch; i += 4096) h_in[i] = 1, h_out[i] = 1;
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    printf("frames  MB   | contiguous H2D  D2H  both(H2D,D2H) | 2-D strided H2D  D2H  both   [GB/s]   | one 2-D H2D call: us\n");
    for (int frames : {1, 2, 4, 8, 16, 32, 64}) {
        const size_t width = (size_t) frames * 512, bytes = rows * width;
        const int reps = frames <= 4 ? 40 : 12;
        double r[6], lat = 0;
        for (int mode = 0; mode < 6; ++mode) {
            const bool two_d = mode >= 3;
            const int dir = mode % 3;  // 0 H2D, 1 D2H, 2 both
            auto h2d = [&]() { if (two_d) CK(hipMemcpy2DAsync(d_in, width, h_in, pitch, width, rows, hipMemcpyHostToDevice, s1)); else CK(hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, s1)); };
            auto d2h = [&]() { if (two_d) CK(hipMemcpy2DAsync(h_out, pitch, d_out, width, width, rows, hipMemcpyDeviceToHost, s2)); else CK(hipMemcpyAsync(h_out, d_out, bytes, hipMemcpyDeviceToHost, s2)); };
            for (int w = 0; w < 2; ++w) { if (dir != 1) h2d(); if (dir != 0) d2h(); }
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            const double t0 = now();
            for (int i = 0; i < reps; ++i) { if (dir != 1) h2d(); if (dir != 0) d2h(); }
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            const double dt = (now() - t0) / reps;
            r[mode] = bytes / dt / 1e9;  // per direction
            if (mode == 3) { const double t1 = now(); h2d(); CK(hipStreamSynchronize(s1)); lat = (now() - t1) * 1e6; }
        }
        printf("%5d %6.1f |   %6.1f %6.1f %6.1f            |    %6.1f %6.1f %6.1f              | %8.1f\n", frames, bytes / 1e6, r[0], r[1], r[2], r[3], r[4], r[5], lat);
    }
    return 0;
}
