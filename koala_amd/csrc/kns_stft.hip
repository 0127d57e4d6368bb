// kns_stft.hip -- analysis (int16 -> STFT -> features), synthesis (mask x spectrum -> iSTFT -> OLA -> int16) and state reset;
// SURVEY.md 8a rows a2, a3, a5, a6.  See kns_kernels.h for the launch interface.
#include "kns_device.hpp"

namespace kns {

// ------------------------------------------------------------------------------------------------ analysis

template <class P>
__global__ __launch_bounds__(256) void analysis_kernel(AnalysisArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *tw = (float2 *) smem;                         // 4 KiB
    float *win = (float *) (smem + 4096);                 // 2 KiB
    float *fftbuf = (float *) (smem + 6144);              // 4 waves x kFftBufFloats
    typename P::elem_t *tile = (typename P::elem_t *) (smem + 6144 + 4 * kFftBufFloats * 4);  // nbf KiB, A-packed feature tile

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x, t = blockIdx.y;
    const int mtiles = g.Bpad >> 4;

    // all 16 sample loads of this wave's four frames go out before anything else: the kernel is latency-bound on them
    const size_t row_len = (size_t) g.T * kFrame;
    int raw[4][4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int b = mt * 16 + wave * 4 + f;
        const int16_t *cur = g.pcm + (size_t) b * row_len + (size_t) t * kFrame;
        const int16_t *old = (t == 0) ? g.hist_in + (size_t) b * kFrame : cur - kFrame;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = lane + 64 * r;
            const int16_t *src = (r < 2) ? old + 2 * n : cur + 2 * (n - 128);
            raw[f][r] = (b < g.B) ? *(const int *) src : 0;
        }
    }
    for (int i = tid; i < 512; i += 256) {
        tw[i] = ((const float2 *) g.twiddle)[i];
        win[i] = g.window[i];
    }
    // the normalisation constants of this lane's bins, once: loaded inside the frame loop they sit behind the spectrum
    // stores in the vector-memory queue, and waiting for them means waiting for the stores' acknowledgements
    float nmean[4], nscale[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        nmean[r] = g.mean[lane + 64 * r];
        nscale[r] = g.scale[lane + 64 * r];
    }
    const float nmean_nyq = g.mean[256], nscale_nyq = g.scale[256];
    {
        uint4 *z = (uint4 *) tile;
        for (int i = tid; i < g.nbf * 64; i += 256) z[i] = uint4{0, 0, 0, 0};
    }
    __syncthreads();

    float *buf = fftbuf + wave * kFftBufFloats;

#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int row = wave * 4 + f;
        const int b = mt * 16 + row;
        cpx v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = lane + 64 * r;
            const int pr = raw[f][r];
            float lo = (float) (int16_t) (pr & 0xffff), hi = (float) (int16_t) (pr >> 16);
            v[r].x = (lo * (1.0f / 32768.0f)) * win[2 * n];
            v[r].y = (hi * (1.0f / 32768.0f)) * win[2 * n + 1];
        }
        if (t == g.T - 1 && b < g.Bpad) {
            int *h = (int *) (g.hist_out + (size_t) b * kFrame);
            h[lane] = raw[f][2];
            h[lane + 64] = raw[f][3];
        }
        fft256_wave(v, buf, tw, lane);
        float2 *spec = (float2 *) g.spec + ((size_t) t * g.Bpad + b) * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r;
            const int kc = (256 - k) & 255;
            const cpx zk = ((const cpx *) buf)[k];
            cpx zc = ((const cpx *) buf)[kc];
            zc.y = -zc.y;
            float2 w = tw[k];
            cpx s = cadd(zk, zc), d = csub(zk, zc);
            cpx p = cmul(d, cpx{w.x, w.y});
            float xr = 0.5f * (s.x + p.y);
            float xi = 0.5f * (s.y - p.x);
            float pw = __builtin_fmaf(xr, xr, xi * xi);
            float nyq = 0.0f;
            if (k == 0) {  // DC and Nyquist share packed slot 0
                xr = zk.x + zk.y;
                nyq = zk.x - zk.y;
                xi = nyq;
                pw = xr * xr;
            }
            spec[k] = float2{xr, xi};
            float ft = (kns_log(pw + 1e-10f) - nmean[r]) * nscale[r];
            tile[(k / P::KB) * 64 * P::EPL + P::off(row, k % P::KB)] = P::cvt(ft);
            if (k == 0) {
                float fn = (kns_log(nyq * nyq + 1e-10f) - nmean_nyq) * nscale_nyq;
                tile[(256 / P::KB) * 64 * P::EPL + P::off(row, 256 % P::KB)] = P::cvt(fn);
            }
        }
        wave_lds_sync();
    }
    __syncthreads();
    {
        const uint4 *src = (const uint4 *) tile;
        uint4 *dst = (uint4 *) g.feat + ((size_t) t * mtiles + mt) * g.nbf * 64;
        for (int i = tid; i < g.nbf * 64; i += 256) dst[i] = src[i];
    }
}

void launch_analysis(const AnalysisArgs &a, hipStream_t s) {
    dim3 grid(a.Bpad / 16, a.T);
    size_t lds = 6144 + 4 * kFftBufFloats * 4 + (size_t) a.nbf * 1024;
    if (a.precision == kBf16)
        hipLaunchKernelGGL(analysis_kernel<PBF16>, grid, dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL(analysis_kernel<PF32>, grid, dim3(256), lds, s, a);
}

// ------------------------------------------------------------------------------------------------ synthesis

constexpr int kMaskLd = 273;  // row stride (floats) of the row-major mask tile in LDS: odd, so column walks are conflict-free

// 8 waves per workgroup, two streams per wave: the grid is only (streams / 16) x segments workgroups (512 at the bench
// size, two per CU), so with four waves each a SIMD held two waves and every wave paid its FFT's LDS round trips alone.
constexpr int kSynWaves = 8, kSynStreams = 16 / kSynWaves;

__global__ __launch_bounds__(64 * kSynWaves) void synthesis_kernel(SynthesisArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *tw = (float2 *) smem;
    float *win = (float *) (smem + 4096);
    float *fftbuf = (float *) (smem + 6144);
    float *mrow = (float *) (smem + 6144 + kSynWaves * kFftBufFloats * 4);  // [16][kMaskLd] fp32

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x;
    const int mtiles = g.Bpad >> 4;
    // this workgroup produces frames [t0, t1) of its 16 streams; a segment that does not start at 0 first replays
    // frame t0 - 1 (no output) to rebuild the overlap-add tail it inherits
    const int t0 = blockIdx.y * g.seg, t1 = min(g.T, t0 + g.seg);
    for (int i = tid; i < 512; i += 64 * kSynWaves) {
        tw[i] = ((const float2 *) g.twiddle)[i];
        win[i] = g.window[i];
    }
    float *buf = fftbuf + wave * kFftBufFloats;
    const size_t row_len = (size_t) g.T * kFrame;

    // overlap-add tail of this wave's streams: lane holds samples 2n, 2n+1 for n = lane, lane + 64
    float2 tl[kSynStreams][2];
#pragma unroll
    for (int f = 0; f < kSynStreams; ++f) {
        const int b = mt * 16 + wave * kSynStreams + f;
        const float2 *tp = (const float2 *) (g.tail_in + (size_t) b * kFrame);
        tl[f][0] = tp[lane];
        tl[f][1] = tp[lane + 64];
    }

    // software pipeline: the mask tile of frame t+1 and the spectrum of the next (frame, stream) are requested from HBM
    // before the current one is transformed; without it every wave sits out one memory latency per frame
    const int tb = (t0 > 0 ? t0 - 1 : 0);
    constexpr int kSynThreads = 64 * kSynWaves;
    constexpr int kMaskVecs = (kMaskTiles * 64 + kSynThreads - 1) / kSynThreads;  // f32x4 per thread per mask tile
    f32x4 mnext[kMaskVecs];
    auto mask_fetch = [&](int t) {
        const f32x4 *src = (const f32x4 *) g.mask + ((size_t) t * mtiles + mt) * kMaskTiles * 64;
#pragma unroll
        for (int j = 0; j < kMaskVecs; ++j) {
            const int i = tid + kSynThreads * j;
            if (i < kMaskTiles * 64) mnext[j] = src[i];
        }
    };
    // The spectrum of the next (frame, stream) is requested before the current one is transformed.  Two register sets
    // alternate between the (two) streams of a frame, so the set filled last in a frame is the one read first in the next
    // and the loop carries no register copies (with a single set hipcc rotated it at the back edge, which needs the data
    // -- and every store before it in the vector-memory queue -- to have arrived).  Every fetch and store in the loop is
    // unconditional: past the last frame the fetches re-read the last frame, and frames that must not be written get a
    // zero-length buffer descriptor; conditional vector-memory operations make hipcc's s_waitcnt placement drain the queue.
    float2 skA[4], scA[4], skB[4], scB[4];
    auto spec_fetch = [&](float2 (&sk)[4], float2 (&sc)[4], int t, int f) {
        const int b = mt * 16 + wave * kSynStreams + f;
        const float2 *spec = (const float2 *) g.spec + ((size_t) t * g.Bpad + b) * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r;
            sk[r] = spec[k];
            sc[r] = spec[(256 - k) & 255];
        }
    };
    mask_fetch(tb);
    spec_fetch(skA, scA, tb, 0);
    const unsigned lane4 = lane * 4u;

    auto stream = [&](const int t, const int f, const bool emit, float2 (&sk)[4], float2 (&sc)[4], float2 (&nk)[4],
                      float2 (&nc)[4]) {
        const int row = wave * kSynStreams + f;
        const int b = mt * 16 + row;
        const float *mk_row = mrow + row * kMaskLd;
        {
            const int tn = f < kSynStreams - 1 ? t : (t + 1 < t1 ? t + 1 : t);
            spec_fetch(nk, nc, tn, f < kSynStreams - 1 ? f + 1 : 0);
        }
        cpx v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r;
            float2 xk = sk[r], xc = sc[r];
            float mk = mk_row[k];
            float mc = mk_row[256 - k];  // mirrored bin (256 when k == 0)
            cpx yk, yc;
            if (k == 0) {
                yk = {mk * xk.x, 0.0f};
                yc = {mc * xk.y, 0.0f};
            } else {
                yk = {mk * xk.x, mk * xk.y};
                yc = {mc * xc.x, -(mc * xc.y)};
            }
            float2 w = tw[k];
            cpx e = cadd(yk, yc), d = csub(yk, yc);
            cpx o = cmul(d, cpx{w.x, -w.y});  // conj(W^k) (yk - yc)
            // Z' = E + i O (both carry the factor 1/2); fed to the forward FFT with re/im swapped = inverse FFT
            float zr = 0.5f * (e.x - o.y), zi = 0.5f * (e.y + o.x);
            v[r] = {zi, zr};
        }
        fft256_wave(v, buf, tw, lane);
        int packed[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = lane + 64 * r;
            // swapped output: re <-> im
            const cpx zz = ((const cpx *) buf)[n];
            float x0 = zz.y * (1.0f / 256.0f);
            float x1 = zz.x * (1.0f / 256.0f);
            float y0 = x0 * win[2 * n], y1 = x1 * win[2 * n + 1];
            if (r < 2) {
                float a0 = (tl[f][r].x + y0) * 32768.0f, a1 = (tl[f][r].y + y1) * 32768.0f;
                a0 = __builtin_fminf(__builtin_fmaxf(__builtin_roundf(a0), -32768.0f), 32767.0f);
                a1 = __builtin_fminf(__builtin_fmaxf(__builtin_roundf(a1), -32768.0f), 32767.0f);
                packed[r] = ((int) a0 & 0xffff) | ((int) a1 << 16);
            } else {
                tl[f][r - 2] = float2{y0, y1};
            }
        }
        {
            const bool wr = emit && b < g.B;
            const __amdgpu_buffer_rsrc_t o =
                make_rsrc(g.out + (wr ? (size_t) b * row_len + (size_t) t * kFrame : (size_t) 0), wr ? kFrame * 2u : 0u);
            __builtin_amdgcn_raw_buffer_store_b32(packed[0], o, lane4, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(packed[1], o, lane4 + 256u, 0, 0);
        }
        wave_lds_sync();
    };

    for (int t = tb; t < t1; ++t) {
        const bool emit = t >= t0;
        __syncthreads();  // previous frame's readers are done with the mask tile
        // C-packed fp32 tile [17][64 lanes][4 rows] -> row-major [16][kMaskLd]
#pragma unroll
        for (int j = 0; j < kMaskVecs; ++j) {
            const int i = tid + kSynThreads * j;
            if (i < kMaskTiles * 64) {
                const int nt = i >> 6, l = i & 63;
                const int col = nt * 16 + (l & 15), row = (l >> 4) * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) mrow[(row + r) * kMaskLd + col] = mnext[j][r];
            }
        }
        mask_fetch(t + 1 < t1 ? t + 1 : t);
        __syncthreads();
        static_assert(kSynStreams == 2, "the two prefetch sets alternate over an even number of streams");
        stream(t, 0, emit, skA, scA, skB, scB);
        stream(t, 1, emit, skB, scB, skA, scA);
    }
    if (t1 == g.T) {
#pragma unroll
        for (int f = 0; f < kSynStreams; ++f) {
            const int b = mt * 16 + wave * kSynStreams + f;
            float2 *tp = (float2 *) (g.tail_out + (size_t) b * kFrame);
            tp[lane] = tl[f][0];
            tp[lane + 64] = tl[f][1];
        }
    }
}

void launch_synthesis(const SynthesisArgs &a, hipStream_t s) {
    size_t lds = 6144 + kSynWaves * kFftBufFloats * 4 + 16 * kMaskLd * 4;
    hipLaunchKernelGGL(synthesis_kernel, dim3(a.Bpad / 16, (a.T + a.seg - 1) / a.seg), dim3(64 * kSynWaves), lds, s, a);
}

// ------------------------------------------------------------------------------------------------ reset

__global__ void reset_kernel(ResetArgs g) {
    // one workgroup per stream: history, overlap-add tail, and this stream's row of the 8 hidden-state tiles
    const int b = blockIdx.x, tid = threadIdx.x;
    if (g.mask && !g.mask[b]) return;
    g.hist[(size_t) b * kFrame + tid] = 0;
    g.hist2[(size_t) b * kFrame + tid] = 0;
    g.tail[(size_t) b * kFrame + tid] = 0.0f;
    g.tail2[(size_t) b * kFrame + tid] = 0.0f;
    const int mtiles = g.Bpad >> 4, mt = b >> 4, row = b & 15;
    for (int i = tid; i < kGruLayers * kUnitTiles * 16; i += 256) {
        const int col = i & 15, u = (i >> 4) % kUnitTiles, layer = (i >> 4) / kUnitTiles;
        const size_t idx = (((size_t) layer * mtiles + mt) * kUnitTiles + u) * 256 + cpack_off(row, col);
        g.hstate[idx] = 0.0f;
        g.hstate2[idx] = 0.0f;
    }
}

void launch_reset(const ResetArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(reset_kernel, dim3(a.Bpad), dim3(256), 0, s, a);
}

}  // namespace kns
