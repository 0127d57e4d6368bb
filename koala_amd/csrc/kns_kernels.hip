// kns_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the KNS-v1 noise suppressor.
//
// Hot path of pv_koala_process (reference include/pv_koala.h:65-80), batched over B independent streams and
// T frames per call:
//   analysis_kernel   a2+a3  int16 -> window -> real FFT-512 (radix-4 Stockham, one frame per wavefront, LDS
//                            exchange) -> log-power features, written in MFMA A-fragment order
//   gemm_kernel       a4     every input-side / front-end / head GEMM: A tile staged once in LDS in fragment
//                            order, weights streamed from L2 in B-fragment order, MFMA 16x16x32 bf16 or 16x16x4 f32
//   gru_kernel        a4     recurrent half of a GRU layer: one workgroup owns 16 streams for all T frames, hidden
//                            state in registers (fp32) and LDS (operand type), no inter-workgroup traffic
//   synthesis_kernel  a5     mask x spectrum -> inverse real FFT -> window -> overlap-add -> saturated int16
//
// Numerics follow DESIGN.md section 2 exactly (k-ascending fmaf chains, polynomial exp/log built from IEEE ops),
// compiled with -ffp-contract=off so that no operation is fused or split behind the spec's back.
#include "kns_kernels.h"

#include <stdlib.h>

namespace kns {

#ifdef KNS_TIMING
__device__ unsigned long long g_kns_timing[64];
#define KNS_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && t == 5) g_kns_timing[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define KNS_STAMP_WS(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && mt == 5 * (int) gridDim.x) g_kns_timing[16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define KNS_STAMP(i) do { } while (0)
#define KNS_STAMP_WS(i) do { } while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ scalar math

__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }

__device__ __forceinline__ uint16_t f2bf(float x) {  // round to nearest even
    uint32_t u = f2u(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t) (u >> 16);
}

__device__ __forceinline__ float kns_exp(float x) {
    x = __builtin_fminf(__builtin_fmaxf(x, -87.0f), 88.0f);
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float z = r * r;
    float y = __builtin_fmaf(p, z, r) + 1.0f;
    int ni = (int) n;
    return y * u2f((uint32_t) (ni + 127) << 23);
}

__device__ __forceinline__ float kns_log(float x) {
    uint32_t u = f2u(x);
    int e = (int) ((u >> 23) & 0xffu) - 126;
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, m, -1.1514610310e-1f);
    p = __builtin_fmaf(p, m, 1.1676998740e-1f);
    p = __builtin_fmaf(p, m, -1.2420140846e-1f);
    p = __builtin_fmaf(p, m, 1.4249322787e-1f);
    p = __builtin_fmaf(p, m, -1.6668057665e-1f);
    p = __builtin_fmaf(p, m, 2.0000714765e-1f);
    p = __builtin_fmaf(p, m, -2.4999993993e-1f);
    p = __builtin_fmaf(p, m, 3.3333331174e-1f);
    float fe = (float) e;
    float y = (p * m) * z;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(z, -0.5f, y);
    float r = m + y;
    return __builtin_fmaf(fe, 0.693359375f, r);
}

__device__ __forceinline__ float kns_sigmoid(float x) { return 1.0f / (1.0f + kns_exp(-x)); }

__device__ __forceinline__ float kns_tanh(float x) {
    float a = __builtin_fabsf(x);
    float t = kns_exp(-2.0f * a);
    float v = (1.0f - t) / (1.0f + t);
    return __builtin_copysignf(v, x);
}

// LDS traffic between the lanes of ONE wavefront: DS operations of a wave execute in program order, so only the
// compiler has to be kept from reordering across this point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------ precision traits

struct PF32 {
    static constexpr int kPrec = kFp32;
    static constexpr int KB = 16, EPL = 4, NPB = 1;
    static constexpr int NBH = 17;  // k-blocks covering the 271 hidden units
    static constexpr bool kHoldA = false;  // 17 x 4 registers of A fragments would spill: re-read them from LDS
    static constexpr int kGruWaves = 1;
    typedef f32x4 frag_t;
    typedef f32x4 gi_t;
    typedef float elem_t;
    static __device__ __forceinline__ f32x4 mma(frag_t a, frag_t b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ int off(int rc, int kk) { return pack_off_f32(rc, kk); }
    static __device__ __forceinline__ elem_t cvt(float v) { return v; }
    static __device__ __forceinline__ gi_t to_gi(f32x4 v) { return v; }
    static __device__ __forceinline__ f32x4 from_gi(gi_t v) { return v; }
};

struct PBF16 {
    static constexpr int kPrec = kBf16;
    static constexpr int KB = 32, EPL = 8, NPB = 2;
    static constexpr int NBH = 9;
    static constexpr bool kHoldA = true;
    static constexpr int kGruWaves = 1;
    typedef bf16x8 frag_t;
    typedef f16x4 gi_t;
    typedef uint16_t elem_t;
    static __device__ __forceinline__ f32x4 mma(frag_t a, frag_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int off(int rc, int kk) { return pack_off_bf16(rc, kk); }
    static __device__ __forceinline__ elem_t cvt(float v) { return f2bf(v); }
    static __device__ __forceinline__ gi_t to_gi(f32x4 v) { return __builtin_convertvector(v, gi_t); }
    static __device__ __forceinline__ f32x4 from_gi(gi_t v) {
        f32x4 r;
        r[0] = (float) v[0];
        r[1] = (float) v[1];
        r[2] = (float) v[2];
        r[3] = (float) v[3];
        return r;
    }
};

// ------------------------------------------------------------------------------------------------ FFT-256 per wave

// complex numbers as packed pairs: add/sub are one v_pk_add_f32, a complex multiply is v_pk_mul_f32 + v_pk_fma_f32
typedef float cpx __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return a + b; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return a - b; }
__device__ __forceinline__ cpx cmul(cpx a, cpx w) {
    const cpx t = cpx{a.y, a.y} * cpx{-w.y, w.x};
    return __builtin_elementwise_fma(cpx{a.x, a.x}, w, t);
}

__device__ __forceinline__ void radix4(cpx (&v)[4]) {
    cpx a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]), a2 = cadd(v[1], v[3]), d = csub(v[1], v[3]);
    cpx a3 = {d.y, -d.x};  // (v1 - v3) * (-i)
    v[0] = cadd(a0, a2);
    v[1] = cadd(a1, a3);
    v[2] = csub(a0, a2);
    v[3] = csub(a1, a3);
}

// LDS exchange buffers of the wave FFT: two ping-pong arrays of 256 interleaved complex values (ds_read/write_b64
// straight into the packed register pairs).  Padding them against the 4-way write conflicts of the first two radix-4
// stages was measured SLOWER on MI355X (synthesis 246 vs 193 us): the extra address VALU costs more than the conflicts.
constexpr int kFftBufFloats = 4 * 256;  // per wave: 2 buffers x 256 complex

// Forward 256-point complex FFT of one wavefront, radix-4 Stockham autosort.  On entry v[r] = z[lane + 64 r].
// `buf` is this wave's LDS scratch (kFftBufFloats floats).  On return the spectrum is in natural order in the first
// buffer (((cpx *) buf)[k]) and visible to the whole wave.  tw = exp(-2 pi i k / 512), k = 0..511, in LDS.
__device__ __forceinline__ void fft256_wave(cpx (&v)[4], float *buf, const float2 *tw, int lane) {
    cpx *b0 = (cpx *) buf, *b1 = (cpx *) buf + 256;
    // stage Ns = 1 (all twiddles are 1)
    radix4(v);
#pragma unroll
    for (int r = 0; r < 4; ++r) b1[4 * lane + r] = v[r];
    wave_lds_sync();
    // stages Ns = 4, 16, 64
#pragma unroll
    for (int s = 1; s < 4; ++s) {
        const int Ns = 1 << (2 * s);
        cpx *src = (s & 1) ? b1 : b0;
        cpx *dst = (s & 1) ? b0 : b1;
        const int k = lane & (Ns - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = src[lane + 64 * r];
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            float2 w = tw[r * k * (128 / Ns)];
            v[r] = cmul(v[r], cpx{w.x, w.y});
        }
        radix4(v);
        const int j0 = (lane / Ns) * Ns * 4 + k;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[j0 + r * Ns] = v[r];
        wave_lds_sync();
    }
    // s = 1 -> b0, s = 2 -> b1, s = 3 -> b0: result is in b0
}

// ------------------------------------------------------------------------------------------------ analysis

template <class P>
__global__ __launch_bounds__(256) void analysis_kernel(AnalysisArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *tw = (float2 *) smem;                         // 4 KiB
    float *win = (float *) (smem + 4096);                 // 2 KiB
    float *fftbuf = (float *) (smem + 6144);              // 4 waves x kFftBufFloats
    typename P::elem_t *tile = (typename P::elem_t *) (smem + 6144 + 4 * kFftBufFloats * 4);  // nbf KiB, A-packed feature tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
            float ft = (kns_log(pw + 1e-10f) - g.mean[k]) * g.scale[k];
            tile[(k / P::KB) * 64 * P::EPL + P::off(row, k % P::KB)] = P::cvt(ft);
            if (k == 0) {
                float fn = (kns_log(nyq * nyq + 1e-10f) - g.mean[256]) * g.scale[256];
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

__global__ __launch_bounds__(256) void synthesis_kernel(SynthesisArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *tw = (float2 *) smem;
    float *win = (float *) (smem + 4096);
    float *fftbuf = (float *) (smem + 6144);
    float *mrow = (float *) (smem + 6144 + 4 * kFftBufFloats * 4);  // [16][kMaskLd] fp32

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mt = blockIdx.x;
    const int mtiles = g.Bpad >> 4;
    // this workgroup produces frames [t0, t1) of its 16 streams; a segment that does not start at 0 first replays
    // frame t0 - 1 (no output) to rebuild the overlap-add tail it inherits
    const int t0 = blockIdx.y * g.seg, t1 = min(g.T, t0 + g.seg);
    for (int i = tid; i < 512; i += 256) {
        tw[i] = ((const float2 *) g.twiddle)[i];
        win[i] = g.window[i];
    }
    float *buf = fftbuf + wave * kFftBufFloats;
    const size_t row_len = (size_t) g.T * kFrame;

    // overlap-add tail of this wave's four streams: lane holds samples 2n, 2n+1 for n = lane, lane + 64
    float2 tl[4][2];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int b = mt * 16 + wave * 4 + f;
        const float2 *tp = (const float2 *) (g.tail_in + (size_t) b * kFrame);
        tl[f][0] = tp[lane];
        tl[f][1] = tp[lane + 64];
    }

    // software pipeline: the mask tile of frame t+1 and the spectrum of the next (frame, stream) are requested from HBM
    // before the current one is transformed; without it every wave sits out one memory latency per frame
    const int tb = (t0 > 0 ? t0 - 1 : 0);
    constexpr int kMaskVecs = (kMaskTiles * 64 + 255) / 256;  // f32x4 per thread per mask tile
    f32x4 mnext[kMaskVecs];
    auto mask_fetch = [&](int t) {
        const f32x4 *src = (const f32x4 *) g.mask + ((size_t) t * mtiles + mt) * kMaskTiles * 64;
#pragma unroll
        for (int j = 0; j < kMaskVecs; ++j) {
            const int i = tid + 256 * j;
            if (i < kMaskTiles * 64) mnext[j] = src[i];
        }
    };
    float2 sk[4], sc[4];
    auto spec_fetch = [&](int t, int f) {
        const int b = mt * 16 + wave * 4 + f;
        const float2 *spec = (const float2 *) g.spec + ((size_t) t * g.Bpad + b) * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r;
            sk[r] = spec[k];
            sc[r] = spec[(256 - k) & 255];
        }
    };
    mask_fetch(tb);
    spec_fetch(tb, 0);

    for (int t = tb; t < t1; ++t) {
        const bool emit = t >= t0;
        __syncthreads();  // previous frame's readers are done with the mask tile
        // C-packed fp32 tile [17][64 lanes][4 rows] -> row-major [16][kMaskLd]
#pragma unroll
        for (int j = 0; j < kMaskVecs; ++j) {
            const int i = tid + 256 * j;
            if (i < kMaskTiles * 64) {
                const int nt = i >> 6, l = i & 63;
                const int col = nt * 16 + (l & 15), row = (l >> 4) * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) mrow[(row + r) * kMaskLd + col] = mnext[j][r];
            }
        }
        if (t + 1 < t1) mask_fetch(t + 1);
        __syncthreads();
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int row = wave * 4 + f;
            const int b = mt * 16 + row;
            const float *mk_row = mrow + row * kMaskLd;
            float2 ck[4], cc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ck[r] = sk[r];
                cc[r] = sc[r];
            }
            if (f < 3)
                spec_fetch(t, f + 1);
            else if (t + 1 < t1)
                spec_fetch(t + 1, 0);
            cpx v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = lane + 64 * r;
                float2 xk = ck[r], xc = cc[r];
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
            if (emit && b < g.B) {
                int *o = (int *) (g.out + (size_t) b * row_len + (size_t) t * kFrame);
                o[lane] = packed[0];
                o[lane + 64] = packed[1];
            }
            wave_lds_sync();
        }
    }
    if (t1 == g.T) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int b = mt * 16 + wave * 4 + f;
            float2 *tp = (float2 *) (g.tail_out + (size_t) b * kFrame);
            tp[lane] = tl[f][0];
            tp[lane + 64] = tl[f][1];
        }
    }
}

void launch_synthesis(const SynthesisArgs &a, hipStream_t s) {
    size_t lds = 6144 + 4 * kFftBufFloats * 4 + 16 * kMaskLd * 4;
    hipLaunchKernelGGL(synthesis_kernel, dim3(a.Bpad / 16, (a.T + a.seg - 1) / a.seg), dim3(256), lds, s, a);
}

// ------------------------------------------------------------------------------------------------ GEMM

constexpr int kGemmMT = 4;  // m-tiles (of 16 stream-frames) per workgroup
constexpr int kPF = 4;      // weight prefetch depth in k-blocks

template <class P, int OUT>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename P::frag_t frag_t;
    constexpr bool kApack = (OUT == kOutAPlain || OUT == kOutASigmoid);
    constexpr bool kSigmoid = (OUT == kOutMask || OUT == kOutASigmoid);
    constexpr int NU = kApack ? P::NPB : 1;  // n-tiles per unit of work

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = g.nb0 + g.nb1;
    const int mt0 = blockIdx.x * kGemmMT;
    const int mcount = min(kGemmMT, g.mtiles - mt0);

    // stage the A tile in LDS, keeping fragment order: [m-tile][k-block][lane] 16-byte words
    uint4 *lds_a = (uint4 *) smem;
    for (int m = 0; m < mcount; ++m) {
        if (g.nb0) {
            const uint4 *src = (const uint4 *) g.a0 + (size_t) (mt0 + m) * g.nb0 * 64;
            for (int i = tid; i < g.nb0 * 64; i += 256) lds_a[m * nb * 64 + i] = src[i];
        }
        const uint4 *src1 = (const uint4 *) g.a1 + (size_t) (mt0 + m) * g.nb1 * 64;
        for (int i = tid; i < g.nb1 * 64; i += 256) lds_a[(m * nb + g.nb0) * 64 + i] = src1[i];
    }
    for (int m = mcount; m < kGemmMT; ++m)
        for (int i = tid; i < nb * 64; i += 256) lds_a[m * nb * 64 + i] = uint4{0, 0, 0, 0};
    __syncthreads();

    const frag_t *lds_f = (const frag_t *) smem;
    char *scratch = smem + (size_t) kGemmMT * nb * 1024 + (size_t) wave * kGemmMT * 1024;  // per-wave transposer

    const int units = g.ntiles / NU;
    const int units_per_y = ceil_div(units, (int) gridDim.y);
    const int u_begin = blockIdx.y * units_per_y;
    const int u_end = min(units, u_begin + units_per_y);
    const frag_t *w = (const frag_t *) g.w;

    for (int u = u_begin + wave; u < u_end; u += 4) {
        const int nt0 = u * NU;
        f32x4 acc[NU][kGemmMT];
#pragma unroll
        for (int j = 0; j < NU; ++j)
#pragma unroll
            for (int m = 0; m < kGemmMT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};

        frag_t bq[kPF][NU];
#pragma unroll
        for (int p = 0; p < kPF; ++p)
            if (p < nb)
#pragma unroll
                for (int j = 0; j < NU; ++j) bq[p][j] = w[((size_t) (nt0 + j) * nb + p) * 64 + lane];
        for (int blk0 = 0; blk0 < nb; blk0 += kPF) {
#pragma unroll
            for (int p = 0; p < kPF; ++p) {
                const int blk = blk0 + p;
                if (blk < nb) {
                    frag_t bc[NU];
#pragma unroll
                    for (int j = 0; j < NU; ++j) bc[j] = bq[p][j];
                    if (blk + kPF < nb)
#pragma unroll
                        for (int j = 0; j < NU; ++j) bq[p][j] = w[((size_t) (nt0 + j) * nb + blk + kPF) * 64 + lane];
#pragma unroll
                    for (int m = 0; m < kGemmMT; ++m) {
                        frag_t a = lds_f[(m * nb + blk) * 64 + lane];
#pragma unroll
                        for (int j = 0; j < NU; ++j) acc[j][m] = P::mma(a, bc[j], acc[j][m]);
                    }
                }
            }
        }

        // epilogue: lane owns column (lane & 15) of each n-tile, rows (lane >> 4) * 4 + i
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const int nt = nt0 + j;
            const int col = nt * 16 + (lane & 15);
            const float bias = g.bias[col];
#pragma unroll
            for (int m = 0; m < kGemmMT; ++m) {
                f32x4 v = acc[j][m];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = v[i] + bias;
                    if (kSigmoid) x = kns_sigmoid(x);
                    if (kApack && col >= g.n_valid) x = 0.0f;
                    v[i] = x;
                }
                if (!kApack) {
                    if (m < mcount) {
                        const size_t idx = ((size_t) (mt0 + m) * g.ntiles + nt) * 64 + lane;
                        if (OUT == kOutGi)
                            ((typename P::gi_t *) g.out)[idx] = P::to_gi(v);
                        else
                            ((f32x4 *) g.out)[idx] = v;
                    }
                } else {
                    typename P::elem_t *sc = (typename P::elem_t *) (scratch + m * 1024);
#pragma unroll
                    for (int i = 0; i < 4; ++i) sc[P::off((lane >> 4) * 4 + i, j * 16 + (lane & 15))] = P::cvt(v[i]);
                }
            }
        }
        if (kApack) {
            wave_lds_sync();
            const int out_nb = g.ntiles / NU;
            for (int m = 0; m < mcount; ++m) {
                uint4 word = ((const uint4 *) (scratch + m * 1024))[lane];
                ((uint4 *) g.out)[((size_t) (mt0 + m) * out_nb + u) * 64 + lane] = word;
            }
            wave_lds_sync();
        }
    }
}

constexpr int kGruTilesPerWave = 5;  // 17 unit tiles over 4 waves: 5,4,4,4

// ---- bf16 recurrent kernel with the layer's W_hh RESIDENT on the CU for all T steps ("persistent RNN"):
// 459 KiB of B-fragments = 4 waves x 3 unit tiles x 27 blocks in VGPRs (324 registers per lane, one wave per SIMD with
// the whole 512-register file) + 4 x 27 KiB + 27 KiB in LDS.  Per step a wave then needs only the 16 x 288 bf16 hidden
// tile from LDS and its 15/12 pre-activation tiles from HBM: no weight traffic at all after the prologue.
// Gate nonlinearities use the hardware transcendentals (v_exp_f32, v_rcp_f32): the bf16 configuration is specified to a
// tolerance, not bit for bit.

__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
__device__ __forceinline__ float fast_tanh(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.88539008177792681f));
}

constexpr int kResTileBytes = 3 * PBF16::NBH * 1024;  // one unit tile of W_hh: 3 gates x 9 k-blocks x 1 KiB
constexpr int kResBiasBytes = kGateTiles * 16 * 4;
constexpr int kResLds = 2 * PBF16::NBH * 1024 + 5 * kResTileBytes + kResBiasBytes;  // h double buffer, 4 + 1 tiles, b_hh

// hipcc keeps values it loaded itself in VGPRs and reaches the accumulator half of the register file only through
// v_accvgpr copies.  Passing a fragment once through an "a"-constrained empty asm re-defines it as an AGPR value; the
// MFMA builtins then take it as an AGPR source operand directly, so 216 registers of weights cost no VGPR and no copy.
__device__ __forceinline__ bf16x8 pin_to_agpr(bf16x8 w) {
    asm volatile("" : "+a"(w));
    return w;
}

// acc[gt] += a[blk] . W for one LDS-resident unit tile stored as [k-block][gate][lane]; the explicit queue keeps
// kQueue ds_read_b128 in flight so that LDS latency is not paid per MFMA
template <int NBH, int kQueue>
__device__ __forceinline__ void mma_lds_tile(f32x4 (&acc)[3], const bf16x8 (&a)[NBH], const bf16x8 *wl, int lane) {
    constexpr int N = 3 * NBH;
    bf16x8 qb[kQueue];
#pragma unroll
    for (int p = 0; p < kQueue; ++p) qb[p] = wl[p * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);  // pin the order: without it hipcc sinks each read next to its MFMA
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const bf16x8 b = qb[i % kQueue];
        if (i + kQueue < N) qb[i % kQueue] = wl[(i + kQueue) * 64 + lane];
        acc[i % 3] = PBF16::mma(a[i / 3], b, acc[i % 3]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- weight-stationary form of the GRU input-side GEMM (bf16):  Gi = [y_prev ; e] . W_ih + b_ih  over ALL stream-frames.
// The 51 x 9 e-part blocks of W_ih stay on the CU exactly as W_hh does in gru_resident_kernel (VGPR / AGPR / LDS per
// wave); the y_prev part (0..2 k-blocks) is re-read from L2 per m-tile.  A persistent workgroup walks over m-tiles:
// A fragments come straight from HBM in fragment order (no LDS, no barrier), C tiles leave as fp16 fragments.
// HBM traffic per m-tile is the algorithmic minimum: 9-11 KiB in, 25.5 KiB out.
constexpr int kWsABlocks = PBF16::NBH + 2;                               // k-blocks of one A tile (y part <= 2)
constexpr int kWsLds = 5 * kResTileBytes + 2 * kWsABlocks * 1024;        // weight tiles 3, 4 + A double buffer

template <int NB0>
__global__ __launch_bounds__(256, 1) void gemm_ws_kernel(GemmArgs g) {
    typedef PBF16 P;
    typedef P::frag_t frag_t;
    constexpr int NBH = P::NBH;
    __shared__ __attribute__((aligned(16))) char smem[kWsLds];
    frag_t *abuf = (frag_t *) (smem + 5 * kResTileBytes);  // [2][kWsABlocks][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int colq = lane & 15;
    constexpr int nb0 = NB0, nb = NB0 + NBH;
    const frag_t *w = (const frag_t *) g.w;

    frag_t wv[3][NBH];
    frag_t wa[2][3][NBH];
#pragma unroll
    for (int gt = 0; gt < 3; ++gt)
#pragma unroll
        for (int blk = 0; blk < NBH; ++blk) wv[gt][blk] = w[((size_t) (wave * 3 + gt) * nb + nb0 + blk) * 64 + lane];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int gt = 0; gt < 3; ++gt)
#pragma unroll
            for (int blk = 0; blk < NBH; ++blk) {
                wa[q][gt][blk] = pin_to_agpr(w[((size_t) ((wave + 4 * (q + 1)) * 3 + gt) * nb + nb0 + blk) * 64 + lane]);
            }
    frag_t *wl3 = (frag_t *) (smem + wave * kResTileBytes);
    frag_t *wl4 = (frag_t *) (smem + 4 * kResTileBytes);
    for (int gt = 0; gt < 3; ++gt)
        for (int blk = 0; blk < NBH; ++blk)
            wl3[(blk * 3 + gt) * 64 + lane] = w[((size_t) ((wave + 12) * 3 + gt) * nb + nb0 + blk) * 64 + lane];
    if (wave == 0)
        for (int gt = 0; gt < 3; ++gt)
            for (int blk = 0; blk < NBH; ++blk)
                wl4[(blk * 3 + gt) * 64 + lane] = w[((size_t) (16 * 3 + gt) * nb + nb0 + blk) * 64 + lane];
    float bias[kGruTilesPerWave][3];
#pragma unroll
    for (int q = 0; q < kGruTilesPerWave; ++q)
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) {
            const int u = wave + 4 * q;
            bias[q][gt] = u < kUnitTiles ? g.bias[(u * 3 + gt) * 16 + colq] : 0.0f;
        }

    // A tile staging: block j of an m-tile (j < nb0: y part, else e part) is fetched by wave j & 3
    const frag_t *a0p = (const frag_t *) g.a0;
    const frag_t *a1p = (const frag_t *) g.a1;
    auto fetch = [&](int mtile, int j) -> frag_t {
        return j < nb0 ? a0p[((size_t) mtile * nb0 + j) * 64 + lane] : a1p[((size_t) mtile * NBH + (j - nb0)) * 64 + lane];
    };
    int mt = blockIdx.x;
    if (mt < g.mtiles)
        for (int j = wave; j < nb; j += 4) abuf[j * 64 + lane] = fetch(mt, j);
    __syncthreads();

    int cur = 0;
    for (; mt < g.mtiles; mt += gridDim.x) {
        KNS_STAMP_WS(0);
        const int mn = mt + gridDim.x;
        frag_t stage[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = wave + 4 * i;
            if (mn < g.mtiles && j < nb) stage[i] = fetch(mn, j);  // in flight during this tile's MFMAs
        }
        const frag_t *ab = abuf + cur * kWsABlocks * 64;
        frag_t cy[NB0 > 0 ? NB0 : 1], ce[NBH];
#pragma unroll
        for (int blk = 0; blk < NB0; ++blk) cy[blk] = ab[blk * 64 + lane];
#pragma unroll
        for (int blk = 0; blk < NBH; ++blk) ce[blk] = ab[(nb0 + blk) * 64 + lane];
        KNS_STAMP_WS(1);
        P::gi_t *out = (P::gi_t *) g.out + (size_t) mt * kGateTiles * 64;
#pragma unroll
        for (int q = 0; q < kGruTilesPerWave; ++q) {
            const int u = wave + 4 * q;
            {
                f32x4 acc[3];
#pragma unroll
                for (int gt = 0; gt < 3; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (NB0 > 0) {
                    frag_t wy[NB0 > 0 ? NB0 : 1][3];
                    __builtin_amdgcn_sched_barrier(0);  // keep the streamed y-part loads of other tiles out of here
#pragma unroll
                    for (int blk = 0; blk < NB0; ++blk)
#pragma unroll
                        for (int gt = 0; gt < 3; ++gt)
                            wy[blk][gt] = w[((size_t) ((u < kUnitTiles ? u : 0) * 3 + gt) * nb + blk) * 64 + lane];
#pragma unroll
                    for (int blk = 0; blk < NB0; ++blk)
#pragma unroll
                        for (int gt = 0; gt < 3; ++gt) acc[gt] = P::mma(cy[blk], wy[blk][gt], acc[gt]);
                }
                if (q < 3) {
#pragma unroll
                    for (int blk = 0; blk < NBH; ++blk)
#pragma unroll
                        for (int gt = 0; gt < 3; ++gt)
                            acc[gt] = P::mma(ce[blk], q == 0 ? wv[gt][blk] : wa[q == 2 ? 1 : 0][gt][blk], acc[gt]);
                } else {
                    mma_lds_tile<NBH, 4>(acc, ce, q == 3 ? wl3 : wl4, lane);
                }
#pragma unroll
                for (int gt = 0; gt < 3; ++gt) {
                    f32x4 v = acc[gt];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = v[i] + bias[q][gt];
                    if (q < 4 || wave == 0) out[(u * 3 + gt) * 64 + lane] = P::to_gi(v);
                }
                KNS_STAMP_WS(2 + q);
            }
        }
        frag_t *an = abuf + (cur ^ 1) * kWsABlocks * 64;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = wave + 4 * i;
            if (mn < g.mtiles && j < nb) an[j * 64 + lane] = stage[i];
        }
        KNS_STAMP_WS(7);
        __syncthreads();
        KNS_STAMP_WS(8);
        cur ^= 1;
    }
}

// ---- second form of the weight-stationary input GEMM: the 51 n-tiles are split over a PAIR of workgroups that sit on
// the same XCD (blocks g and g + 8), so one wave keeps at most 7 n-tiles x (9 + NB0) k-blocks = 77 fragments -- all of
// them in registers (3 n-tiles in VGPRs, 4 pinned in AGPRs), including the y_prev part.  No weight ever comes from LDS
// or L2 inside the loop; LDS only double-buffers A tiles, kWs2Stage m-tiles per barrier.  The partner's second read of
// an A tile hits the XCD's L2.
constexpr int kWs2Waves = 8;    // two waves per SIMD: one wave's MFMAs run under the other's epilogue VALU
constexpr int kWs2Tiles = 4;    // n-tiles per wave (the 4th only on some waves): 26 / 8 -> 4,4,3,...

template <int NB0, int kWs2Stage>  // kWs2Stage: m-tiles staged per barrier
__global__ __launch_bounds__(64 * kWs2Waves, 2) void gemm_ws2_kernel(GemmArgs g) {
    typedef PBF16 P;
    typedef P::frag_t frag_t;
    constexpr int NB = P::NBH + NB0;
    __shared__ __attribute__((aligned(16))) char smem[2 * kWs2Stage * NB * 1024];
    frag_t *abuf = (frag_t *) smem;  // [2][kWs2Stage][NB][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int colq = lane & 15;
    const int bid = blockIdx.x;
    const int half = (bid >> 3) & 1;
    const int mgroup = (bid >> 4) * 8 + (bid & 7);      // 0 .. gridDim.x / 2 - 1
    const int mstride = (gridDim.x >> 1) * kWs2Stage;   // m-tiles between consecutive stages of this workgroup
    const int nt_base = half ? 26 : 0, nt_count = half ? kGateTiles - 26 : 26;
    const frag_t *w = (const frag_t *) g.w;

    // this wave's n-tiles: nt_base + wave + 8 j; j < 3 always exists, j = 3 only on the first waves of a half
    int nt[kWs2Tiles];
#pragma unroll
    for (int j = 0; j < kWs2Tiles; ++j) nt[j] = nt_base + (wave + kWs2Waves * j < nt_count ? wave + kWs2Waves * j : 0);
    const bool has4 = wave + kWs2Waves * 3 < nt_count;
    frag_t wr[kWs2Tiles][NB];
    float bias[kWs2Tiles];
#pragma unroll
    for (int j = 0; j < kWs2Tiles; ++j) {
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) wr[j][blk] = w[((size_t) nt[j] * NB + blk) * 64 + lane];
        bias[j] = g.bias[nt[j] * 16 + colq];
    }

    // A staging: block i of a stage (i = m * NB + blk; blk < NB0 is the y part) is fetched by wave i % 8.
    // The launch guarantees mtiles % (kWs2Stage * gridDim.x / 2) == 0, so every staged m-tile exists.
    constexpr int kStageBlocks = kWs2Stage * NB;
    constexpr int kFetch = (kStageBlocks + kWs2Waves - 1) / kWs2Waves;
    const frag_t *src[kFetch];
    size_t step[kFetch];
#pragma unroll
    for (int i = 0; i < kFetch; ++i) {
        const int idx = wave + kWs2Waves * i;
        const int m = idx / NB, blk = idx % NB;
        const int mt = mgroup * kWs2Stage + m;
        if (blk < NB0) {
            src[i] = (const frag_t *) g.a0 + ((size_t) mt * NB0 + blk) * 64 + lane;
            step[i] = (size_t) mstride * NB0 * 64;
        } else {
            src[i] = (const frag_t *) g.a1 + ((size_t) mt * P::NBH + (blk - NB0)) * 64 + lane;
            step[i] = (size_t) mstride * P::NBH * 64;
        }
    }
#pragma unroll
    for (int i = 0; i < kFetch; ++i)
        if (wave + kWs2Waves * i < kStageBlocks) abuf[(wave + kWs2Waves * i) * 64 + lane] = *src[i];
    __syncthreads();

    auto store_tile = [&](P::gi_t *out, int j, f32x4 v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] + bias[j];
        out[nt[j] * 64 + lane] = P::to_gi(v);
    };

    int cur = 0;
    for (int mt0 = mgroup * kWs2Stage; mt0 < g.mtiles; mt0 += mstride) {
        const bool more = mt0 + mstride < g.mtiles;
        frag_t stage[kFetch];
#pragma unroll
        for (int i = 0; i < kFetch; ++i) {
            src[i] += step[i];
            if (more && wave + kWs2Waves * i < kStageBlocks) stage[i] = *src[i];  // in flight during this stage's MFMAs
        }
#pragma unroll
        for (int m = 0; m < kWs2Stage; ++m) {
            const frag_t *ab = abuf + (cur * kStageBlocks + m * NB) * 64;
            P::gi_t *out = (P::gi_t *) g.out + (size_t) (mt0 + m) * kGateTiles * 64;
            // three independent accumulator chains over the always-present n-tiles; A fragments streamed from LDS
            f32x4 acc[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 acc3 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const frag_t a = ab[blk * 64 + lane];
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = P::mma(a, wr[c][blk], acc[c]);
                if (has4) acc3 = P::mma(a, wr[3][blk], acc3);
            }
            if (m == kWs2Stage - 1) {
                // hand the next stage's A tiles to LDS before this m-tile's stores are issued: vmcnt counts stores too on
                // gfx950, so a wait placed after them would also wait for their write acknowledgements
#pragma unroll
                for (int i = 0; i < kFetch; ++i)
                    if (more && wave + kWs2Waves * i < kStageBlocks)
                        abuf[((cur ^ 1) * kStageBlocks + wave + kWs2Waves * i) * 64 + lane] = stage[i];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) store_tile(out, c, acc[c]);
            if (has4) store_tile(out, 3, acc3);
        }
        __syncthreads();
        cur ^= 1;
    }
}

// ---- weight-stationary form of the narrow GEMMs (front-end 257->271, heads 271->{1,5,40,257}; bf16): 8 waves per
// workgroup, every wave keeps its n-tiles' 9 k-blocks in VGPRs (at most 36 fragments), a persistent workgroup walks
// m-tiles with the A tile double-buffered in LDS.  These GEMMs are bound by streaming A in and the result out.
constexpr int kWsrStage = 2;

template <int OUT, int UW>  // UW: units (pairs of n-tiles for A-packed outputs, single n-tiles for the mask) per wave
__global__ __launch_bounds__(512, 2) void gemm_wsr_kernel(GemmArgs g) {
    typedef PBF16 P;
    typedef P::frag_t frag_t;
    constexpr int NB = P::NBH;  // both the feature tile (257 -> 288) and a hidden tile (271 -> 288) are 9 k-blocks
    constexpr bool kApack = (OUT == kOutAPlain || OUT == kOutASigmoid);
    constexpr bool kSigmoid = (OUT == kOutMask || OUT == kOutASigmoid);
    constexpr int NU = kApack ? P::NPB : 1;
    __shared__ __attribute__((aligned(16))) char smem[2 * kWsrStage * NB * 1024 + 8 * 1024];
    frag_t *abuf = (frag_t *) smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char *scratch = smem + 2 * kWsrStage * NB * 1024 + wave * 1024;  // per-wave transposer for A-packed outputs
    const int colq = lane & 15;
    const int units = g.ntiles / NU;
    const frag_t *w = (const frag_t *) g.w;

    int unit[UW];
    bool live[UW];
    frag_t wr[UW * NU][NB];
    float bias[UW * NU];
#pragma unroll
    for (int q = 0; q < UW; ++q) {
        live[q] = wave + 8 * q < units;
        unit[q] = live[q] ? wave + 8 * q : 0;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const int nt = unit[q] * NU + j;
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) wr[q * NU + j][blk] = w[((size_t) nt * NB + blk) * 64 + lane];
            bias[q * NU + j] = g.bias[nt * 16 + colq];
        }
    }

    constexpr int kStageBlocks = kWsrStage * NB;
    constexpr int kFetch = (kStageBlocks + 7) / 8;
    const int mstride = gridDim.x * kWsrStage;
    const frag_t *a1p = (const frag_t *) g.a1;
    int mt0 = blockIdx.x * kWsrStage;
    for (int i = wave; i < kStageBlocks; i += 8)
        if (mt0 + i / NB < g.mtiles) abuf[i * 64 + lane] = a1p[((size_t) mt0 * NB + i) * 64 + lane];
    __syncthreads();

    int cur = 0;
    for (; mt0 < g.mtiles; mt0 += mstride) {
        const int mn = mt0 + mstride;
        frag_t stage[kFetch];
#pragma unroll
        for (int i = 0; i < kFetch; ++i) {
            const int idx = wave + 8 * i;
            if (idx < kStageBlocks && mn + idx / NB < g.mtiles) stage[i] = a1p[((size_t) mn * NB + idx) * 64 + lane];
        }
        if (live[0]) {
#pragma unroll
            for (int m = 0; m < kWsrStage; ++m) {
                const int mt = mt0 + m;
                if (mt < g.mtiles) {
                    const frag_t *ab = abuf + (cur * kStageBlocks + m * NB) * 64;
                    frag_t a[NB];
#pragma unroll
                    for (int blk = 0; blk < NB; ++blk) a[blk] = ab[blk * 64 + lane];
#pragma unroll
                    for (int q = 0; q < UW; ++q) {
                        if (live[q]) {
                            f32x4 acc[NU];
#pragma unroll
                            for (int j = 0; j < NU; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int blk = 0; blk < NB; ++blk)
#pragma unroll
                                for (int j = 0; j < NU; ++j) acc[j] = P::mma(a[blk], wr[q * NU + j][blk], acc[j]);
#pragma unroll
                            for (int j = 0; j < NU; ++j) {
                                const int nt = unit[q] * NU + j;
                                f32x4 v = acc[j];
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    float x = v[i] + bias[q * NU + j];
                                    if (kSigmoid) x = kns_sigmoid(x);
                                    if (kApack && nt * 16 + colq >= g.n_valid) x = 0.0f;
                                    v[i] = x;
                                }
                                if (!kApack) {
                                    ((f32x4 *) g.out)[((size_t) mt * g.ntiles + nt) * 64 + lane] = v;
                                } else {
                                    uint16_t *sc = (uint16_t *) scratch;
#pragma unroll
                                    for (int i = 0; i < 4; ++i)
                                        sc[P::off((lane >> 4) * 4 + i, j * 16 + colq)] = P::cvt(v[i]);
                                }
                            }
                            if (kApack) {
                                wave_lds_sync();
                                ((uint4 *) g.out)[((size_t) mt * units + unit[q]) * 64 + lane] = ((const uint4 *) scratch)[lane];
                                wave_lds_sync();
                            }
                        }
                    }
                }
            }
        }
        frag_t *an = abuf + (cur ^ 1) * kStageBlocks * 64;
#pragma unroll
        for (int i = 0; i < kFetch; ++i) {
            const int idx = wave + 8 * i;
            if (idx < kStageBlocks && mn + idx / NB < g.mtiles) an[idx * 64 + lane] = stage[i];
        }
        __syncthreads();
        cur ^= 1;
    }
}

// ---- narrow heads (271 -> 1, 5, 40; bf16): the whole weight image is only 2-4 n-tiles, so every wave keeps ALL of it
// in registers and the waves split the m-tiles instead of the n-tiles: no LDS staging, no barrier, A fragments
// double-buffered in registers straight from HBM.  (With the n-tiles split over waves one or two waves did all the
// sigmoids of a workgroup and the kernel was VALU-bound at 36 us; this form is bound by reading A.)
template <int NT>  // n-tiles (2 or 4)
__global__ __launch_bounds__(512, 2) void gemm_head_kernel(GemmArgs g) {
    typedef PBF16 P;
    typedef P::frag_t frag_t;
    constexpr int NB = P::NBH;
    __shared__ __attribute__((aligned(16))) char smem[8 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint16_t *sc = (uint16_t *) (smem + wave * 1024);  // this wave's transposer: C-fragments -> one A-packed block
    const int colq = lane & 15;
    const frag_t *w = (const frag_t *) g.w;
    frag_t wr[NT][NB];
    float bias[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) wr[j][blk] = w[((size_t) j * NB + blk) * 64 + lane];
        bias[j] = g.bias[j * 16 + colq];
    }
    const frag_t *a1p = (const frag_t *) g.a1;
    const int stride = gridDim.x * 8;
    int mt = blockIdx.x * 8 + wave;
    frag_t an[NB];
    if (mt < g.mtiles) {
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) an[blk] = a1p[((size_t) mt * NB + blk) * 64 + lane];
    }
    for (; mt < g.mtiles; mt += stride) {
        frag_t a[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) a[blk] = an[blk];
        if (mt + stride < g.mtiles) {
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) an[blk] = a1p[((size_t) (mt + stride) * NB + blk) * 64 + lane];
        }
#pragma unroll
        for (int pair = 0; pair < NT / 2; ++pair) {
            f32x4 acc[2];
            acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int blk = 0; blk < NB; ++blk)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = P::mma(a[blk], wr[pair * 2 + j][blk], acc[j]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nt = pair * 2 + j;
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (nt * 16 < g.n_valid) {  // an n-tile made of padding columns only needs no sigmoid
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float x = kns_sigmoid(acc[j][i] + bias[nt]);
                        v[i] = nt * 16 + colq < g.n_valid ? x : 0.0f;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) sc[P::off((lane >> 4) * 4 + i, j * 16 + colq)] = P::cvt(v[i]);
            }
            wave_lds_sync();
            ((uint4 *) g.out)[((size_t) mt * (NT / 2) + pair) * 64 + lane] = ((const uint4 *) sc)[lane];
            wave_lds_sync();
        }
    }
}

template <class P>
static void launch_gemm_p(const GemmArgs &a, hipStream_t s) {
    const int nb = a.nb0 + a.nb1;
    const int gx = ceil_div(a.mtiles, kGemmMT);
    const bool apack = a.out_kind == kOutAPlain || a.out_kind == kOutASigmoid;
    const int units = a.ntiles / (apack ? P::NPB : 1);
    // few stream-frames: split the n-tiles over more workgroups so the weight stream is spread over the CUs
    int gy = 1;
    if (gx < 256) gy = min(ceil_div(units, 4), max(1, 512 / gx));
    size_t lds = (size_t) kGemmMT * nb * 1024 + 4 * kGemmMT * 1024;
    dim3 grid(gx, gy);
    switch (a.out_kind) {
        case kOutGi: hipLaunchKernelGGL((gemm_kernel<P, kOutGi>), grid, dim3(256), lds, s, a); break;
        case kOutMask: hipLaunchKernelGGL((gemm_kernel<P, kOutMask>), grid, dim3(256), lds, s, a); break;
        case kOutAPlain: hipLaunchKernelGGL((gemm_kernel<P, kOutAPlain>), grid, dim3(256), lds, s, a); break;
        default: hipLaunchKernelGGL((gemm_kernel<P, kOutASigmoid>), grid, dim3(256), lds, s, a); break;
    }
}

void launch_gemm(const GemmArgs &a, hipStream_t s) {
    static const bool no_ws = getenv("KOALA_AMD_GEMM_GENERIC") != nullptr;  // A/B switch for profiling
    if (a.precision == kBf16 && a.out_kind == kOutGi && a.ntiles == kGateTiles && a.nb1 == PBF16::NBH && a.nb0 <= 2 &&
        a.mtiles >= 256 && !no_ws) {
        static const bool ws1 = getenv("KOALA_AMD_GEMM_WS1") != nullptr;  // A/B switch: first weight-stationary form
        const dim3 grid(256), block(64 * kWs2Waves);
        if (ws1 || a.mtiles % 256 != 0) {
            if (a.nb0 == 0)
                hipLaunchKernelGGL(gemm_ws_kernel<0>, dim3(256), dim3(256), 0, s, a);
            else if (a.nb0 == 1)
                hipLaunchKernelGGL(gemm_ws_kernel<1>, dim3(256), dim3(256), 0, s, a);
            else
                hipLaunchKernelGGL(gemm_ws_kernel<2>, dim3(256), dim3(256), 0, s, a);
        } else if (a.mtiles % 512 == 0 && a.nb0 < 2) {  // four m-tiles per barrier (NB0 = 2 would spill)
            if (a.nb0 == 0)
                hipLaunchKernelGGL((gemm_ws2_kernel<0, 4>), grid, block, 0, s, a);
            else
                hipLaunchKernelGGL((gemm_ws2_kernel<1, 4>), grid, block, 0, s, a);
        } else {
            if (a.nb0 == 0)
                hipLaunchKernelGGL((gemm_ws2_kernel<0, 2>), grid, block, 0, s, a);
            else if (a.nb0 == 1)
                hipLaunchKernelGGL((gemm_ws2_kernel<1, 2>), grid, block, 0, s, a);
            else
                hipLaunchKernelGGL((gemm_ws2_kernel<2, 2>), grid, block, 0, s, a);
        }
        return;
    }
    static const bool no_wsr = getenv("KOALA_AMD_GEMM_NO_WSR") != nullptr;  // A/B switch
    if (a.precision == kBf16 && a.nb0 == 0 && a.nb1 == PBF16::NBH && a.mtiles >= 512 && !no_wsr) {
        const dim3 grid(256), block(512);
        if (a.out_kind == kOutAPlain && a.ntiles <= 32) {
            hipLaunchKernelGGL((gemm_wsr_kernel<kOutAPlain, 2>), grid, block, 0, s, a);
            return;
        }
        if (a.out_kind == kOutASigmoid && a.ntiles == 2) {
            hipLaunchKernelGGL(gemm_head_kernel<2>, grid, block, 0, s, a);
            return;
        }
        if (a.out_kind == kOutASigmoid && a.ntiles == 4) {
            hipLaunchKernelGGL(gemm_head_kernel<4>, grid, block, 0, s, a);
            return;
        }
        if (a.out_kind == kOutASigmoid && a.ntiles <= 16) {
            hipLaunchKernelGGL((gemm_wsr_kernel<kOutASigmoid, 1>), grid, block, 0, s, a);
            return;
        }
        if (a.out_kind == kOutMask && a.ntiles <= 24) {
            hipLaunchKernelGGL((gemm_wsr_kernel<kOutMask, 3>), grid, block, 0, s, a);
            return;
        }
    }
    if (a.precision == kBf16)
        launch_gemm_p<PBF16>(a, s);
    else
        launch_gemm_p<PF32>(a, s);
}

// ------------------------------------------------------------------------------------------------ recurrent GRU

// Weights streamed from L2 every step (fp32 parity path; bf16 only as an A/B switch).  W waves per workgroup share the
// 17 unit tiles round-robin; the B fragments of a tile are fetched two k-blocks (six fragments) ahead of their MFMAs, and with W = 8 two waves per SIMD cover each other's latencies.
template <class P, int W>
__global__ __launch_bounds__(64 * W, W / 4) void gru_kernel(GruArgs g) {
    typedef typename P::frag_t frag_t;
    typedef typename P::elem_t elem_t;
    constexpr int NBH = P::NBH;
    constexpr int TPW = (kUnitTiles + W - 1) / W;  // unit tiles per wave
    __shared__ __attribute__((aligned(16))) char hbuf[2][NBH * 1024];  // operand-typed hidden state, A-packed

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x;
    const int colq = lane & 15, rowq = (lane >> 4) * 4;

    // fp32 hidden state of the (row, unit) elements this lane owns
    f32x4 hreg[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int u = wave + W * q;
        hreg[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (u < kUnitTiles) hreg[q] = ((const f32x4 *) g.hstate_in)[((size_t) mt * kUnitTiles + u) * 64 + lane];
    }
    for (int i = tid; i < 2 * NBH * 64; i += 64 * W) ((uint4 *) hbuf)[i] = uint4{0, 0, 0, 0};
    __syncthreads();
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int u = wave + W * q;
        if (u < kUnitTiles) {
            const int k = u * 16 + colq;
            elem_t *dst = (elem_t *) hbuf[0] + (k / P::KB) * 64 * P::EPL;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[P::off(rowq + i, k % P::KB)] = P::cvt(hreg[q][i]);
        }
    }
    __syncthreads();

    const frag_t *whh = (const frag_t *) g.whh;
    int cur = 0;
    for (int t = 0; t < g.T; ++t) {
        const frag_t *ha = (const frag_t *) hbuf[cur];
        if (t > 0) {  // what is in LDS now is h_{t-1}: publish it as the next layer's A operand
            frag_t *hs = (frag_t *) g.hseq + ((size_t) (t - 1) * g.mtiles + mt) * NBH * 64;
            for (int blk = wave; blk < NBH; blk += W) hs[blk * 64 + lane] = ha[blk * 64 + lane];
        }
        const typename P::gi_t *gi = (const typename P::gi_t *) g.gi + ((size_t) t * g.mtiles + mt) * kGateTiles * 64;
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int u = wave + W * q;
            if (u < kUnitTiles) {
                typename P::gi_t gir = gi[(u * 3 + 0) * 64 + lane];
                typename P::gi_t giz = gi[(u * 3 + 1) * 64 + lane];
                typename P::gi_t gin = gi[(u * 3 + 2) * 64 + lane];
                f32x4 acc[3];
#pragma unroll
                for (int gt = 0; gt < 3; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
                // B fragments two k-blocks ahead of their MFMAs, rotated through registers (a rolled loop: unrolling all 51
                // fp32 k-block/gate pairs makes hipcc materialise an address pair per load and spill)
                const frag_t *wu = whh + (size_t) u * 3 * NBH * 64 + lane;
                frag_t b0[3], b1[3], b2[3];
#pragma unroll
                for (int gt = 0; gt < 3; ++gt) {
                    b0[gt] = wu[(gt * NBH + 0) * 64];
                    b1[gt] = wu[(gt * NBH + 1) * 64];
                    b2[gt] = wu[(gt * NBH + 2) * 64];
                }
#pragma nounroll
                for (int blk = 0; blk < NBH; ++blk) {
                    const frag_t ab = ha[blk * 64 + lane];
                    frag_t bc[3];
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) {
                        bc[gt] = b0[gt];
                        b0[gt] = b1[gt];
                        b1[gt] = b2[gt];
                    }
                    const int nb = blk + 3 < NBH ? blk + 3 : NBH - 1;
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) b2[gt] = wu[(gt * NBH + nb) * 64];
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) acc[gt] = P::mma(ab, bc[gt], acc[gt]);
                }
                const float br = g.bhh[(u * 3 + 0) * 16 + colq];
                const float bz = g.bhh[(u * 3 + 1) * 16 + colq];
                const float bn = g.bhh[(u * 3 + 2) * 16 + colq];
                f32x4 ir = P::from_gi(gir), iz = P::from_gi(giz), in = P::from_gi(gin);
                const int k = u * 16 + colq;
                elem_t *dst = (elem_t *) hbuf[cur ^ 1] + (k / P::KB) * 64 * P::EPL;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float r = kns_sigmoid(ir[i] + (acc[0][i] + br));
                    float z = kns_sigmoid(iz[i] + (acc[1][i] + bz));
                    float n = kns_tanh(__builtin_fmaf(r, acc[2][i] + bn, in[i]));
                    float h = __builtin_fmaf(z, hreg[q][i] - n, n);
                    hreg[q][i] = h;
                    dst[P::off(rowq + i, k % P::KB)] = P::cvt(h);
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    {
        frag_t *hs = (frag_t *) g.hseq + ((size_t) (g.T - 1) * g.mtiles + mt) * NBH * 64;
        for (int blk = wave; blk < NBH; blk += W) hs[blk * 64 + lane] = ((const frag_t *) hbuf[cur])[blk * 64 + lane];
    }
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int u = wave + W * q;
        if (u < kUnitTiles) ((f32x4 *) g.hstate_out)[((size_t) mt * kUnitTiles + u) * 64 + lane] = hreg[q];
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 fast_sigmoid2(f32x2 x) {
    f32x2 e = x * f32x2{-1.44269504088896341f, -1.44269504088896341f};
    e = f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + f32x2{1.0f, 1.0f};
    return f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
}
__device__ __forceinline__ f32x2 fast_tanh2(f32x2 x) {
    f32x2 e = x * f32x2{2.88539008177792681f, 2.88539008177792681f};
    e = f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + f32x2{1.0f, 1.0f};
    f32x2 r = f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    return f32x2{1.0f, 1.0f} - (r + r);
}

__global__ __launch_bounds__(256, 1) void gru_resident_kernel(GruArgs g) {
    typedef PBF16 P;
    typedef P::frag_t frag_t;
    constexpr int NBH = P::NBH;
    __shared__ __attribute__((aligned(16))) char smem[kResLds];
    char *hbuf0 = smem, *hbuf1 = smem + NBH * 1024;
    char *wl = smem + 2 * NBH * 1024;
    float *lbias = (float *) (smem + 2 * NBH * 1024 + 5 * kResTileBytes);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x;
    const int colq = lane & 15, rowq = (lane >> 4) * 4;
    const frag_t *whh = (const frag_t *) g.whh;

    // ---- prologue: this wave's 5 (4) unit tiles of W_hh -> VGPRs (tile 0), AGPRs (tiles 1, 2), LDS (tiles 3, 4)
    frag_t wv[3][NBH];
    frag_t wa[2][3][NBH];
#pragma unroll
    for (int gt = 0; gt < 3; ++gt)
#pragma unroll
        for (int blk = 0; blk < NBH; ++blk) wv[gt][blk] = whh[((size_t) (wave * 3 + gt) * NBH + blk) * 64 + lane];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int gt = 0; gt < 3; ++gt)
#pragma unroll
            for (int blk = 0; blk < NBH; ++blk) {
                wa[q][gt][blk] = pin_to_agpr(whh[((size_t) ((wave + 4 * (q + 1)) * 3 + gt) * NBH + blk) * 64 + lane]);
            }
    frag_t *wl3 = (frag_t *) (wl + wave * kResTileBytes);  // unit tile wave + 12, private to this wave
    frag_t *wl4 = (frag_t *) (wl + 4 * kResTileBytes);     // unit tile 16, wave 0 only
    for (int gt = 0; gt < 3; ++gt)
        for (int blk = 0; blk < NBH; ++blk) {
            wl3[(blk * 3 + gt) * 64 + lane] = whh[((size_t) ((wave + 12) * 3 + gt) * NBH + blk) * 64 + lane];
            if (wave == 0) wl4[(blk * 3 + gt) * 64 + lane] = whh[((size_t) (16 * 3 + gt) * NBH + blk) * 64 + lane];
        }
    for (int i = tid; i < kGateTiles * 16; i += 256) lbias[i] = g.bhh[i];

    f32x4 hreg[kGruTilesPerWave];
#pragma unroll
    for (int q = 0; q < kGruTilesPerWave; ++q) {
        const int u = wave + 4 * q;
        hreg[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (u < kUnitTiles) hreg[q] = ((const f32x4 *) g.hstate_in)[((size_t) mt * kUnitTiles + u) * 64 + lane];
    }
    for (int i = tid; i < 2 * NBH * 64; i += 256) ((uint4 *) smem)[i] = uint4{0, 0, 0, 0};
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kGruTilesPerWave; ++q) {
        const int u = wave + 4 * q;
        if (u < kUnitTiles) {
            const int k = u * 16 + colq;
            uint16_t *dst = (uint16_t *) hbuf0 + (k / P::KB) * 64 * P::EPL;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[P::off(rowq + i, k % P::KB)] = P::cvt(hreg[q][i]);
        }
    }
    // pre-activations of step 0
    P::gi_t gi[kGruTilesPerWave][3];
    {
        const P::gi_t *gp = (const P::gi_t *) g.gi + (size_t) mt * kGateTiles * 64;
#pragma unroll
        for (int q = 0; q < kGruTilesPerWave; ++q)
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) {
                const int u = wave + 4 * q;
                if (u < kUnitTiles) gi[q][gt] = gp[(u * 3 + gt) * 64 + lane];
            }
    }
    __syncthreads();

    for (int t = 0; t < g.T; ++t) {
        KNS_STAMP(0);
        const char *hc = (t & 1) ? hbuf1 : hbuf0;
        char *hn = (t & 1) ? hbuf0 : hbuf1;
        frag_t a[NBH];
#pragma unroll
        for (int blk = 0; blk < NBH; ++blk) a[blk] = ((const frag_t *) hc)[blk * 64 + lane];
        if (t > 0) {  // LDS holds h_{t-1}: publish it as the next layer's A operand
            frag_t *hs = (frag_t *) g.hseq + ((size_t) (t - 1) * g.mtiles + mt) * NBH * 64;
#pragma unroll
            for (int blk = 0; blk < NBH; ++blk)
                if ((blk & 3) == wave) hs[blk * 64 + lane] = a[blk];
        }
        KNS_STAMP(1);
        const P::gi_t *gnext =
            (const P::gi_t *) g.gi + ((size_t) (t + 1 < g.T ? t + 1 : t) * g.mtiles + mt) * kGateTiles * 64;
#pragma unroll
        for (int q = 0; q < kGruTilesPerWave; ++q) {
            const int u = wave + 4 * q;
            if (q < 4 || wave == 0) {
                f32x4 acc[3];
#pragma unroll
                for (int gt = 0; gt < 3; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (q < 3) {
#pragma unroll
                    for (int blk = 0; blk < NBH; ++blk)
#pragma unroll
                        for (int gt = 0; gt < 3; ++gt)
                            acc[gt] = P::mma(a[blk], q == 0 ? wv[gt][blk] : wa[q == 2 ? 1 : 0][gt][blk], acc[gt]);
                } else {
                    mma_lds_tile<NBH, 6>(acc, a, q == 3 ? wl3 : wl4, lane);
                }
                f32x4 ir = P::from_gi(gi[q][0]), iz = P::from_gi(gi[q][1]), in = P::from_gi(gi[q][2]);
                // this tile's pre-activations of the next step: in flight while the other tiles compute
#pragma unroll
                for (int gt = 0; gt < 3; ++gt) gi[q][gt] = gnext[(u * 3 + gt) * 64 + lane];
                const float br = lbias[(u * 3 + 0) * 16 + colq], bz = lbias[(u * 3 + 1) * 16 + colq],
                            bn = lbias[(u * 3 + 2) * 16 + colq];
                const int k = u * 16 + colq;
                uint16_t *dst = (uint16_t *) hn + (k / P::KB) * 64 * P::EPL + P::off(rowq, k % P::KB);
                const f32x2 vbr = {br, br}, vbz = {bz, bz}, vbn = {bn, bn};
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const f32x2 ar = {acc[0][2 * p], acc[0][2 * p + 1]}, az = {acc[1][2 * p], acc[1][2 * p + 1]},
                                an = {acc[2][2 * p], acc[2][2 * p + 1]};
                    const f32x2 xr = {ir[2 * p], ir[2 * p + 1]}, xz = {iz[2 * p], iz[2 * p + 1]},
                                xn = {in[2 * p], in[2 * p + 1]};
                    const f32x2 r = fast_sigmoid2(xr + (ar + vbr));
                    const f32x2 z = fast_sigmoid2(xz + (az + vbz));
                    const f32x2 n = fast_tanh2(r * (an + vbn) + xn);
                    const f32x2 hp = {hreg[q][2 * p], hreg[q][2 * p + 1]};
                    const f32x2 h = z * (hp - n) + n;
                    hreg[q][2 * p] = h[0];
                    hreg[q][2 * p + 1] = h[1];
                    const uint32_t bits = __builtin_bit_cast(uint32_t, __builtin_convertvector(h, bf16x2));
                    dst[(2 * p) * 8] = (uint16_t) bits;  // consecutive rows sit 8 elements apart in an A-packed block
                    dst[(2 * p + 1) * 8] = (uint16_t) (bits >> 16);
                }
                KNS_STAMP(2 + q);
            }
        }
        KNS_STAMP(7);
        __syncthreads();
        KNS_STAMP(8);
    }
    {
        const char *hc = (g.T & 1) ? hbuf1 : hbuf0;
        frag_t *hs = (frag_t *) g.hseq + ((size_t) (g.T - 1) * g.mtiles + mt) * NBH * 64;
        for (int blk = wave; blk < NBH; blk += 4) hs[blk * 64 + lane] = ((const frag_t *) hc)[blk * 64 + lane];
    }
#pragma unroll
    for (int q = 0; q < kGruTilesPerWave; ++q) {
        const int u = wave + 4 * q;
        if (u < kUnitTiles) ((f32x4 *) g.hstate_out)[((size_t) mt * kUnitTiles + u) * 64 + lane] = hreg[q];
    }
}

// ---- low-latency GRU layer: input GEMM + recurrent GEMM + gates of one frame, one wavefront per (unit tile, m-tile).
// Arithmetic is, operation for operation, what the chunked path does (same MFMA, same k order, gi rounded to its storage
// type before the gates, same gate formulas per precision), so a stream's samples do not depend on which path ran.
template <class P>
__global__ __launch_bounds__(64) void gru_small_kernel(GruSmallArgs g) {
    typedef typename P::frag_t frag_t;
    typedef typename P::elem_t elem_t;
    constexpr int NBH = P::NBH;
    __shared__ __attribute__((aligned(16))) char hbuf[NBH * 1024];
    const int lane = threadIdx.x;
    const int u = blockIdx.x, mt = blockIdx.y;
    const int colq = lane & 15, rowq = (lane >> 4) * 4;
    const int nb = g.nb0 + NBH;

    // h_{t-1}: fp32 C-fragments -> operand-typed A-fragments through LDS
    for (int i = lane; i < NBH * 64; i += 64) ((uint4 *) hbuf)[i] = uint4{0, 0, 0, 0};
    wave_lds_sync();
    f32x4 hown = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int v = 0; v < kUnitTiles; ++v) {
        const f32x4 hv = ((const f32x4 *) g.hstate_in)[((size_t) mt * kUnitTiles + v) * 64 + lane];
        if (v == u) hown = hv;
        const int k = v * 16 + colq;
        elem_t *dst = (elem_t *) hbuf + (k / P::KB) * 64 * P::EPL;
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[P::off(rowq + i, k % P::KB)] = P::cvt(hv[i]);
    }
    wave_lds_sync();

    const frag_t *wih = (const frag_t *) g.wih, *whh = (const frag_t *) g.whh;
    f32x4 acci[3], acch[3];
#pragma unroll
    for (int gt = 0; gt < 3; ++gt) acci[gt] = acch[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int blk = 0; blk < nb; ++blk) {
        const frag_t a = blk < g.nb0 ? ((const frag_t *) g.a0)[((size_t) mt * g.nb0 + blk) * 64 + lane]
                                     : ((const frag_t *) g.a1)[((size_t) mt * NBH + (blk - g.nb0)) * 64 + lane];
#pragma unroll
        for (int gt = 0; gt < 3; ++gt)
            acci[gt] = P::mma(a, wih[((size_t) (u * 3 + gt) * nb + blk) * 64 + lane], acci[gt]);
    }
#pragma unroll
    for (int blk = 0; blk < NBH; ++blk) {
        const frag_t a = ((const frag_t *) hbuf)[blk * 64 + lane];
#pragma unroll
        for (int gt = 0; gt < 3; ++gt)
            acch[gt] = P::mma(a, whh[((size_t) (u * 3 + gt) * NBH + blk) * 64 + lane], acch[gt]);
    }
    f32x4 gin[3];
#pragma unroll
    for (int gt = 0; gt < 3; ++gt) {
        const float b = g.bih[(u * 3 + gt) * 16 + colq];
        f32x4 v = acci[gt];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] + b;
        gin[gt] = P::from_gi(P::to_gi(v));
    }
    const float br = g.bhh[(u * 3 + 0) * 16 + colq], bz = g.bhh[(u * 3 + 1) * 16 + colq],
                bn = g.bhh[(u * 3 + 2) * 16 + colq];
    f32x4 hnew;
    if (P::kPrec == kBf16) {
        const f32x2 vbr = {br, br}, vbz = {bz, bz}, vbn = {bn, bn};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const f32x2 ar = {acch[0][2 * p], acch[0][2 * p + 1]}, az = {acch[1][2 * p], acch[1][2 * p + 1]},
                        an = {acch[2][2 * p], acch[2][2 * p + 1]};
            const f32x2 xr = {gin[0][2 * p], gin[0][2 * p + 1]}, xz = {gin[1][2 * p], gin[1][2 * p + 1]},
                        xn = {gin[2][2 * p], gin[2][2 * p + 1]};
            const f32x2 r = fast_sigmoid2(xr + (ar + vbr));
            const f32x2 z = fast_sigmoid2(xz + (az + vbz));
            const f32x2 n = fast_tanh2(r * (an + vbn) + xn);
            const f32x2 hp = {hown[2 * p], hown[2 * p + 1]};
            const f32x2 h = z * (hp - n) + n;
            hnew[2 * p] = h[0];
            hnew[2 * p + 1] = h[1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float r = kns_sigmoid(gin[0][i] + (acch[0][i] + br));
            float z = kns_sigmoid(gin[1][i] + (acch[1][i] + bz));
            float n = kns_tanh(__builtin_fmaf(r, acch[2][i] + bn, gin[2][i]));
            hnew[i] = __builtin_fmaf(z, hown[i] - n, n);
        }
    }
    ((f32x4 *) g.hstate_out)[((size_t) mt * kUnitTiles + u) * 64 + lane] = hnew;
    const int k = u * 16 + colq;
    elem_t *hs = (elem_t *) g.hseq + ((size_t) mt * NBH + k / P::KB) * 64 * P::EPL;
#pragma unroll
    for (int i = 0; i < 4; ++i) hs[P::off(rowq + i, k % P::KB)] = P::cvt(hnew[i]);
}

void launch_gru_small(const GruSmallArgs &a, hipStream_t s) {
    dim3 grid(kUnitTiles, a.mtiles);
    if (a.precision == kBf16)
        hipLaunchKernelGGL(gru_small_kernel<PBF16>, grid, dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL(gru_small_kernel<PF32>, grid, dim3(64), 0, s, a);
}

// ---- 8-wave form of the resident recurrent kernel.  Measured on MI355X: a single wave per SIMD executes its MFMAs and
// its gate VALU one after the other (a unit tile costs 432 + ~650 cycles), while two waves on one SIMD overlap them.
// Eight waves of <= 256 registers hold the 459 KiB of W_hh as: wave w owns unit tiles w and w + 8 (wave 0 also tile 16);
// tile w entirely in VGPRs (27 fragments), the first 14 fragments of tile w + 8 in VGPRs and its last 13 in LDS, tile 16
// in LDS: 328 KiB of registers + 131 KiB of LDS.  A fragments are re-read from LDS per k-block.
constexpr int kR8Waves = 8;
constexpr int kR8RegFrags1 = 14;                      // fragments of the second tile kept in registers
constexpr int kR8LdsFrags1 = 27 - kR8RegFrags1;       // ... and in LDS
constexpr int kR8Lds = 2 * PBF16::NBH * 1024 + kR8Waves * kR8LdsFrags1 * 1024 + 27 * 1024 + kResBiasBytes;

// MFMAs of one unit tile whose fragments [first_lds, 27) live in LDS at wl[(i - first_lds)] (i = k_block * 3 + gate) and
// the rest in wreg[i]; LDS fragments go through a kQ-deep register queue
template <int kFirstLds, int kQ, int kNReg>
__device__ __forceinline__ void r8_tile_mma(f32x4 (&acc)[3], const bf16x8 *ha, const bf16x8 (&wreg)[kNReg], const bf16x8 *wl,
                                            int lane) {
    constexpr int N = 27;
    bf16x8 qb[kQ];
#pragma unroll
    for (int p = 0; p < kQ; ++p)
        if (kFirstLds + p < N) qb[p] = wl[p * 64 + lane];
    bf16x8 a = ha[lane];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (i % 3 == 0 && i > 0) a = ha[(i / 3) * 64 + lane];
        bf16x8 b;
        if (i < kFirstLds) {
            b = wreg[i < kNReg ? i : 0];
        } else {
            const int j = i - kFirstLds;
            b = qb[j % kQ];
            if (i + kQ < N) qb[j % kQ] = wl[(j + kQ) * 64 + lane];
        }
        acc[i % 3] = PBF16::mma(a, b, acc[i % 3]);
    }
}

__global__ __launch_bounds__(64 * kR8Waves, 2) void gru_resident8_kernel(GruArgs g) {
    typedef PBF16 P;
    typedef P::frag_t frag_t;
    constexpr int NBH = P::NBH;
    __shared__ __attribute__((aligned(16))) char smem[kR8Lds + 3 * 1024];
    char *hbuf0 = smem, *hbuf1 = smem + NBH * 1024;
    frag_t *wl1 = (frag_t *) (smem + 2 * NBH * 1024);                                   // [8 waves][13][64]
    frag_t *wl16 = (frag_t *) (smem + 2 * NBH * 1024 + kR8Waves * kR8LdsFrags1 * 1024);  // [27][64], i = blk * 3 + gate
    float *lbias = (float *) (smem + 2 * NBH * 1024 + kR8Waves * kR8LdsFrags1 * 1024 + 27 * 1024);
    f32x4 *acc16 = (f32x4 *) (smem + kR8Lds);  // [3 gates][64 lanes]: unit tile 16's accumulators, handed across waves

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x;
    const int colq = lane & 15, rowq = (lane >> 4) * 4;
    const frag_t *whh = (const frag_t *) g.whh;
    const int u0 = wave, u1 = wave + 8, u2 = 16;
    // Unit tile 16 (the 17th) would make one wave's serial chain 3 tiles long while the others wait at the barrier.
    // Its 27 MFMAs are done by waves 5, 6, 7 (one gate each, the full k chain in one accumulator, so the arithmetic is
    // unchanged), the accumulators cross LDS, and after a barrier waves 0..3 each do the gate math of one of the four
    // rows a lane owns.
    const int g16 = wave - 5;         // gate whose tile-16 MFMAs this wave computes (waves 5..7)
    const bool q16 = wave < 4;        // this wave finishes row (lane >> 4) * 4 + wave of tile 16

    // ---- prologue
    frag_t w0[27], w1[kR8RegFrags1];
#pragma unroll
    for (int i = 0; i < 27; ++i) w0[i] = whh[((size_t) (u0 * 3 + i % 3) * NBH + i / 3) * 64 + lane];
#pragma unroll
    for (int i = 0; i < kR8RegFrags1; ++i) w1[i] = whh[((size_t) (u1 * 3 + i % 3) * NBH + i / 3) * 64 + lane];
    frag_t *wl1w = wl1 + wave * kR8LdsFrags1 * 64;
    for (int i = kR8RegFrags1; i < 27; ++i)
        wl1w[(i - kR8RegFrags1) * 64 + lane] = whh[((size_t) (u1 * 3 + i % 3) * NBH + i / 3) * 64 + lane];
    for (int i = wave; i < 27; i += kR8Waves) wl16[i * 64 + lane] = whh[((size_t) (u2 * 3 + i % 3) * NBH + i / 3) * 64 + lane];
    for (int i = tid; i < kGateTiles * 16; i += 64 * kR8Waves) lbias[i] = g.bhh[i];

    f32x4 hreg[2];
    hreg[0] = ((const f32x4 *) g.hstate_in)[((size_t) mt * kUnitTiles + u0) * 64 + lane];
    hreg[1] = ((const f32x4 *) g.hstate_in)[((size_t) mt * kUnitTiles + u1) * 64 + lane];
    const int e16 = q16 ? wave : 0;  // element of the f32x4 this wave owns in tile 16
    float h16 = g.hstate_in[(((size_t) mt * kUnitTiles + u2) * 64 + lane) * 4 + e16];
    for (int i = tid; i < 2 * NBH * 64; i += 64 * kR8Waves) ((uint4 *) smem)[i] = uint4{0, 0, 0, 0};
    __syncthreads();
    auto put_h = [&](char *buf, int u, const f32x4 &h) {
        const int k = u * 16 + colq;
        uint16_t *dst = (uint16_t *) buf + (k / P::KB) * 64 * P::EPL + P::off(rowq, k % P::KB);
        const uint32_t lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{h[0], h[1]}, bf16x2));
        const uint32_t hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{h[2], h[3]}, bf16x2));
        dst[0] = (uint16_t) lo;  // consecutive rows sit 8 elements apart in an A-packed block
        dst[8] = (uint16_t) (lo >> 16);
        dst[16] = (uint16_t) hi;
        dst[24] = (uint16_t) (hi >> 16);
    };
    auto put_h16 = [&](char *buf, float h) {  // one row of tile 16
        const int k = u2 * 16 + colq;
        uint16_t *dst = (uint16_t *) buf + (k / P::KB) * 64 * P::EPL + P::off(rowq + e16, k % P::KB);
        dst[0] = f2bf(h);
    };
    put_h(hbuf0, u0, hreg[0]);
    put_h(hbuf0, u1, hreg[1]);
    if (q16) put_h16(hbuf0, h16);

    P::gi_t gi[2][3], gi16[3];
    {
        const P::gi_t *gp = (const P::gi_t *) g.gi + (size_t) mt * kGateTiles * 64;
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) {
            gi[0][gt] = gp[(u0 * 3 + gt) * 64 + lane];
            gi[1][gt] = gp[(u1 * 3 + gt) * 64 + lane];
            gi16[gt] = gp[(u2 * 3 + gt) * 64 + lane];
        }
    }
    __syncthreads();

    for (int t = 0; t < g.T; ++t) {
        KNS_STAMP(0);
        const char *hc = (t & 1) ? hbuf1 : hbuf0;
        char *hn = (t & 1) ? hbuf0 : hbuf1;
        const frag_t *ha = (const frag_t *) hc;
        if (t > 0) {  // LDS holds h_{t-1}: publish it as the next layer's A operand
            frag_t *hs = (frag_t *) g.hseq + ((size_t) (t - 1) * g.mtiles + mt) * NBH * 64;
            for (int blk = wave; blk < NBH; blk += kR8Waves) hs[blk * 64 + lane] = ha[blk * 64 + lane];
        }
        const P::gi_t *gnext =
            (const P::gi_t *) g.gi + ((size_t) (t + 1 < g.T ? t + 1 : t) * g.mtiles + mt) * kGateTiles * 64;

        auto gates = [&](const int q, const int u, f32x4 (&acc)[3]) {
            const f32x4 ir = P::from_gi(gi[q][0]), iz = P::from_gi(gi[q][1]), in = P::from_gi(gi[q][2]);
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) gi[q][gt] = gnext[(u * 3 + gt) * 64 + lane];
            const float br = lbias[(u * 3 + 0) * 16 + colq], bz = lbias[(u * 3 + 1) * 16 + colq],
                        bn = lbias[(u * 3 + 2) * 16 + colq];
            const f32x2 vbr = {br, br}, vbz = {bz, bz}, vbn = {bn, bn};
            f32x4 hnew;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const f32x2 ar = {acc[0][2 * p], acc[0][2 * p + 1]}, az = {acc[1][2 * p], acc[1][2 * p + 1]},
                            an = {acc[2][2 * p], acc[2][2 * p + 1]};
                const f32x2 xr = {ir[2 * p], ir[2 * p + 1]}, xz = {iz[2 * p], iz[2 * p + 1]}, xn = {in[2 * p], in[2 * p + 1]};
                const f32x2 r = fast_sigmoid2(xr + (ar + vbr));
                const f32x2 z = fast_sigmoid2(xz + (az + vbz));
                const f32x2 n = fast_tanh2(r * (an + vbn) + xn);
                const f32x2 hp = {hreg[q][2 * p], hreg[q][2 * p + 1]};
                const f32x2 h = z * (hp - n) + n;
                hnew[2 * p] = h[0];
                hnew[2 * p + 1] = h[1];
            }
            hreg[q] = hnew;
            put_h(hn, u, hnew);
        };

        KNS_STAMP(1);
        f32x4 acc[3];
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
        r8_tile_mma<27, 1, 27>(acc, ha, w0, wl16, lane);
        KNS_STAMP(2);
        gates(0, u0, acc);
        KNS_STAMP(3);
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
        r8_tile_mma<kR8RegFrags1, 3, kR8RegFrags1>(acc, ha, w1, wl1w, lane);
        KNS_STAMP(4);
        gates(1, u1, acc);
        KNS_STAMP(5);
        if (g16 >= 0) {  // waves 5, 6, 7: one gate of unit tile 16, k-blocks in order in one accumulator
            f32x4 a16 = f32x4{0.f, 0.f, 0.f, 0.f};
            frag_t qb[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) qb[p] = wl16[(p * 3 + g16) * 64 + lane];
#pragma unroll
            for (int blk = 0; blk < NBH; ++blk) {
                const frag_t a = ha[blk * 64 + lane];
                const frag_t b = qb[blk % 3];
                if (blk + 3 < NBH) qb[blk % 3] = wl16[((blk + 3) * 3 + g16) * 64 + lane];
                a16 = P::mma(a, b, a16);
            }
            acc16[g16 * 64 + lane] = a16;
        }
        KNS_STAMP(6);
        __syncthreads();
        if (q16) {  // waves 0..3: row e16 of every lane's four rows of unit tile 16
            const float ar = ((const float *) acc16)[(0 * 64 + lane) * 4 + e16];
            const float az = ((const float *) acc16)[(1 * 64 + lane) * 4 + e16];
            const float an = ((const float *) acc16)[(2 * 64 + lane) * 4 + e16];
            const float xr = (float) gi16[0][e16], xz = (float) gi16[1][e16], xn = (float) gi16[2][e16];
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) gi16[gt] = gnext[(u2 * 3 + gt) * 64 + lane];
            const float br = lbias[(u2 * 3 + 0) * 16 + colq], bz = lbias[(u2 * 3 + 1) * 16 + colq],
                        bn = lbias[(u2 * 3 + 2) * 16 + colq];
            // same operations, element by element, as the packed gate math above
            const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((xr + (ar + br)) * -1.44269504088896341f));
            const float z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((xz + (az + bz)) * -1.44269504088896341f));
            const float rr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((r * (an + bn) + xn) * 2.88539008177792681f));
            const float n = 1.0f - (rr + rr);
            h16 = z * (h16 - n) + n;
            put_h16(hn, h16);
        }
        KNS_STAMP(7);
        __syncthreads();
        KNS_STAMP(8);
    }
    {
        const char *hc = (g.T & 1) ? hbuf1 : hbuf0;
        frag_t *hs = (frag_t *) g.hseq + ((size_t) (g.T - 1) * g.mtiles + mt) * NBH * 64;
        for (int blk = wave; blk < NBH; blk += kR8Waves) hs[blk * 64 + lane] = ((const frag_t *) hc)[blk * 64 + lane];
    }
    ((f32x4 *) g.hstate_out)[((size_t) mt * kUnitTiles + u0) * 64 + lane] = hreg[0];
    ((f32x4 *) g.hstate_out)[((size_t) mt * kUnitTiles + u1) * 64 + lane] = hreg[1];
    if (q16) g.hstate_out[(((size_t) mt * kUnitTiles + u2) * 64 + lane) * 4 + e16] = h16;
}

void launch_gru(const GruArgs &a, hipStream_t s) {
    static const bool stream_weights = getenv("KOALA_AMD_GRU_STREAM") != nullptr;  // A/B switch for profiling
    static const bool four_waves = getenv("KOALA_AMD_GRU_4WAVE") != nullptr;  // A/B switch: one wave per SIMD
    if (a.precision == kBf16 && !stream_weights && !four_waves)
        hipLaunchKernelGGL(gru_resident8_kernel, dim3(a.mtiles), dim3(64 * kR8Waves), 0, s, a);
    else if (a.precision == kBf16 && !stream_weights)
        hipLaunchKernelGGL(gru_resident_kernel, dim3(a.mtiles), dim3(256), 0, s, a);
    else if (a.precision == kBf16)
        hipLaunchKernelGGL((gru_kernel<PBF16, 8>), dim3(a.mtiles), dim3(512), 0, s, a);
    else
        hipLaunchKernelGGL((gru_kernel<PF32, 8>), dim3(a.mtiles), dim3(512), 0, s, a);
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

#ifdef KNS_TIMING
void read_timing(unsigned long long *out) { (void) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kns_timing), sizeof(unsigned long long) * 64); }
#endif

void launch_reset(const ResetArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(reset_kernel, dim3(a.Bpad), dim3(256), 0, s, a);
}

}  // namespace kns
