// kns_device.hpp -- device-side building blocks shared by the kernel translation units: the KNS-v1 scalar math
// (DESIGN.md section 2), the two precision traits (MFMA instruction + fragment packing), the wave-level FFT-256, and
// the helpers of the weight-resident kernels.
// Hand-written gfx950 (CDNA4) kernels of the KNS-v1 noise suppressor: kns_stft.hip, kns_gemm.hip, kns_gru.hip.
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
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "kns_kernels.h"

namespace kns {

#ifdef KNS_TIMING
static __device__ unsigned long long g_kns_timing[8 * 16];  // [wave][stamp] of workgroup 0 at step 5
#define KNS_STAMP(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && t == 5) g_kns_timing[(threadIdx.x >> 6) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define KNS_STAMP_AT(i, step) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && t == (step)) g_kns_timing[(threadIdx.x >> 6) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
// (gru_wave_kernel: role r of the workgroup in slot 1 of XCD 3, second m-tile of its group, in the launch the engine marks)
#define KNS_WSTAMP(i) do { if (dbg_stamp && (threadIdx.x & 63) == 0 && mt == m0 + 1) g_kns_timing[role * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define KNS_WSTAMP(i) do { } while (0)
#define KNS_STAMP(i) do { } while (0)
#define KNS_STAMP_AT(i, step) do { } while (0)
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

// ln x of the bf16 configuration's FEATURES (round 5): x = m 2^e with m in [0.5, 1) (frexp), ln x = e ln 2 + p(m), p a degree-4
// polynomial (max error 7e-5: a feature moves by 9e-6, its bf16 rounding step is ~2e-3).  Built from IEEE operations only and written
// twice (oracle/kns_oracle.c, kns_log_fast), so the bf16 features are the SAME BITS on both sides -- with the hardware v_log_f32 a
// feature's rounding flipped now and then, and the default model turned such a flip (a 0.1-0.3 dB step of a band level) into its
// largest PCM differences (tools/model_sensitivity.py).  Eight full-rate instructions per bin against a quarter-rate v_log_f32 and a
// multiplication: +3 cycles per bin and lane group.
__device__ __forceinline__ float kns_log_fast(float x) {
    const float m = __builtin_amdgcn_frexp_mantf(x);
    const float e = (float) __builtin_amdgcn_frexp_expf(x);
    float p = -8.873490199e-01f;
    p = __builtin_fmaf(p, m, 3.524021909e+00f);
    p = __builtin_fmaf(p, m, -5.820779088e+00f);
    p = __builtin_fmaf(p, m, 5.613961063e+00f);
    p = __builtin_fmaf(p, m, -2.429906919e+00f);
    return __builtin_fmaf(e, 0.693147180559945309f, p);
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
    // two independent chains advanced by one k-block each, their MFMAs alternating (a chain's next MFMA needs its previous result)
    static __device__ __forceinline__ void mma2(frag_t a0, frag_t b0, f32x4 &c0, frag_t a1, frag_t b1, f32x4 &c1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], b0[q], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], b1[q], c1, 0, 0, 0);
        }
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
    static __device__ __forceinline__ void mma2(frag_t a0, frag_t b0, f32x4 &c0, frag_t a1, frag_t b1, f32x4 &c1) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, c1, 0, 0, 0);
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

// complex numbers
#ifdef KNS_CPX_PACKED
// as packed pairs: add/sub are one v_pk_add_f32, a complex multiply is v_pk_mul_f32 + v_pk_fma_f32
typedef float cpx __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return a + b; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return a - b; }
__device__ __forceinline__ cpx cmul(cpx a, cpx w) {
    const cpx t = cpx{a.y, a.y} * cpx{-w.y, w.x};
    return __builtin_elementwise_fma(cpx{a.x, a.x}, w, t);
}
#else
// as two scalars (the same operations as the packed form, one lane-operation each: on gfx950 a packed f32 instruction
// takes the issue time of two plain ones, and its operands must sit in aligned register pairs, which costs moves)
struct alignas(8) cpx {
    float x, y;
};
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return cpx{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return cpx{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cpx cmul(cpx a, cpx w) {
    return cpx{__builtin_fmaf(a.x, w.x, a.y * -w.y), __builtin_fmaf(a.x, w.y, a.y * w.x)};
}
#endif

__device__ __forceinline__ void radix4(cpx (&v)[4]) {
    cpx a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]), a2 = cadd(v[1], v[3]), d = csub(v[1], v[3]);
    cpx a3 = {d.y, -d.x};  // (v1 - v3) * (-i)
    v[0] = cadd(a0, a2);
    v[1] = cadd(a1, a3);
    v[2] = csub(a0, a2);
    v[3] = csub(a1, a3);
}

// ---- FFT-256 of FOUR frames per wavefront: a row of 16 lanes owns one frame, a lane 16 complex points.
// 256 = 16 x 16: a 16-point DFT in registers (two radix-4 levels), the W_256 twiddles, ONE 16 x 16 transpose among the
// row's lanes through LDS, a second 16-point DFT in registers.  Index maps:
//   input   v[j]  = z[16 j + c]                       (stride-16 points: the first DFT needs no exchange)
//   output  v[k2] = Z[c + 16 k2]                      (the same map: an FFT's output feeds the next FFT as it is)
// with c = fft_column(lane), the lane's column in its row.
// Against the one-frame-per-wave radix-4 Stockham form this replaces (four exchanges per frame, 4-way bank conflicts on
// two of them) a frame crosses LDS once, at addresses that are the same for every frame (a padded [16][17] tile per row:
// conflict-free both ways, all offsets immediate).
constexpr int kFftRowBytes = 16 * 17 * 8;            // one frame's exchange tile
constexpr int kFftWaveBytes = 4 * kFftRowBytes;      // per wave

// natural order in, natural order out; forward transform (W = exp(-2 pi i / 16))
__device__ __forceinline__ void dft16(cpx (&x)[16]) {
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, c2 = 0.70710678118654752f;
    cpx u[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        cpx q[4] = {x[b], x[4 + b], x[8 + b], x[12 + b]};
        radix4(q);
#pragma unroll
        for (int c = 0; c < 4; ++c) u[b][c] = q[c];
    }
    // u[b][c] *= W_16^(b c)
    u[1][1] = cmul(u[1][1], cpx{c1, -s1});
    u[1][2] = cpx{(u[1][2].x + u[1][2].y) * c2, (u[1][2].y - u[1][2].x) * c2};
    u[1][3] = cmul(u[1][3], cpx{s1, -c1});
    u[2][1] = cpx{(u[2][1].x + u[2][1].y) * c2, (u[2][1].y - u[2][1].x) * c2};
    u[2][2] = cpx{u[2][2].y, -u[2][2].x};
    u[2][3] = cpx{(u[2][3].y - u[2][3].x) * c2, -((u[2][3].x + u[2][3].y) * c2)};
    u[3][1] = cmul(u[3][1], cpx{s1, -c1});
    u[3][2] = cpx{(u[3][2].y - u[3][2].x) * c2, -((u[3][2].x + u[3][2].y) * c2)};
    u[3][3] = cmul(u[3][3], cpx{-c1, s1});
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        cpx q[4] = {u[0][c], u[1][c], u[2][c], u[3][c]};
        radix4(q);
#pragma unroll
        for (int d = 0; d < 4; ++d) x[c + 4 * d] = q[d];
    }
}

// Column of a row's lane: the 16 columns c = 0..15 sit on the lanes so that the two columns c and 16 - c, which hold each
// other's mirrored bins (k <-> 256 - k), are NEIGHBOURS: lanes (2, 3) = columns (1, 15), (4, 5) = (2, 14) ... (14, 15) =
// (7, 9), and lanes 0, 1 = the self-paired columns 0 and 8.  The mirrored-bin exchange is then one quad_perm DPP move
// per register (a wavefront shuffle, no LDS).
__device__ __forceinline__ int fft_column(int lane) {
    const int l = lane & 15;
    return (l & 1) ? (l == 1 ? 8 : 16 - (l >> 1)) : (l >> 1);
}

// twl: this lane's view of the LDS table [k1][column] of exp(-2 pi i column k1 / 256) (fft_fill_twiddles); xw / xr: this
// lane's write and read base inside the wave's exchange tiles (fft_lane_bases); all offsets below are immediates
__device__ __forceinline__ void fft256_rows(cpx (&v)[16], const char *twl, char *xw, const char *xr) {
    dft16(v);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) v[k1] = cmul(v[k1], *(const cpx *) (twl + k1 * 128));
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) *(cpx *) (xw + k1 * 136) = v[k1];  // element (k1, c) of the row's [16][17] tile
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = *(const cpx *) (xr + j * 8);   // element (c, j)
    wave_lds_sync();
    dft16(v);
}

constexpr int kFftTwiddleBytes = 16 * 16 * 8;

// tw512: exp(-2 pi i k / 512), k = 0..511 (anywhere); dst: LDS, kFftTwiddleBytes
__device__ __forceinline__ void fft_fill_twiddles(char *dst, const float2 *tw512, int tid, int nthreads) {
    for (int i = tid; i < 256; i += nthreads) ((float2 *) dst)[i] = tw512[2 * (((i & 15) * (i >> 4)) & 255)];
}

__device__ __forceinline__ void fft_lane_bases(char *wave_buf, const char *twl, int lane, char **xw, const char **xr,
                                               const char **tw_lane) {
    const int c = fft_column(lane), q = lane >> 4;
    *xw = wave_buf + q * kFftRowBytes + c * 8;
    *xr = wave_buf + q * kFftRowBytes + c * 136;
    *tw_lane = twl + c * 8;
}

// value of the same register in the neighbouring lane (lane ^ 1)
__device__ __forceinline__ float lane_pair(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true));
}

// p[k2] = a at index 256 - k, k = c + 16 k2 (index 0 pairs with itself): the neighbouring lane (column 16 - c) holds it in
// register 15 - k2; column 0 holds its own in register 16 - k2, column 8 in register 15 - k2
__device__ __forceinline__ void fft_partner(const cpx (&a)[16], cpx (&p)[16], int c) {
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
        const cpx t = {lane_pair(a[15 - k2].x), lane_pair(a[15 - k2].y)};
        const cpx own = c == 0 ? a[(16 - k2) & 15] : a[15 - k2];
        p[k2] = (c & 7) == 0 ? own : t;
    }
}

// ---- helpers of the weight-resident bf16 kernels.  Gate nonlinearities use the hardware transcendentals (v_exp_f32,
// v_rcp_f32): the bf16 configuration is specified to a tolerance, not bit for bit.

__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
// sigmoid of the heads: the spec's polynomial form in the fp32 configuration (bit-comparable with the oracle); in the bf16
// configuration, which is specified to a tolerance, the hardware exp2 / rcp (<= 2 ulp) -- the mask head alone evaluates
// 67 M sigmoids per call of the bench, and the polynomial exp plus an IEEE division made it VALU-bound
template <class P>
__device__ __forceinline__ float head_sigmoid(float x) {
    if (P::kPrec == kBf16) return fast_sigmoid(x);
    return kns_sigmoid(x);
}
__device__ __forceinline__ float fast_tanh(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.88539008177792681f));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// Buffer addressing (descriptor in 4 SGPRs + a 32-bit per-lane offset + a scalar offset): for streams whose per-lane part
// of the address is a launch constant it takes the 64-bit address arithmetic and the address registers out of the loop.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *) base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8 buf_load_frag(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// 16-byte buffer stores take NO scalar offset here (fold it into the descriptor's base): with one, gfx950 reads the
// upper data dwords late and hipcc 7.2 lets the next VALU instruction overwrite them (kns_gemm.hip, gemm_wsr_kernel).
__device__ __forceinline__ void buf_store_frag(__amdgpu_buffer_rsrc_t r, unsigned voff, bf16x8 v) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
}
__device__ __forceinline__ void buf_store_f32x4(__amdgpu_buffer_rsrc_t r, unsigned voff, f32x4 v) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
}
// `gi` is written once and read once (428 MB per layer at the bench shape, more than the 256 MB memory-side cache): both sides
// mark it non-temporal (aux = 2), which leaves the cache to the hidden sequences the next kernels re-read -- +0.5 % on the step, the
// narrow heads 42 -> 39 us (round 4; either side alone measured -0.2 ... -0.4 %)
constexpr int kGiStreamAux = 2;
__device__ __forceinline__ void buf_store_gi(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, f16x4 v) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, kGiStreamAux);
}

// a * b + c of the bf16 configuration's gate arithmetic: fused (what the spec and the oracle write: n = tanh(fma(r, gh_n,
// gi_n)), h' = fma(z, h - n, n)); -DKNS_GATE_UNFUSED restores round 1's separate multiply and add for A/B runs
__device__ __forceinline__ f32x2 gate_fma2(f32x2 a, f32x2 b, f32x2 c) {
#ifdef KNS_GATE_UNFUSED
    return a * b + c;
#else
    return __builtin_elementwise_fma(a, b, c);
#endif
}
__device__ __forceinline__ float gate_fma1(float a, float b, float c) {
#ifdef KNS_GATE_UNFUSED
    return a * b + c;
#else
    return __builtin_fmaf(a, b, c);
#endif
}
// (float) half HI of the packed fp16 pair x, plus t / times-and-plus: one v_fma_mix_f32 each (an f16 operand of an f32 fma), one
// rounding -- exactly the add / fma on the converted value.  Written as plain fma so that hipcc SELECTS v_fma_mix_f32 itself:
// as inline asm the instructions were invisible to its hazard recognizer, which then kept no wait states between an MFMA or a
// transcendental and these readers of its result (gate values varied from run to run), and explicit s_nops cost what the
// shorter arithmetic had gained.  `one` is 1.0f the optimizer cannot see through (an fma by a literal 1 would be folded into an
// add of the converted half: two instructions).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float opaque_one() {
    float one = 1.0f;
    asm("" : "+v"(one));
    return one;
}
template <int HI>
__device__ __forceinline__ float mix_add(unsigned x, float t, float one) {
    return __builtin_fmaf((float) __builtin_bit_cast(f16x2, x)[HI], one, t);
}
template <int HI>
__device__ __forceinline__ float mix_fma(float a, float b, unsigned x) {
    return __builtin_fmaf(a, b, (float) __builtin_bit_cast(f16x2, x)[HI]);
}

// ---- the bf16 configuration's gate arithmetic, shared by every bf16 recurrent kernel.  The operands arrive PRE-SCALED (the
// constants of the exponentials are folded into the packed weights and biases, kns_layout.h) and b_hh is part of the recurrent
// GEMM itself (two rows of the packed W_hh against a constant 1 in the operand, kBiasK0), so a hidden unit costs 9 plain vector
// operations and 6 transcendentals and a step fetches no bias:
//   r = 1 / (1 + 2^(gi_r + gh_r))   z = 1 / (1 + 2^(gi_z + gh_z))   n = 1 - 2 / (1 + 2^(fma(r, gh_n, gi_n)))   h' = fma(z, h - n, n)
__device__ __forceinline__ float gate_rcp1p_exp2(float y) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y)); }
// gi as packed fp16 pairs (element 2 p and 2 p + 1 of the C fragment in word p), gh as fp32 C fragments, h the previous state
__device__ __forceinline__ f32x4 gate_block_bf16(const unsigned (&pr)[2], const unsigned (&pz)[2], const unsigned (&pn)[2],
                                                 const f32x4 &ar_, const f32x4 &az_, const f32x4 &an_, const f32x4 &hprev) {
    const f32x4 &ar = ar_, &az = az_, &an = an_;
    const float one = opaque_one();
    f32x4 hnew;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float r0 = gate_rcp1p_exp2(mix_add<0>(pr[p], ar[2 * p], one)), r1 = gate_rcp1p_exp2(mix_add<1>(pr[p], ar[2 * p + 1], one));
        const float z0 = gate_rcp1p_exp2(mix_add<0>(pz[p], az[2 * p], one)), z1 = gate_rcp1p_exp2(mix_add<1>(pz[p], az[2 * p + 1], one));
        const float q0 = gate_rcp1p_exp2(mix_fma<0>(r0, an[2 * p], pn[p])), q1 = gate_rcp1p_exp2(mix_fma<1>(r1, an[2 * p + 1], pn[p]));
        const float n0 = __builtin_fmaf(q0, -2.0f, 1.0f), n1 = __builtin_fmaf(q1, -2.0f, 1.0f);
        hnew[2 * p] = __builtin_fmaf(z0, hprev[2 * p] - n0, n0);
        hnew[2 * p + 1] = __builtin_fmaf(z1, hprev[2 * p + 1] - n1, n1);
    }
    return hnew;
}
// the same on one element with the pre-activations already converted to fp32 (exactly: fp16 -> fp32 is exact, and
// fma(x, 1, t) = x + t)
__device__ __forceinline__ float gate_elem_bf16(float xr, float xz, float xn, float ar, float az, float an, float hprev) {
    const float r = gate_rcp1p_exp2(xr + ar), z = gate_rcp1p_exp2(xz + az);
    const float q = gate_rcp1p_exp2(__builtin_fmaf(r, an, xn));
    const float n = __builtin_fmaf(q, -2.0f, 1.0f);
    return __builtin_fmaf(z, hprev - n, n);
}

__device__ __forceinline__ f32x2 fast_sigmoid2(f32x2 x) {
    f32x2 e = x * f32x2{-1.44269504088896341f, -1.44269504088896341f};
    e = f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + f32x2{1.0f, 1.0f};
    return f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
}
__device__ __forceinline__ f32x2 fast_tanh2(f32x2 x) {
    f32x2 e = x * f32x2{2.88539008177792681f, 2.88539008177792681f};
    e = f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + f32x2{1.0f, 1.0f};
    f32x2 r = f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    return gate_fma2(r, f32x2{-2.0f, -2.0f}, f32x2{1.0f, 1.0f});  // 1 - 2 r (exact doubling: the same value as 1 - (r + r))
}

}  // namespace kns
