// kns_stft.hip -- analysis (int16 -> STFT -> features), synthesis (mask x spectrum -> iSTFT -> OLA -> int16) and state reset;
// SURVEY.md 8a rows a2, a3, a5, a6.  See kns_kernels.h for the launch interface.
#include "kns_device.hpp"

namespace kns {

// ------------------------------------------------------------------------------------------------ shared pieces

// One row of 16 lanes owns one stream; a wave four streams; a workgroup (4 waves) the 16 streams of an m-tile.
// The lane in column c of its row (c = fft_column(lane), kns_device.hpp) holds the 16 points n = c + 16 j of whatever
// 256-point complex sequence is live: packed samples z[n] = x[2n] + i x[2n+1], FFT bins, inverse-FFT input.  Every table
// the lane needs is indexed by such n, so it sits in LDS in natural order and is read at (lane base + immediate offset).
constexpr int kOffTw = 0;                               // float2[512]: exp(-2 pi i k / 512)
constexpr int kOffWin = 4096;                           // float[512]:  sin(pi n / 512) / 32768 (analysis: int16 -> windowed sample)
constexpr int kOffWinS = kOffWin + 2048;                // float[512]:  sin(pi n / 512) / 256   (synthesis: FFT sum -> windowed sample)
constexpr int kOffTwl = kOffWinS + 2048;                // fft_fill_twiddles table
constexpr int kOffXbuf = kOffTwl + kFftTwiddleBytes;    // 4 waves x kFftWaveBytes
constexpr int kOffStftEnd = kOffXbuf + 4 * kFftWaveBytes;

// kSynth: also the synthesis window.  (The analysis kernel leaves that table out and puts its exchange tiles there: 2 KiB
// decide whether three of its workgroups fit a CU's 160 KiB.)
// The tables of a workgroup, requested by its 256 STFT threads in ONE go and stored to LDS when they are there (as loops of
// load -> store, hipcc waited for every load before the next one was requested: five memory round trips at the head of every
// launch, most of a one-frame call's STFT kernels).
struct StftTables {
    float2 tw[2];
    float win[2];
    float2 twl;
};
__device__ __forceinline__ void stft_request_tables(StftTables &t, const float *twiddle, const float *window, int tid) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        t.tw[q] = ((const float2 *) twiddle)[tid + 256 * q];
        t.win[q] = window[tid + 256 * q];
    }
    t.twl = ((const float2 *) twiddle)[2 * (((tid & 15) * (tid >> 4)) & 255)];  // fft_fill_twiddles' entry of this thread
}
template <bool kSynth>
__device__ __forceinline__ void stft_store_tables(const StftTables &t, char *smem, int tid) {
    float2 *tw = (float2 *) (smem + kOffTw);
    float *win = (float *) (smem + kOffWin), *wins = (float *) (smem + kOffWinS);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + 256 * q;
        tw[i] = t.tw[q];
        // the powers of two of the spec's (x / 32768) w and (y / 256) w are folded into the window: exact, so the
        // products are the spec's bit for bit
        win[i] = t.win[q] * (1.0f / 32768.0f);
        if (kSynth) wins[i] = t.win[q] * (1.0f / 256.0f);
    }
    ((float2 *) (smem + (kSynth ? kOffTwl : kOffWinS)))[tid] = t.twl;
}
// analysis kernel: [tw | win | fft twiddles | exchange tiles | mean, scale | feature tile]
constexpr int kOffATwl = kOffWinS, kOffAXbuf = kOffATwl + kFftTwiddleBytes, kOffAEnd = kOffAXbuf + 4 * kFftWaveBytes;

// the eight dwords (sample pairs) this lane owns of one 256-sample frame: pairs 16 jj + c
__device__ __forceinline__ void load_frame(int (&f)[8], const int16_t *frame, int c) {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) f[jj] = ((const int *) frame)[16 * jj + c];
}

// windowed packed samples of the 512-sample analysis block [prev | cur]; win_c = LDS window + 8 c
__device__ __forceinline__ void window_block(cpx (&v)[16], const int (&prev)[8], const int (&cur)[8], const char *win_c) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int pr = j < 8 ? prev[j] : cur[j - 8];
        const float2 w = *(const float2 *) (win_c + j * 128);  // window[2n], window[2n + 1], n = 16 j + c
        const float lo = (float) (int16_t) (pr & 0xffff), hi = (float) (int16_t) (pr >> 16);
        v[j].x = lo * w.x;  // = (lo / 32768) window[2n]: the window table carries the 2^-15
        v[j].y = hi * w.y;
    }
}

// Half spectrum of the real 512-point block from the 256-point FFT Z of its packed samples: X[k], k = c + 16 k2.
// Bin 0 carries {X[0], X[256]} (both real).  The arithmetic is the same in the analysis and the synthesis kernel, so a
// spectrum that is recomputed instead of stored is the stored one bit for bit.  tw_c = LDS exp(-2 pi i k / 512) + 8 c.
__device__ __forceinline__ void real_spectrum(const cpx (&z)[16], cpx (&x)[16], const char *tw_c, int c) {
    cpx zp[16];
    fft_partner(z, zp, c);
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
        const cpx zk = z[k2];
        const float2 w = *(const float2 *) (tw_c + k2 * 128);
        // s = Z[k] + conj(Z[256 - k]), d = Z[k] - conj(Z[256 - k])
        const cpx s = {zk.x + zp[k2].x, zk.y - zp[k2].y}, d = {zk.x - zp[k2].x, zk.y + zp[k2].y};
        const cpx p = cmul(d, cpx{w.x, w.y});
        cpx r = {0.5f * (s.x + p.y), 0.5f * (s.y - p.x)};
        if (k2 == 0 && c == 0) r = cpx{zk.x + zk.y, zk.x - zk.y};
        x[k2] = r;
    }
}

// ------------------------------------------------------------------------------------------------ analysis

template <class P>
__device__ __forceinline__ float feature_log(float x) {
    // fp32 configuration: the spec's full-precision polynomial.  bf16 configuration: the feature is rounded to bf16 (8 bits) right
    // after, so a short polynomial serves -- an exactly reproducible one (kns_log_fast), not the hardware logarithm: the features of
    // the two configurations are each bit-identical to the oracle's.
    if (P::kPrec == kBf16) return kns_log_fast(x);
    return kns_log(x);
}

// grid (stream tiles, time segments): a workgroup walks the frames [t0, t1) of its 16 streams, so every PCM sample is
// read once (the previous frame stays in registers) and the per-lane constants are set up once per segment.
template <class P, bool kSpec>
__global__ __launch_bounds__(256, 3) void analysis_kernel(AnalysisArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lmean = (float *) (smem + kOffAEnd), *lscale = lmean + 272;
    typename P::elem_t *tile = (typename P::elem_t *) (smem + kOffAEnd + 2 * 272 * 4);  // nbf KiB

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = fft_column(lane), q = lane >> 4, row = wave * 4 + q;
    const int mt = blockIdx.x, mtiles = g.Bpad >> 4;
    const int t0 = blockIdx.y * g.seg, t1 = min(g.T, t0 + g.seg);
    const int b = mt * 16 + row;
    const size_t row_len = (size_t) (g.pitch ? g.pitch : g.T) * kFrame;
    // rows past the last stream (ragged last tile) read the last stream's samples: their results are never stored
    // anywhere a stream would see them, and the loop carries no conditional vector-memory operation
    const int16_t *pcm_row = g.pcm + (size_t) (b < g.B ? b : g.B - 1) * row_len;

    // (one-frame calls of a several-frame front-end) this m-tile's feature history, slots 1 .. hist_slots, on its way to LDS; it
    // goes back one slot down after the first barrier, when all of it has been read
    char *hroll = smem + kOffAEnd + 2 * 272 * 4 + (size_t) g.nbf * 1024;
    if (g.feat_hist) {
        typedef const __attribute__((address_space(1))) void *gptr_t;
        typedef __attribute__((address_space(3))) void *lptr_t;
        for (int i = wave; i < g.hist_slots * g.nbf; i += 4) {
            const int slot = i / g.nbf, kb = i - slot * g.nbf;
            __builtin_amdgcn_global_load_lds(
                (gptr_t) ((const uint4 *) g.feat_hist + (((size_t) (slot + 1) * mtiles + mt) * g.nbf + kb) * 64 + lane),
                (lptr_t) (hroll + i * 1024), 16, 0, 0);
        }
    }
    int prev[8], cur[8], nxt[8];
    // (a slice of a longer call, prev_in_pcm: the frame in front of frame 0 sits in the row itself, at t = -1)
    load_frame(prev, t0 == 0 && !g.prev_in_pcm ? g.hist_in + (size_t) b * kFrame : pcm_row + ((ptrdiff_t) t0 - 1) * kFrame, c);
    load_frame(cur, pcm_row + (size_t) t0 * kFrame, c);
    StftTables tbl;
    stft_request_tables(tbl, g.twiddle, g.window, tid);
    float ms[2][2];  // mean, scale of bins tid and tid + 256 (clamped: 257 bins)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + 256 * q < kBins ? tid + 256 * q : kBins - 1;
        ms[q][0] = g.mean[i];
        ms[q][1] = g.scale[i];
    }
    stft_store_tables<false>(tbl, smem, tid);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + 256 * q;
        if (i < kBins) {
            lmean[i] = ms[q][0];
            lscale[i] = ms[q][1];
        }
    }
    {
        uint4 *z = (uint4 *) tile;
        for (int i = tid; i < g.nbf * 64; i += 256) z[i] = uint4{0, 0, 0, 0};
    }
    __syncthreads();
    if (g.feat_hist) {
        for (int i = tid; i < g.hist_slots * g.nbf * 64; i += 256) {
            const int slot = i / (g.nbf * 64), w = i - slot * g.nbf * 64;
            ((uint4 *) g.feat_hist)[((size_t) slot * mtiles + mt) * g.nbf * 64 + w] = ((const uint4 *) hroll)[i];
        }
    }
    char *xw;
    const char *xr, *twl_c;
    fft_lane_bases(smem + kOffAXbuf + wave * kFftWaveBytes, smem + kOffATwl, lane, &xw, &xr, &twl_c);
    const char *tw_c = smem + kOffTw + c * 8, *win_c = smem + kOffWin + c * 8;
    const float *mean_c = lmean + c, *scale_c = lscale + c;

    for (int t = t0; t < t1; ++t) {
        // next frame's samples are requested before this frame is transformed (clamped, never skipped)
        load_frame(nxt, pcm_row + (size_t) (t + 1 < t1 ? t + 1 : t) * kFrame, c);
        cpx v[16];
        window_block(v, prev, cur, win_c);
        fft256_rows(v, twl_c, xw, xr);
        cpx x[16];
        real_spectrum(v, x, tw_c, c);
        if (kSpec) {
            // stored in the lane order of this kernel pair: bins (c + 16 k2, c + 16 (k2 + 1)), k2 even, as one 16-byte word
            // at [(k2 / 2) * 16 + c] -- a row of lanes moves 256 contiguous bytes per instruction
            f32x4 *spec = (f32x4 *) g.spec + ((size_t) t * g.Bpad + b) * 128;
#pragma unroll
            for (int k2 = 0; k2 < 16; k2 += 2) spec[(k2 >> 1) * 16 + c] = f32x4{x[k2].x, x[k2].y, x[k2 + 1].x, x[k2 + 1].y};
        }
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const int k = c + 16 * k2;
            float pw = __builtin_fmaf(x[k2].x, x[k2].x, x[k2].y * x[k2].y);
            if (k2 == 0 && c == 0) pw = x[k2].x * x[k2].x;
            const float ft = (feature_log<P>(pw + 1e-10f) - mean_c[16 * k2]) * scale_c[16 * k2];
            tile[(k / P::KB) * 64 * P::EPL + P::off(row, k % P::KB)] = P::cvt(ft);
        }
        if (c == 0) {  // Nyquist bin: the imaginary slot of bin 0
            const float nyq = x[0].y;
            const float fn = (feature_log<P>(nyq * nyq + 1e-10f) - lmean[256]) * lscale[256];
            tile[(256 / P::KB) * 64 * P::EPL + P::off(row, 256 % P::KB)] = P::cvt(fn);
        }
        __syncthreads();
        {
            const uint4 *src = (const uint4 *) tile;
            uint4 *dst = (uint4 *) g.feat + ((size_t) t * mtiles + mt) * g.nbf * 64;
            for (int i = tid; i < g.nbf * 64; i += 256) dst[i] = src[i];
        }
        __syncthreads();  // (a single tile and two barriers per frame: a second tile would cost the third workgroup per CU)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            prev[jj] = cur[jj];
            cur[jj] = nxt[jj];
        }
    }
    if (t1 == g.T && b < g.B && g.write_hist) {  // history for the next call: the last frame (prev after the final rotation)
        int *h = (int *) (g.hist_out + (size_t) b * kFrame);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) h[16 * jj + c] = prev[jj];
    }
}

void launch_analysis(const AnalysisArgs &a, hipStream_t s) {
    dim3 grid(a.Bpad / 16, (a.T + a.seg - 1) / a.seg);
    const size_t roll = a.feat_hist ? (size_t) a.hist_slots * a.nbf * 1024 : 0;  // (one-frame calls: one workgroup per CU is plenty)
#ifndef KNS_STFT_LDS_PAD  // timing experiment: extra dynamic LDS per workgroup = fewer workgroups per CU
#define KNS_STFT_LDS_PAD 0
#endif
    const size_t lds = kOffAEnd + 2 * 272 * 4 + (size_t) a.nbf * 1024 + roll + KNS_STFT_LDS_PAD;
    auto go = [&](auto kernel, int threads, size_t bytes) {
        if (bytes > 48 * 1024) (void) hipFuncSetAttribute((const void *) kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
        hipLaunchKernelGGL(kernel, grid, dim3(threads), bytes, s, a);
    };
    if (a.precision == kBf16) {
        if (a.write_spec)
            go(analysis_kernel<PBF16, true>, 256, lds);
        else
            go(analysis_kernel<PBF16, false>, 256, lds);
    } else {
        if (a.write_spec)
            go(analysis_kernel<PF32, true>, 256, lds);
        else
            go(analysis_kernel<PF32, false>, 256, lds);
    }
}

// ------------------------------------------------------------------------------------------------ synthesis

// grid (stream tiles, time segments); a workgroup produces frames [t0, t1) of its 16 streams.  A segment that does not
// start at 0 first replays frame t0 - 1 (no output) to rebuild the overlap-add tail it inherits.
// kRecompute: the spectrum of a frame is rebuilt from its 512 B of PCM (one more FFT in registers) instead of being read
// back as 2 KiB of fp32 that the analysis kernel would have had to write: the arithmetic is the analysis kernel's, so
// the result is the same bit for bit.  The stored form remains for single-frame calls, where the analysis kernel
// updates the history in place and the previous frame is gone by the time this kernel runs.
// (three waves per SIMD: the kernel is bound by VALU issue, and two waves on a SIMD reach an instruction every ~2.8 cycles,
// three come close to the pipe's 2; the register budget of 168 is what decides which loads are prefetched below)
// kMaskIn (one-frame calls, bf16): waves 4..7 are the mask head -- sigmoid(h . W_mask + b_mask) of the workgroup's m-tile as fp16 C
// fragments in LDS (what gemm_kernel<kOutMask> stores), complete at the workgroup's one barrier; the STFT waves read it from there.
template <bool kRecompute, bool kMaskH, bool kMaskIn>  // kMaskH: the mask travels as fp16 C fragments (bf16 configuration), else fp32
__global__ __launch_bounds__(kMaskIn ? 512 : 256, kMaskIn ? 1 : 3) void synthesis_kernel(SynthesisArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(!kMaskIn || (kMaskH && !kRecompute), "mask head inside: bf16, stored spectrum");

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (kMaskIn && wave >= 4) {
        typedef PBF16 P;
        typedef P::frag_t frag_t;
        constexpr int NB = P::NBH, kPer = (kMaskTiles + 3) / 4;
        const int gw = wave - 4, colq = lane & 15;
        frag_t a[NB], w[kPer][NB];
        float bias[kPer];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) a[kb] = ((const frag_t *) g.mask_h)[((size_t) blockIdx.x * NB + kb) * 64 + lane];
#pragma unroll
        for (int q = 0; q < kPer; ++q) {  // n-tiles gw, gw + 4, ...; a slot past the last tile repeats it (same words, same values)
            const int nt = gw + 4 * q < kMaskTiles ? gw + 4 * q : kMaskTiles - 1;
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) w[q][kb] = ((const frag_t *) g.mask_w)[((size_t) nt * NB + kb) * 64 + lane];
            bias[q] = g.mask_b[nt * 16 + colq];
        }
        __builtin_amdgcn_sched_barrier(0);  // every request before the first use (the scheduler would sink the loads to save registers)
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int nt = gw + 4 * q < kMaskTiles ? gw + 4 * q : kMaskTiles - 1;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) acc = P::mma(a[kb], w[q][kb], acc);
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = head_sigmoid<P>(acc[i] + bias[q]);
            ((f16x4 *) (smem + kOffStftEnd))[nt * 64 + lane] = __builtin_convertvector(v, f16x4);
        }
        __syncthreads();
        return;
    }
    const int c = fft_column(lane), q = lane >> 4, row = wave * 4 + q;
    const int mt = blockIdx.x, mtiles = g.Bpad >> 4;
    const int t0 = blockIdx.y * g.seg, t1 = min(g.T, t0 + g.seg);
    const int tb = t0 > 0 ? t0 - 1 : 0;
    const int b = mt * 16 + row;
    const bool valid = b < g.B;
    const size_t row_len = (size_t) (g.pitch ? g.pitch : g.T) * kFrame;
    const int16_t *pcm_row = g.pcm + (size_t) (valid ? b : g.B - 1) * row_len;  // (ragged tile: see analysis_kernel)

    int prev[8], cur[8], nxt[8];
    if (kRecompute) {
        load_frame(prev, tb == 0 && !g.prev_in_pcm ? g.hist_in + (size_t) b * kFrame : pcm_row + ((ptrdiff_t) tb - 1) * kFrame, c);
        load_frame(cur, pcm_row + (size_t) tb * kFrame, c);
    }
    // overlap-add tail of this lane's points n = c + 16 k2, k2 = 8..15 (samples 2n, 2n + 1 of the block's second half)
    cpx tl[8];
    {
        const float2 *tp = (const float2 *) (g.tail_in + (size_t) b * kFrame);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float2 v = tp[c + 16 * j];
            tl[j] = cpx{v.x, v.y};
        }
    }
    // mask element (row, k): C-packed tile k / 16, lane (row >> 2) * 16 + (k & 15), value row & 3 -- for a fixed k2 the
    // wave reads one contiguous 1 KiB tile
    const unsigned mlane = ((unsigned) (wave * 16 + c) * 4u + (unsigned) q) * 4u;
    float mk[17], mkn[17];
    auto mask_fetch = [&](float (&m)[17], int t) {
        if (kMaskIn) {  // this workgroup's tile, computed by waves 4..7 into LDS
            const _Float16 *ml = (const _Float16 *) (smem + kOffStftEnd);
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) m[k2] = (float) ml[k2 * 256 + (mlane >> 2)];
            m[16] = (float) ml[16 * 256 + (wave * 16) * 4 + q];
            return;
        }
        if (kMaskH) {  // fp16 tiles of 512 B; the conversion to fp32 is exact
            const __amdgpu_buffer_rsrc_t mr =
                make_rsrc((const char *) g.mask + ((size_t) t * mtiles + mt) * kMaskTiles * 512, kMaskTiles * 512);
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2)
                m[k2] = (float) __builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(mr, mlane >> 1, k2 * 512u, 0));
            m[16] = (float) __builtin_bit_cast(
                _Float16, __builtin_amdgcn_raw_buffer_load_b16(mr, ((unsigned) (wave * 16) * 4u + (unsigned) q) * 2u, 16 * 512u, 0));
            return;
        }
        const __amdgpu_buffer_rsrc_t mr = make_rsrc(g.mask + ((size_t) t * mtiles + mt) * kMaskTiles * 256, kMaskTiles * 1024);
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2)
            m[k2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mr, mlane, k2 * 1024u, 0));
        m[16] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mr, ((unsigned) (wave * 16) * 4u + (unsigned) q) * 4u, 16 * 1024u, 0));
    };
    cpx xs[16], xsn[16];
    auto spec_fetch = [&](cpx (&dst)[16], int t) {
        const f32x4 *spec = (const f32x4 *) g.spec + ((size_t) t * g.Bpad + b) * 128;
#pragma unroll
        for (int k2 = 0; k2 < 16; k2 += 2) {
            const f32x4 v = spec[(k2 >> 1) * 16 + c];
            dst[k2] = cpx{v[0], v[1]};
            dst[k2 + 1] = cpx{v[2], v[3]};
        }
    };
    // stored spectrum: the first frame's operands are requested before the tables are waited for (one memory round trip in front
    // of the first FFT, not two)
    if (!kRecompute) {
        if (!kMaskIn) mask_fetch(mk, tb);
        spec_fetch(xs, tb);
    }
    StftTables tbl;
    stft_request_tables(tbl, g.twiddle, g.window, tid);
    stft_store_tables<true>(tbl, smem, tid);
    __syncthreads();
    char *xw;
    const char *xr, *twl_c;
    fft_lane_bases(smem + kOffXbuf + wave * kFftWaveBytes, smem + kOffTwl, lane, &xw, &xr, &twl_c);
    const char *tw_c = smem + kOffTw + c * 8, *win_c = smem + kOffWin + c * 8, *wins_c = smem + kOffWinS + c * 8;
    if (kMaskIn) mask_fetch(mk, tb);  // (complete since the barrier)

    for (int t = tb; t < t1; ++t) {
        const bool emit = t >= t0;
        const int tn = t + 1 < t1 ? t + 1 : t;
        // recompute: the mask is not needed before the first FFT and the spectrum are through (~700 instructions), so it is
        // requested here without a second register set; stored spectrum: mask and spectrum of the next frame in flight
        if (kRecompute)
            mask_fetch(mk, t);
        else
            mask_fetch(mkn, tn);
        cpx x[16];
        if (kRecompute) {
            load_frame(nxt, pcm_row + (size_t) tn * kFrame, c);
            cpx v[16];
            window_block(v, prev, cur, win_c);
            fft256_rows(v, twl_c, xw, xr);
            real_spectrum(v, x, tw_c, c);
        } else {
            spec_fetch(xsn, tn);
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) x[k2] = xs[k2];
        }
        // Y = mask . X; bin 0 holds DC and Nyquist (both real): its slot travels as {m[0] X[0], m[256] X[256]} and is taken
        // apart again below
        cpx y[16];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) y[k2] = cpx{mk[k2] * x[k2].x, mk[k2] * x[k2].y};
        cpx yp[16];
        fft_partner(y, yp, c);
        cpx v[16];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            cpx yk = y[k2], yq = yp[k2];  // Y[k], Y[256 - k]
            if (k2 == 0 && c == 0) {
                yk = cpx{mk[0] * x[0].x, 0.0f};
                yq = cpx{mk[16] * x[0].y, 0.0f};
            }
            const float2 w = *(const float2 *) (tw_c + k2 * 128);
            // e = Y[k] + conj(Y[256 - k]), d = Y[k] - conj(Y[256 - k]), o = conj(W^k) d
            const cpx e = {yk.x + yq.x, yk.y - yq.y}, d = {yk.x - yq.x, yk.y + yq.y};
            const cpx o = cmul(d, cpx{w.x, -w.y});
            // Z' = E + i O (both carry the factor 1/2); fed to the forward FFT with re/im swapped = inverse FFT
            const float zr = 0.5f * (e.x - o.y), zi = 0.5f * (e.y + o.x);
            v[k2] = cpx{zi, zr};
        }
        fft256_rows(v, twl_c, xw, xr);
        int packed[8];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const float2 w = *(const float2 *) (wins_c + k2 * 128);  // window[2n] / 256, window[2n + 1] / 256, n = c + 16 k2
            // swapped output: re <-> im; = (v / 256) window: the table carries the 2^-8
            const float y0 = v[k2].y * w.x, y1 = v[k2].x * w.y;
            if (k2 < 8) {
                float a0 = (tl[k2].x + y0) * 32768.0f, a1 = (tl[k2].y + y1) * 32768.0f;
                a0 = __builtin_fminf(__builtin_fmaxf(__builtin_roundf(a0), -32768.0f), 32767.0f);
                a1 = __builtin_fminf(__builtin_fmaxf(__builtin_roundf(a1), -32768.0f), 32767.0f);
                packed[k2] = ((int) a0 & 0xffff) | ((int) a1 << 16);
            } else {
                v[k2] = cpx{y0, y1};
            }
        }
#pragma unroll
        for (int k2 = 8; k2 < 16; ++k2) tl[k2 - 8] = v[k2];
        {
            // one wave-uniform descriptor over the wave's four stream rows of this frame (a per-lane base would make hipcc
            // wrap every store in a loop over the distinct descriptors); rows that must not be written -- the replayed
            // frame, streams past the last one -- fall outside the descriptor's range and are dropped by the hardware
            const int row0 = mt * 16 + wave * 4;
            const int rows_ok = row0 < g.B ? min(4, g.B - row0) : 0;
            const unsigned span = emit && rows_ok ? (unsigned) ((size_t) (rows_ok - 1) * row_len * 2 + kFrame * 2) : 0u;
            const __amdgpu_buffer_rsrc_t o = make_rsrc(g.out + (size_t) row0 * row_len + (size_t) t * kFrame, span);
            const unsigned voff = (unsigned) q * (unsigned) (row_len * 2) + (unsigned) c * 4u;
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) __builtin_amdgcn_raw_buffer_store_b32(packed[k2], o, voff, 64u * k2, 0);
        }
        if (kRecompute) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                prev[jj] = cur[jj];
                cur[jj] = nxt[jj];
            }
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) xs[k2] = xsn[k2];
        }
        if (!kRecompute) {
#pragma unroll
            for (int k2 = 0; k2 < 17; ++k2) mk[k2] = mkn[k2];
        }
    }
    if (t1 == g.T) {
        float2 *tp = (float2 *) (g.tail_out + (size_t) b * kFrame);
#pragma unroll
        for (int j = 0; j < 8; ++j) tp[c + 16 * j] = float2{tl[j].x, tl[j].y};
    }
}

void launch_synthesis(const SynthesisArgs &a, hipStream_t s) {
    const size_t lds = kOffStftEnd + KNS_STFT_LDS_PAD;
    const dim3 grid(a.Bpad / 16, (a.T + a.seg - 1) / a.seg);
#if KNS_STFT_LDS_PAD
    (void) hipFuncSetAttribute((const void *) synthesis_kernel<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
#endif
    if (a.mask_w && !a.recompute && a.mask_fp16 && a.T == 1)
        hipLaunchKernelGGL((synthesis_kernel<false, true, true>), grid, dim3(512), lds + kMaskTiles * 512, s, a);
    else if (a.recompute && a.mask_fp16)
        hipLaunchKernelGGL((synthesis_kernel<true, true, false>), grid, dim3(256), lds, s, a);
    else if (a.recompute)
        hipLaunchKernelGGL((synthesis_kernel<true, false, false>), grid, dim3(256), lds, s, a);
    else if (a.mask_fp16)
        hipLaunchKernelGGL((synthesis_kernel<false, true, false>), grid, dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL((synthesis_kernel<false, false, false>), grid, dim3(256), lds, s, a);
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
    // front-end context: this stream's row of every k-block of every remembered frame = the feature of a silent frame (in
    // an A-packed block a row is the four 16-byte words of lanes row, row + 16, row + 32, row + 48)
    for (int i = tid; i < g.fhist_frames * g.nbf * 4; i += 256) {
        const int q = i & 3, kb = (i >> 2) % g.nbf, f = (i >> 2) / g.nbf;
        const size_t word = (size_t) kb * 64 + row + 16 * q;
        ((uint4 *) g.fhist)[((size_t) f * mtiles + mt) * g.nbf * 64 + word] = ((const uint4 *) g.silent)[word];
    }
}

// ---- completion word of a one-frame graph replay: the last node of the captured frame.  Counts the replays on the device and
// publishes the count to a word in page-locked host memory with a system-scope release: a host that sees the count sees the frame's
// output (written to host memory by the synthesis kernel, complete at the kernel boundary before this node runs).
__global__ void frame_done_kernel(unsigned *counter, unsigned *host_word) {
    const unsigned v = *counter + 1u;
    *counter = v;
    __hip_atomic_store(host_word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

void launch_frame_done(unsigned *counter, unsigned *host_word, hipStream_t s) {
    hipLaunchKernelGGL(frame_done_kernel, dim3(1), dim3(1), 0, s, counter, host_word);
}

void launch_reset(const ResetArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(reset_kernel, dim3(a.Bpad), dim3(256), 0, s, a);
}

}  // namespace kns
