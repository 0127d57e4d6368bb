// kns_gemm.hip -- every GEMM over stream-frames: generic, weight-stationary input GEMMs, narrow GEMMs, heads.
#include "kns_device.hpp"

#include <type_traits>

namespace kns {

// ------------------------------------------------------------------------------------------------ GEMM

constexpr int kPF = 4;      // weight prefetch depth in k-blocks

template <class P, int OUT, int kGemmMT>  // kGemmMT: m-tiles (of 16 stream-frames) per workgroup: as many as the A tile leaves room for in LDS
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename P::frag_t frag_t;
    constexpr bool kApack = (OUT == kOutAPlain || OUT == kOutASigmoid);
    constexpr bool kSigmoid = (OUT == kOutMask || OUT == kOutASigmoid);
    constexpr int NU = kApack ? P::NPB : 1;  // n-tiles per unit of work

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = g.nb0 + g.taps * g.nb1;
    const int mt0 = blockIdx.x * kGemmMT;
    const int mcount = min(kGemmMT, g.mtiles - mt0);

    const int units = g.ntiles / NU;
    const int units_per_y = ceil_div(units, (int) gridDim.y);
    const int u_begin = blockIdx.y * units_per_y;
    const int u_end = min(units, u_begin + units_per_y);
    const frag_t *w = (const frag_t *) g.w;
    // The weights of the wave's first unit are requested BEFORE the A tile is staged and the A tile goes global -> LDS directly:
    // one memory round trip in front of the first MFMA instead of two (one-frame calls are made of these: 7.8-8.9 us per launch
    // with the staging loop and the weight queue one after the other).
    frag_t bq[kPF][NU];
    auto prefetch = [&](int u) {
#pragma unroll
        for (int p = 0; p < kPF; ++p)
            if (p < nb)
#pragma unroll
                for (int j = 0; j < NU; ++j) bq[p][j] = w[((size_t) (u * NU + j) * nb + p) * 64 + lane];
    };
    const int u_first = u_begin + wave;
    if (u_first < u_end) prefetch(u_first);

    // stage the A tile in LDS, keeping fragment order: [m-tile][k-block][lane] 16-byte words (a wave copies 1 KiB per request:
    // a lane's 16 bytes land at the wave-uniform LDS address + 16 lane)
    uint4 *lds_a = (uint4 *) smem;
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    auto copy_in = [&](const uint4 *src, uint4 *dst, int kib) {
        for (int ch = wave; ch < kib; ch += 4)
            __builtin_amdgcn_global_load_lds((gptr_t) (src + ch * 64 + lane), (lptr_t) (dst + ch * 64), 16, 0, 0);
    };
    for (int m = 0; m < mcount; ++m) {
        if (g.nb0) copy_in((const uint4 *) g.a0 + (size_t) (mt0 + m) * g.nb0 * 64, lds_a + m * nb * 64, g.nb0);
        for (int tap = 0; tap < g.taps; ++tap)
            copy_in((const uint4 *) ((const char *) g.a1 + tap * g.tap_stride) + (size_t) (mt0 + m) * g.nb1 * 64,
                    lds_a + (m * nb + g.nb0 + tap * g.nb1) * 64, g.nb1);
    }
    for (int m = mcount; m < kGemmMT; ++m)
        for (int i = tid; i < nb * 64; i += 256) lds_a[m * nb * 64 + i] = uint4{0, 0, 0, 0};
    __syncthreads();

    const frag_t *lds_f = (const frag_t *) smem;
    char *scratch = smem + (size_t) kGemmMT * nb * 1024 + (size_t) wave * kGemmMT * 1024;  // per-wave transposer

    for (int u = u_first; u < u_end; u += 4) {
        const int nt0 = u * NU;
        f32x4 acc[NU][kGemmMT];
#pragma unroll
        for (int j = 0; j < NU; ++j)
#pragma unroll
            for (int m = 0; m < kGemmMT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (u != u_first) prefetch(u);
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
                    if (kSigmoid) x = head_sigmoid<P>(x);
                    if (kApack && col >= g.n_valid) x = 0.0f;
                    v[i] = x;
                }
                if (!kApack) {
                    if (m < mcount) {
                        const size_t idx = ((size_t) (mt0 + m) * g.ntiles + nt) * 64 + lane;
                        if (OUT == kOutGi)
                            ((typename P::gi_t *) g.out)[idx] = P::to_gi(v);
                        else if (P::kPrec == kBf16)  // (kOutMask) the bf16 configuration hands the mask over as fp16
                            ((f16x4 *) g.out)[idx] = __builtin_convertvector(v, f16x4);
                        else
                            ((f32x4 *) g.out)[idx] = v;
                    }
                } else if (OUT == kOutASigmoid && g.pad_nb) {  // a narrow head into the padding of an existing matrix (kns_kernels.h)
                    if (m < mcount && col < g.n_valid) {
                        typename P::elem_t *dst = (typename P::elem_t *) g.out + ((size_t) (mt0 + m) * g.pad_nb + g.pad_blk) * 64 * P::EPL;
#pragma unroll
                        for (int i = 0; i < 4; ++i) dst[P::off((lane >> 4) * 4 + i, g.pad_kk0 + col)] = P::cvt(v[i]);
                    }
                } else {
                    typename P::elem_t *sc = (typename P::elem_t *) (scratch + m * 1024);
#pragma unroll
                    for (int i = 0; i < 4; ++i) sc[P::off((lane >> 4) * 4 + i, j * 16 + (lane & 15))] = P::cvt(v[i]);
                }
            }
        }
        if (kApack && !(OUT == kOutASigmoid && g.pad_nb)) {
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

// ---- weight-stationary form of the GRU input-side GEMM (bf16):  Gi = [y_prev ; e] . W_ih + b_ih  over ALL stream-frames: the 51 n-tiles are split over a PAIR of workgroups that sit on
// the same XCD (blocks g and g + 8), so one wave keeps at most 7 n-tiles x (9 + NB0) k-blocks = 77 fragments -- all of
// them in registers (3 n-tiles in VGPRs, 4 pinned in AGPRs), including the y_prev part.  No weight ever comes from LDS
// or L2 inside the loop; LDS only double-buffers A tiles, kWs2Stage m-tiles per barrier.  The partner's second read of
// an A tile hits the XCD's L2.
constexpr int kWs2Waves = 8;    // two waves per SIMD: one wave's MFMAs run under the other's epilogue VALU
constexpr int kWs2Tiles = 4;    // n-tiles per wave (the 4th only on some waves): 26 / 8 -> 4,4,3,...

#ifndef WS2_NA
#define WS2_NA 2
#endif
constexpr int kWs2NA = WS2_NA;  // A fragments in flight per wave

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
    const int mlast = mgroup * kWs2Stage + ((g.mtiles - 1 - mgroup * kWs2Stage) / mstride) * mstride;  // last stage first
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
    // Buffer-addressed: a stage's A tiles are the kWs2Stage consecutive m-tiles at mt0 of a0 (NB0 blocks each) and a1
    // (9 blocks each); the per-lane part of every address is lane * 16 (A) or lane * 8 (gi), the rest is scalar.
    const unsigned lane16 = lane * 16u, lane8 = lane * 8u;
    unsigned foff[kFetch];  // scalar byte offset of fetch i inside its stage window (of a0 if blk < NB0, else of a1)
#pragma unroll
    for (int i = 0; i < kFetch; ++i) {
        const int idx = wave + kWs2Waves * i;
        const int m = idx / NB, blk = idx % NB;
        foff[i] = blk < NB0 ? (unsigned) (m * NB0 + blk) * 1024u : (unsigned) (m * P::NBH + (blk - NB0)) * 1024u;
    }
    auto fetch = [&](int i, int mt0) {
        const int blk = (wave + kWs2Waves * i) % NB;
        const __amdgpu_buffer_rsrc_t r =
            blk < NB0 ? make_rsrc((const frag_t *) g.a0 + (size_t) mt0 * NB0 * 64, kWs2Stage * (NB0 ? NB0 : 1) * 1024)
                      : make_rsrc((const frag_t *) g.a1 + (size_t) mt0 * P::NBH * 64, kWs2Stage * P::NBH * 1024);
        return buf_load_frag(r, lane16, foff[i]);
    };
#pragma unroll
    for (int i = 0; i < kFetch; ++i)
        if (wave + kWs2Waves * i < kStageBlocks) abuf[(wave + kWs2Waves * i) * 64 + lane] = fetch(i, mlast);
    __syncthreads();

    auto store_tile = [&](__amdgpu_buffer_rsrc_t out, int j, f32x4 v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] + bias[j];
        buf_store_gi(out, lane8, nt[j] * 512u, P::to_gi(v));
    };

    // The main loop exists twice, for waves with and without a fourth n-tile: with the choice inside the k loop every
    // k-block carried a scalar branch, and hipcc's s_waitcnt placement across those branches made the MFMA chain wait for
    // the previous m-tile's stores and the in-flight stage loads (vmcnt(N) with N below what was outstanding).
    auto main_loop = [&](auto has4_tag) {  // (mlast defined above: time-descending traversal)
        constexpr bool kHas4 = decltype(has4_tag)::value;
        int cur = 0;
        for (int mt0 = mlast; mt0 >= 0; mt0 -= mstride) {
            const bool more = mt0 - mstride >= 0;
            frag_t stage[kFetch];
#pragma unroll
            for (int i = 0; i < kFetch; ++i)
                if (more && wave + kWs2Waves * i < kStageBlocks) stage[i] = fetch(i, mt0 - mstride);  // in flight during the MFMAs
            // MP = 2 m-tiles at a time against the same register-resident weights (one where the 11-block layers leave
            // no registers for it): 6 or 8 independent accumulator chains per wave instead of 3 or 4, twice as many MFMAs
            // between two LDS waits.  A fragments roll through kWs2NA registers per m-tile: the read of block b + kWs2NA
            // is issued as soon as block b's MFMAs are, and scheduling fences keep it there (left alone hipcc issues every
            // read right before its use and the wave waits an LDS latency per k-block).
            constexpr int MP = NB0 < 2 ? 2 : 1;
            static_assert(kWs2Stage % MP == 0, "m-tiles are processed MP at a time");
            const frag_t *ab = abuf + cur * kStageBlocks * 64;
#pragma unroll
            for (int m = 0; m < kWs2Stage; m += MP) {
                f32x4 acc[MP][kWs2Tiles];
                frag_t qa[MP][kWs2NA];
#pragma unroll
                for (int mm = 0; mm < MP; ++mm) {
#pragma unroll
                    for (int c = 0; c < kWs2Tiles; ++c) acc[mm][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int p = 0; p < kWs2NA; ++p) qa[mm][p] = ab[((m + mm) * NB + p) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
                    for (int c = 0; c < (kHas4 ? 4 : 3); ++c)
#pragma unroll
                        for (int mm = 0; mm < MP; ++mm) acc[mm][c] = P::mma(qa[mm][blk % kWs2NA], wr[c][blk], acc[mm][c]);
                    if (blk + kWs2NA < NB) {
#pragma unroll
                        for (int mm = 0; mm < MP; ++mm)
                            qa[mm][blk % kWs2NA] = ab[((m + mm) * NB + blk + kWs2NA) * 64 + lane];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (m == kWs2Stage - MP) {
                    // hand the next stage's A tiles to LDS before these m-tiles' stores are issued: vmcnt counts stores
                    // too on gfx950, so a wait placed after them would also wait for their write acknowledgements
#pragma unroll
                    for (int i = 0; i < kFetch; ++i)
                        if (more && wave + kWs2Waves * i < kStageBlocks)
                            abuf[((cur ^ 1) * kStageBlocks + wave + kWs2Waves * i) * 64 + lane] = stage[i];
                }
#pragma unroll
                for (int mm = 0; mm < MP; ++mm) {
                    const __amdgpu_buffer_rsrc_t out =
                        make_rsrc((P::gi_t *) g.out + (size_t) (mt0 + m + mm) * kGateTiles * 64, kGateTiles * 512);
#pragma unroll
                    for (int c = 0; c < (kHas4 ? 4 : 3); ++c) store_tile(out, c, acc[mm][c]);
                }
            }
            __syncthreads();
            cur ^= 1;
        }
    };
    if (has4)
        main_loop(std::true_type{});
    else
        main_loop(std::false_type{});
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
    // Buffer-addressed, bounds-checked by the descriptor: a stage window past the last m-tile reads zeros and stores
    // nothing, so the loop below carries no conditional vector-memory operation (with them, hipcc's s_waitcnt placement
    // drained the previous stage's stores -- a write round trip -- at the top of every stage).
    const unsigned lane16 = lane * 16u;
    auto stage_rsrc = [&](int mt) {  // window of kWsrStage m-tiles of A starting at m-tile mt
        const int left = g.mtiles - mt;
        const int n = left <= 0 ? 0 : (left < kWsrStage ? left : kWsrStage);
        return make_rsrc(a1p + (size_t) (left > 0 ? mt : 0) * NB * 64, (unsigned) n * NB * 1024u);
    };
    // Traversal order follows the memory-side cache (256 MB): the heads read the hidden sequence the recurrent kernel
    // has just written (last steps freshest) and the mask is read by the synthesis kernel from step 0 on, so they walk
    // the m-tiles (= time) DOWN; the front-end feeds an input GEMM that walks down, so it walks UP.
    constexpr bool kDown = OUT != kOutAPlain;
    const int mfirst = blockIdx.x * kWsrStage;
    int mt0 = !kDown ? mfirst : (mfirst < g.mtiles ? mfirst + ((g.mtiles - 1 - mfirst) / mstride) * mstride : -1);
    const int mstep = kDown ? -mstride : mstride;
    {
        const __amdgpu_buffer_rsrc_t r = stage_rsrc(mt0 < 0 ? g.mtiles : mt0);
#pragma unroll
        for (int i = 0; i < kFetch; ++i)
            if (wave + 8 * i < kStageBlocks) abuf[(wave + 8 * i) * 64 + lane] = buf_load_frag(r, lane16, (wave + 8 * i) * 1024u);
    }
    __syncthreads();

    // one copy of the main loop per number of live units of the wave (1 .. UW): no per-unit branches in the loop
    auto main_loop = [&](auto nlive_tag) {
        constexpr int kLive = decltype(nlive_tag)::value;
        int cur = 0;
        for (; mt0 >= 0 && mt0 < g.mtiles; mt0 += mstep) {
            const __amdgpu_buffer_rsrc_t rn = stage_rsrc(mt0 + mstep < 0 ? g.mtiles : mt0 + mstep);
            frag_t stage[kFetch];
#pragma unroll
            for (int i = 0; i < kFetch; ++i)
                if (wave + 8 * i < kStageBlocks) stage[i] = buf_load_frag(rn, lane16, (wave + 8 * i) * 1024u);
#pragma unroll
            for (int m = 0; m < kWsrStage; ++m) {
                const int mt = mt0 + m;
                const bool valid = mt < g.mtiles;
                // Output descriptors are per 1 KiB tile (nothing is stored for an m-tile past the end) and the stores carry
                // no scalar offset: on gfx950 a 16-byte buffer store WITH an SGPR offset reads its upper data dwords late,
                // and hipcc 7.2 does not keep the next VALU write of those registers away from it (seen as 2.0 = the
                // sigmoid's "1 + e" in place of a mask value).
                // (mask tiles are fp16 C fragments: 512 B)
                constexpr unsigned kTile = kApack ? 1024u : 512u;
                const unsigned out_bytes = kApack ? (unsigned) units * kTile : (unsigned) g.ntiles * kTile;
                const char *out_mt = (const char *) g.out + (size_t) (valid ? mt : 0) * out_bytes;
                const unsigned tile_bytes = valid ? kTile : 0u;
                const frag_t *ab = abuf + (cur * kStageBlocks + m * NB) * 64;
                frag_t a[NB];
#pragma unroll
                for (int blk = 0; blk < NB; ++blk) a[blk] = ab[blk * 64 + lane];
#pragma unroll
                for (int q = 0; q < kLive; ++q) {
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
                            if (kSigmoid) x = head_sigmoid<P>(x);
                            if (kApack && nt * 16 + colq >= g.n_valid) x = 0.0f;
                            v[i] = x;
                        }
                        if (!kApack) {
                            buf_store_gi(make_rsrc(out_mt + (size_t) nt * 512, tile_bytes), lane * 8u, 0, __builtin_convertvector(v, f16x4));
                        } else {
                            uint16_t *sc = (uint16_t *) scratch;
#pragma unroll
                            for (int i = 0; i < 4; ++i) sc[P::off((lane >> 4) * 4 + i, j * 16 + colq)] = P::cvt(v[i]);
                        }
                    }
                    if (kApack) {
                        wave_lds_sync();
                        buf_store_frag(make_rsrc(out_mt + (size_t) unit[q] * 1024, tile_bytes), lane16,
                                       ((const frag_t *) scratch)[lane]);
                        wave_lds_sync();
                    }
                }
            }
            frag_t *an = abuf + (cur ^ 1) * kStageBlocks * 64;
#pragma unroll
            for (int i = 0; i < kFetch; ++i)
                if (wave + 8 * i < kStageBlocks) an[(wave + 8 * i) * 64 + lane] = stage[i];
            __syncthreads();
            cur ^= 1;
        }
    };
    int nlive = 0;
#pragma unroll
    for (int q = 0; q < UW; ++q) nlive += live[q] ? 1 : 0;
    // live units are a prefix (unit q of a wave is wave + 8 q); a wave without any still takes part in the staging
    if (UW >= 3 && nlive == 3)
        main_loop(std::integral_constant<int, (UW >= 3 ? 3 : UW)>{});
    else if (UW >= 2 && nlive == 2)
        main_loop(std::integral_constant<int, (UW >= 2 ? 2 : UW)>{});
    else if (nlive == 1)
        main_loop(std::integral_constant<int, 1>{});
    else
        main_loop(std::integral_constant<int, 0>{});
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
    // m-tiles in DESCENDING order: the A operand is the hidden sequence the recurrent kernel has just written step by step,
    // so its last steps are the ones still in the 256 MB memory-side cache
    const int stride = gridDim.x * 8;
    const int first = blockIdx.x * 8 + wave;
    int mt = first < g.mtiles ? first + ((g.mtiles - 1 - first) / stride) * stride : -1;
    frag_t an[NB];
    if (mt >= 0) {
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) an[blk] = a1p[((size_t) mt * NB + blk) * 64 + lane];
    }
    for (; mt >= 0; mt -= stride) {
        frag_t a[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) a[blk] = an[blk];
        if (mt - stride >= 0) {
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) an[blk] = a1p[((size_t) (mt - stride) * NB + blk) * 64 + lane];
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
                        const float x = head_sigmoid<PBF16>(acc[j][i] + bias[nt]);
                        v[i] = nt * 16 + colq < g.n_valid ? x : 0.0f;
                    }
                }
                if (g.pad_nb) {  // into the padding of an existing matrix (kns_kernels.h): the valid columns only, 2-byte elements
                    if (nt * 16 + colq < g.n_valid) {
                        uint16_t *dst = (uint16_t *) g.out + ((size_t) mt * g.pad_nb + g.pad_blk) * 64 * P::EPL;
#pragma unroll
                        for (int i = 0; i < 4; ++i) dst[P::off((lane >> 4) * 4 + i, g.pad_kk0 + nt * 16 + colq)] = P::cvt(v[i]);
                    }
                    continue;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) sc[P::off((lane >> 4) * 4 + i, j * 16 + colq)] = P::cvt(v[i]);
            }
            if (g.pad_nb) continue;
            wave_lds_sync();
            ((uint4 *) g.out)[((size_t) mt * (NT / 2) + pair) * 64 + lane] = ((const uint4 *) sc)[lane];
            wave_lds_sync();
        }
    }
}

// ---- weight-stationary form of the five-frame front-end (KNS-v1.1, bf16): out[t] = sum_tau feat[t - 4 + tau] . W_tau + b,
// K = 5 x 9 k-blocks, N = 18 n-tiles.  The 810 KiB weight image does not fit a CU, a third of its n-tiles does: the three
// workgroups of a triple (blocks b, b + 8, b + 16: one XCD, so the partners' reads of the feature tiles hit its L2) take six
// n-tiles = three A-packed output units each, one n-tile with all 45 of its k-blocks per wave (180 VGPRs).  A workgroup
// walks TIME for one tile of 16 streams: output frame t needs the feature tiles of frames t .. t + 4 of the buffer
// [context | call], so a ring of feature tiles in LDS takes one new tile per frame; kF5P frames are worked per barrier (with
// more than one, a fragment read from LDS feeds every frame's chain it belongs to: tile t0 + i is tap i - m of frame t0 + m).
// Every chain is k-ascending from a zero accumulator, bias after it: the same arithmetic as gemm_kernel, bit for bit.
// Measured (4 096 streams x 64 frames): ~205 us against 1 050 for gemm_kernel, which re-streams the weights per 32 rows; 1, 2, 3
// or 4 frames per barrier and a request distance of 1 or 2 stages all within 5 % -- the kernel is bound by the MFMA issue of
// the two SIMDs that carry two of the six waves (2 x 45 x 16 cycles per frame).
constexpr int kF5Waves = 6;
constexpr int kF5Taps = 5;
#ifndef F5_P
#define F5_P 1
#endif
#ifndef F5_RING
#define F5_RING 8
#endif
#ifndef F5_ABL
#define F5_ABL 0  // timing ablations (tools/front5_time.py; garbage out): 1 no feature prefetch, 2 no MFMAs, 4 no output stores
#endif
#ifndef F5_DEPTH
#define F5_DEPTH 2
#endif
constexpr int kF5Depth = F5_DEPTH;            // stages between the request for a feature tile and the stage that reads it
constexpr int kF5P = F5_P;                    // output frames per barrier
constexpr int kF5Ring = F5_RING;              // feature tiles in LDS: kF5P + 4 being read, kF5P being filled
constexpr int kF5TileBytes = PBF16::NBH * 1024;
constexpr int kF5OutBytes = kF5P * 3 * 1024;  // one parity of the output transposer: kF5P frames x 3 units
constexpr int kF5LdsBytes = kF5Ring * kF5TileBytes + 2 * kF5OutBytes;

__global__ __launch_bounds__(64 * kF5Waves) void gemm_front5_kernel(GemmArgs g, int mtb, int T, int seglen) {
    typedef PBF16 P;
    typedef P::frag_t frag_t;
    constexpr int NB = P::NBH;
    static_assert(kF5Ring >= 2 * kF5P + kF5Taps - 1, "ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int job = (b / 24) * 8 + (b & 7), c = (b >> 3) % 3;
    const int nseg = (T + seglen - 1) / seglen;
    if (job >= mtb * nseg) return;
    const int st = job % mtb, seg = job / mtb;
    const int t_begin = seg * seglen, t_end = min(T, t_begin + seglen);
    const int colq = lane & 15;
    const int nt = 6 * c + wave;

    frag_t wr[kF5Taps * NB];
    {
        const frag_t *w = (const frag_t *) g.w + (size_t) nt * (kF5Taps * NB) * 64 + lane;
#pragma unroll
        for (int k = 0; k < kF5Taps * NB; ++k) wr[k] = w[k * 64];
    }
    const float bias = g.bias[nt * 16 + colq];
    const bool pad = nt * 16 + colq >= g.n_valid;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the resident loads are complete before the loop (see kns_gruq.hip)

    // feature tile i of the buffer [context | call] for this stream tile; past the end: a descriptor of zero bytes (reads 0)
    const char *feat = (const char *) g.a1;
    const int ntile_i = T + kF5Taps - 1;
    auto tile_rsrc = [&](int i) {
        const bool ok = i < ntile_i;
        return make_rsrc(feat + ((size_t) (ok ? i : 0) * mtb + st) * kF5TileBytes, ok ? (unsigned) kF5TileBytes : 0u);
    };
    const unsigned lane16 = lane * 16u;
    frag_t *ring = (frag_t *) smem;
    // prologue: tiles t_begin .. t_begin + kF5P + 3 (what the first stage reads)
    for (int f = wave; f < (kF5P + kF5Taps - 1) * NB; f += kF5Waves) {
        const int j = f / NB, kb = f - j * NB;
        ring[(((t_begin + j) % kF5Ring) * NB + kb) * 64 + lane] = buf_load_frag(tile_rsrc(t_begin + j), lane16, kb * 1024u);
    }
    __syncthreads();

    constexpr int kFetch = (kF5P * NB + kF5Waves - 1) / kF5Waves;
    // Feature tiles are requested kF5Depth stages before the stage that reads them (a global load takes ~1.5 us here, a stage
    // ~1.4): they wait in registers, set (stage mod kF5Depth), and go into the ring at the end of the stage before theirs.
    frag_t stage[kF5Depth][kFetch];
    auto request = [&](auto set_tag, int tfirst) {  // the kF5P new tiles of the stage whose first frame is tfirst
        constexpr int kSet = decltype(set_tag)::value;
#pragma unroll
        for (int q = 0; q < kFetch; ++q) {
            const int f = wave + kF5Waves * q;
            const int j = f / NB, kb = f - j * NB;
            if (F5_ABL & 1)
                stage[kSet][q] = wr[q];
            else if (f < kF5P * NB)
                stage[kSet][q] = buf_load_frag(tile_rsrc(tfirst + kF5Taps - 1 + j), lane16, kb * 1024u);
        }
    };
    auto body = [&](auto set_req, auto set_put, int t0, int par) {
        request(set_req, t0 + kF5Depth * kF5P);
        f32x4 acc[kF5P];
#pragma unroll
        for (int m = 0; m < kF5P; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < kF5P + kF5Taps - 1; ++i) {
            const frag_t *tile = ring + (((t0 + i) % kF5Ring) * NB) * 64 + lane;
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
                const frag_t a = tile[kb * 64];
#pragma unroll
                for (int m = 0; m < kF5P; ++m)
                    if (i - m >= 0 && i - m < kF5Taps) {
                        if (F5_ABL & 2)
                            acc[m][0] += (float) a[0] * (float) wr[(i - m) * NB + kb][1];
                        else
                            acc[m] = P::mma(a, wr[(i - m) * NB + kb], acc[m]);
                    }
            }
        }
        char *outb = smem + kF5Ring * kF5TileBytes + par * kF5OutBytes;
#pragma unroll
        for (int m = 0; m < kF5P; ++m) {
            uint16_t *sc = (uint16_t *) (outb + (m * 3 + (wave >> 1)) * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = pad ? 0.0f : acc[m][i] + bias;
                sc[P::off((lane >> 4) * 4 + i, (wave & 1) * 16 + colq)] = P::cvt(x);
            }
        }
        // the next stage's new tiles (requested kF5Depth - 1 stages ago) into the ring: their slots were last read a stage ago
        constexpr int kPut = decltype(set_put)::value;
#pragma unroll
        for (int q = 0; q < kFetch; ++q) {
            const int f = wave + kF5Waves * q;
            const int j = f / NB, kb = f - j * NB;
            if (f < kF5P * NB) ring[(((t0 + kF5P + kF5Taps - 1 + j) % kF5Ring) * NB + kb) * 64 + lane] = stage[kPut][q];
        }
        __syncthreads();
        // finished output tiles: kF5P frames x 3 units (descriptor of zero bytes for a frame past the segment)
#pragma unroll
        for (int q = wave; q < kF5P * 3 && !(F5_ABL & 4); q += kF5Waves) {
            const int m = q / 3, ul = q - 3 * m;
            const int t = t0 + m;
            const bool ok = t < t_end;
            const char *dst = (const char *) g.out + (((size_t) (ok ? t : 0) * mtb + st) * (g.ntiles / 2) + 3 * c + ul) * 1024;
            buf_store_frag(make_rsrc(dst, ok ? 1024u : 0u), lane16, ((const frag_t *) (outb + (m * 3 + ul) * 1024))[lane]);
        }
    };
    static_assert(kF5Depth == 1 || kF5Depth == 2, "register sets alternate");
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, kF5Depth - 1> S1;
    if (kF5Depth == 2) request(S1{}, t_begin + kF5P);  // stage 1's tiles; stage 0's went into the ring in the prologue
    for (int t0 = t_begin; t0 < t_end; t0 += 2 * kF5P) {
        body(S0{}, S1{}, t0, 0);
        if (t0 + kF5P < t_end) body(S1{}, S0{}, t0 + kF5P, 1);
    }
}

// ---- the five-frame front-end of a ONE-frame call (or of a few frames: one workgroup triple per m-tile of stream-frames): the same
// split -- six n-tiles per workgroup, one n-tile with all 45 k-blocks per wave -- but nothing to walk: the wave's 45 weight
// fragments and the m-tile's five feature tiles (global -> LDS directly) are requested at once, one memory round trip, 45 MFMAs.
// (gemm_kernel needs ~25 us for this shape: it streams the 810 KiB through a four-deep queue per unit.)
constexpr int kF5tLdsBytes = kF5Taps * kF5TileBytes + 3 * 1024;

__global__ __launch_bounds__(64 * kF5Waves) void gemm_front5_t1_kernel(GemmArgs g) {
    typedef PBF16 P;
    typedef P::frag_t frag_t;
    constexpr int NB = P::NBH, NK = kF5Taps * NB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int mt = (b / 24) * 8 + (b & 7), c = (b >> 3) % 3;  // the triple of an m-tile on one XCD: blocks b, b + 8, b + 16
    if (mt >= g.mtiles) return;
    const int colq = lane & 15, nt = 6 * c + wave;
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    for (int i = wave; i < NK; i += kF5Waves) {
        const int tap = i / NB, kb = i - tap * NB;
        __builtin_amdgcn_global_load_lds(
            (gptr_t) ((const frag_t *) ((const char *) g.a1 + tap * g.tap_stride) + ((size_t) mt * NB + kb) * 64 + lane),
            (lptr_t) (smem + i * 1024), 16, 0, 0);
    }
    frag_t w[NK];
    {
        const frag_t *wp = (const frag_t *) g.w + (size_t) nt * NK * 64 + lane;
#pragma unroll
        for (int k = 0; k < NK; ++k) w[k] = wp[k * 64];
    }
    const float bias = g.bias[nt * 16 + colq];
    const bool pad = nt * 16 + colq >= g.n_valid;
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NK; ++k) acc = P::mma(((const frag_t *) smem)[k * 64 + lane], w[k], acc);
    char *outb = smem + kF5Taps * kF5TileBytes;
    {
        uint16_t *sc = (uint16_t *) (outb + (wave >> 1) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) sc[P::off((lane >> 4) * 4 + i, (wave & 1) * 16 + colq)] = P::cvt(pad ? 0.0f : acc[i] + bias);
    }
    __syncthreads();
    if (wave < 3)
        ((frag_t *) g.out)[(((size_t) mt * (g.ntiles / 2)) + 3 * c + wave) * 64 + lane] = ((const frag_t *) (outb + wave * 1024))[lane];
}

template <class P, int MT>
static void launch_gemm_mt(const GemmArgs &a, hipStream_t s) {
    const int nb = a.nb0 + a.taps * a.nb1;
    const int gx = ceil_div(a.mtiles, MT);
    const bool apack = a.out_kind == kOutAPlain || a.out_kind == kOutASigmoid;
    const int units = a.ntiles / (apack ? P::NPB : 1);
    // few stream-frames: split the n-tiles over more workgroups so the weight stream is spread over the CUs
    int gy = 1;
    if (gx < 256) gy = min(ceil_div(units, 4), max(1, 512 / gx));
    const size_t lds = (size_t) MT * nb * 1024 + 4 * MT * 1024;
    dim3 grid(gx, gy);
    auto go = [&](auto kernel) {
        if (lds > 48 * 1024) (void) hipFuncSetAttribute((const void *) kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);  // (the size varies per call)
        hipLaunchKernelGGL(kernel, grid, dim3(256), lds, s, a);
    };
    switch (a.out_kind) {
        case kOutGi: go(gemm_kernel<P, kOutGi, MT>); break;
        case kOutMask: go(gemm_kernel<P, kOutMask, MT>); break;
        case kOutAPlain: go(gemm_kernel<P, kOutAPlain, MT>); break;
        default: go(gemm_kernel<P, kOutASigmoid, MT>); break;
    }
}

template <class P>
static void launch_gemm_p(const GemmArgs &a, hipStream_t s) {
    // m-tiles per workgroup: four where the A tile (K up to 11 k-blocks) fits LDS four times; the five-frame front-end
    // (K = 45 / 85 k-blocks) leaves room for two / one
    const int nb = a.nb0 + a.taps * a.nb1;
    if (nb <= 30)
        launch_gemm_mt<P, 4>(a, s);
    else if (nb <= 60)
        launch_gemm_mt<P, 2>(a, s);
    else
        launch_gemm_mt<P, 1>(a, s);
}

void launch_gemm(const GemmArgs &a, hipStream_t s) {
    const bool no_ws = (a.dev & kDevGemmGeneric) != 0;  // A/B switch (developer build only)
    if (a.precision == kBf16 && a.out_kind == kOutGi && a.ntiles == kGateTiles && a.nb1 == PBF16::NBH && a.nb0 <= 2 && a.taps == 1 &&
        a.mtiles >= 256 && !no_ws) {
        // The weight-stationary kernel splits the m-tiles over 256 workgroup pairs: it takes the largest multiple of 256,
        // the remaining < 256 m-tiles (ragged stream counts) go through the generic kernel -- the same arithmetic, bit
        // for bit (tests/test_gpu_parity.py::test_alternative_kernels_give_identical_pcm).
        GemmArgs m = a;
        m.mtiles = a.mtiles / 256 * 256;
        // (a narrower grid -- GemmArgs::grid, whole groups of 16 workgroups -- walks more stages per workgroup pair; 256 = 2 x grid / 2 x
        // stage divides by every such grid)
        const int wgs = a.grid >= 16 && a.grid < 256 ? (a.grid / 16 >= 8 ? 128 : a.grid / 16 >= 4 ? 64 : a.grid / 16 >= 2 ? 32 : 16) : 256;
        const dim3 grid(wgs), block(64 * kWs2Waves);
        if (m.mtiles % 512 == 0) {  // four m-tiles per barrier
            if (a.nb0 == 0)  // (eight per barrier measured the same)
                hipLaunchKernelGGL((gemm_ws2_kernel<0, 4>), grid, block, 0, s, m);
            else if (a.nb0 == 1)
                hipLaunchKernelGGL((gemm_ws2_kernel<1, 4>), grid, block, 0, s, m);
            else
                hipLaunchKernelGGL((gemm_ws2_kernel<2, 4>), grid, block, 0, s, m);
        } else {
            if (a.nb0 == 0)
                hipLaunchKernelGGL((gemm_ws2_kernel<0, 2>), grid, block, 0, s, m);
            else if (a.nb0 == 1)
                hipLaunchKernelGGL((gemm_ws2_kernel<1, 2>), grid, block, 0, s, m);
            else
                hipLaunchKernelGGL((gemm_ws2_kernel<2, 2>), grid, block, 0, s, m);
        }
        if (a.mtiles > m.mtiles) {
            GemmArgs t = a;
            t.mtiles = a.mtiles - m.mtiles;
            if (a.a0) t.a0 = (const char *) a.a0 + (size_t) m.mtiles * a.nb0 * 1024;
            t.a1 = (const char *) a.a1 + (size_t) m.mtiles * a.nb1 * 1024;
            t.out = (char *) a.out + (size_t) m.mtiles * kGateTiles * 64 * sizeof(PBF16::gi_t);
            launch_gemm_p<PBF16>(t, s);
        }
        return;
    }
    const bool no_wsr = (a.dev & kDevGemmNoWsr) != 0;  // A/B switch (developer build only)
    if (a.precision == kBf16 && a.nb0 == 0 && a.nb1 == PBF16::NBH && a.taps == kF5Taps && a.out_kind == kOutAPlain && a.ntiles == 18 &&
        a.tap_stride % ((size_t) a.nb1 * 1024) == 0 && !no_wsr) {
        const int mtb = (int) (a.tap_stride / ((size_t) a.nb1 * 1024));  // m-tiles (of 16 streams) per frame
        const int T = mtb > 0 ? a.mtiles / mtb : 0;
        if (T >= 8 && a.mtiles == T * mtb) {
            // one workgroup triple per (stream tile, time segment); segments only when the stream tiles alone do not fill the chip
            int nseg = 1;
            while (mtb * nseg < 256 && T / (nseg * 2) >= 16) nseg *= 2;
            int seglen = (T + nseg - 1) / nseg;
            seglen = (seglen + kF5P - 1) / kF5P * kF5P;
            nseg = (T + seglen - 1) / seglen;
            const int jobs = mtb * nseg;
            allow_dynamic_lds(gemm_front5_kernel, kF5LdsBytes);
            hipLaunchKernelGGL(gemm_front5_kernel, dim3((jobs + 7) / 8 * 24), dim3(64 * kF5Waves), kF5LdsBytes, s, a, mtb, T, seglen);
            return;
        }
        if (a.mtiles <= 1024) {  // one frame (or a few) per call: no time to walk
            allow_dynamic_lds(gemm_front5_t1_kernel, kF5tLdsBytes);
            hipLaunchKernelGGL(gemm_front5_t1_kernel, dim3((a.mtiles + 7) / 8 * 24), dim3(64 * kF5Waves), kF5tLdsBytes, s, a);
            return;
        }
    }
    if (a.precision == kBf16 && a.nb0 == 0 && a.nb1 == PBF16::NBH && a.taps == 1 && a.mtiles >= 512 && !no_wsr) {
        const dim3 grid(a.grid >= 16 && a.grid < 256 ? a.grid : 256), block(512);
        if (a.out_kind == kOutAPlain && a.ntiles <= 32) {
            hipLaunchKernelGGL((gemm_wsr_kernel<kOutAPlain, 2>), grid, block, 0, s, a);
            return;
        }
        if (a.out_kind == kOutASigmoid && a.pad_nb && a.ntiles != 2) {
            launch_gemm_p<PBF16>(a, s);  // (into-the-padding output: the narrow-head kernel and the generic one write it)
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

}  // namespace kns
