// kns_gruq.hip -- a whole GRU layer (input GEMM + recurrent GEMM + gates) of ONE frame in one launch, fused
// over CU QUADS: the pre-activations x . W_ih never leave the CU, and a CU pulls a quarter of both weight matrices.
//
// Decomposition.  Four workgroups (same XCD by dispatch order: blocks b, b + 8, b + 16, b + 24) own four m-tiles = 64 streams.
// Workgroup c keeps the columns of W_ih and W_hh of hidden units 64 c .. 64 c + 63 (unit tiles 4 c .. 4 c + 3) in registers; unit
// tile 16 (units 256 .. 270) is served by workgroup c for m-tile c.
// (Rounds 3's multi-frame form of this decomposition -- phases, self-tagged granules through L2 -- measured 354-366 us per layer
// against 278 us for input GEMM + recurrent kernel, profiles/r03_quad_kernel.txt, and left the tree in round 4; DESIGN.md
// section 6 names the last commit that holds it.)
#include "kns_device.hpp"

#include <limits.h>

#include <type_traits>

namespace kns {

constexpr int kQWaves = 8;
constexpr int kQHsBytes = 9 * 1024;                   // one hidden-state operand image: 9 k-blocks in A-fragment order

// a unit tile of h in C-fragment order (lane: column colq, rows 4 q .. 4 q + 3) -> this lane's two packed words of the A
// operand: neighbouring lanes trade two values so that a word holds two consecutive k of one row
__device__ __forceinline__ void q_pack(const f32x4 &h, bool even, unsigned &w0, unsigned &w1) {
    const float s0 = even ? h[2] : h[0], s1 = even ? h[3] : h[1];
    const float r0 = lane_pair(s0), r1 = lane_pair(s1);
    const float lo0 = even ? h[0] : r0, hi0 = even ? r0 : h[2];
    const float lo1 = even ? h[1] : r1, hi1 = even ? r1 : h[3];
    w0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo0, hi0}, bf16x2));
    w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo1, hi1}, bf16x2));
}
__device__ __forceinline__ int q_tile_off(int u) { return (u >> 1) * 1024 + (u & 1) * 512; }

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// the bf16 configuration's gate arithmetic -- what every bf16 recurrent kernel does (kns_device.hpp)
__device__ __forceinline__ f32x4 q_gates(const f32x4 (&acc)[3], u32x2 vr, u32x2 vz, u32x2 vn, const f32x4 &hprev) {
    const unsigned pr[2] = {vr[0], vr[1]}, pz[2] = {vz[0], vz[1]}, pn[2] = {vn[0], vn[1]};
    return gate_block_bf16(pr, pz, pn, acc[0], acc[1], acc[2], hprev);
}

// ------------------------------------------------------------------------------------------------ one-step form
//
// One frame per call (T = 1): the same decomposition -- workgroup c of a quad holds the W_ih and W_hh columns of hidden units
// 64 c .. 64 c + 63 and serves unit tile 16 for m-tile c, so a CU pulls 300 KiB of weights per layer instead of ~740 -- but no
// recurrence, hence no exchange between workgroups, no rings and no phases.  A launch is bound by what a CU can pull through its
// vector-memory path (445 KiB at 64 B per clock: ~7 000 cycles) and by 2 x ~120 MFMAs per SIMD (~4 000), so the two are overlapped:
//   prologue   requests, in this order: x of the four m-tiles and unit tile 16's W_ih (global -> LDS directly), h_{-1} of the four
//              m-tiles (fp32 -> four operand images), then the first kQ1Ahead k-blocks of the wave's resident weights
//   barrier A  (staging complete; weights still streaming in)
//   k loop     k-block by k-block: request the weights of k-block k + kQ1Ahead, then the MFMAs of k-block k for all four blocks
//              x waves: x . W_ih (waves 0..2 also one gate of unit tile 16's, m-tile c);  h waves: [h ; 1 ; 1] . [W_hh ; b_hh] (waves
//              0..2 also one gate of unit tile 16's)
//   x waves    + b_ih -> fp16 in LDS
//   barrier B
//   h waves    gates of the four blocks -> hidden state (fp32) and hidden sequence (operand words); x wave 3: unit tile 16's
// Every chain is k-ascending: bit for bit the arithmetic of every other bf16 path.
constexpr int kQ1OffHs = 0;                            // [4] operand images of h_{-1}
constexpr int kQ1OffXs = 4 * kQHsBytes;                // [4][NBX] KiB (sized for NBX = 11)
constexpr int kQ1OffW16x = kQ1OffXs + 4 * 11 * 1024;   // [3 gates][NBX] KiB
constexpr int kQ1OffGi = kQ1OffW16x + 3 * 11 * 1024;   // [4 blocks][4 pairs][3 gates][64][8 B]
constexpr int kQ1OffGh16 = kQ1OffGi + 4 * 4 * 1536;    // [3][64][16 B]: h . W_hh of unit tile 16 (fp32)
constexpr int kQ1OffGi16 = kQ1OffGh16 + 3072;          // [3][64][8 B]: x . W_ih + b_ih of unit tile 16 (fp32 -> fp16)
// (kHead) the previous stage's layer-B hidden images [4][9] KiB are staged where the pre-activations go later
constexpr int kQ1OffYh = kQ1OffGi;
constexpr int kQ1Lds = kQ1OffYh + 4 * kQHsBytes;
static_assert(kQ1OffGi16 + 1536 <= kQ1Lds && kQ1Lds <= 160 * 1024, "LDS");
#ifndef Q1_AHEAD
#define Q1_AHEAD 2
#endif
constexpr int kQ1Ahead = Q1_AHEAD;  // k-blocks of weights in flight ahead of the MFMAs (a global load takes ~2 300 cycles, a k-block ~430)

// YB > 0: the previous stage's narrow head (YB k-blocks = 2 YB n-tiles wide) is computed here instead of read.  With NB0 == YB its
// output is the y part in front of x ([y_prev ; features]); with NB0 == 0 it is written INTO the features' last k-block, columns
// 1 ... yvalid of block 8 (kns_layout.h, kYPadMax: [features ; y_prev] sharing a k-block)
template <int NB0, int YB>
__global__ __launch_bounds__(64 * kQWaves, 2) void gru_quad1_kernel(GruQuadArgs g) {
    typedef bf16x8 frag_t;
    constexpr int NBX = 9 + NB0;
    constexpr bool kHead = YB > 0, kPad = YB > 0 && NB0 == 0;
    static_assert(YB == 0 || NB0 == YB || NB0 == 0, "a head feeds a y part in front of x, or the padding of x's last block");
    __shared__ __attribute__((aligned(16))) char smem[kQ1Lds];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int c = (bid >> 3) & 3, colq = lane & 15;
    const int qq = g.quad0 + (bid >> 5) * 8 + (bid & 7);
    if (qq >= (g.mtiles >> 2)) return;
    const int mt0 = 4 * qq;
    const bool even = (lane & 1) == 0;
    const int lane_off = ((lane >> 4) * 4 + (even ? 0 : 2) + 16 * ((colq & ~1) >> 3)) * 16 + ((colq & ~1) & 7) * 2;
    const int j = wave & 3, u = 4 * c + j;
    // developer stamps (KOALA_AMD_QUAD_DBG=<workgroup>, tools/t1_stamps.py): row [wave][0][0..5] + [wave][1][0]
    unsigned long long *dbg = (g.dbg && bid == g.dbg_block && lane == 0) ? g.dbg + (size_t) wave * 4 * 8 : nullptr;
    auto stamp = [&](int slot) {
        if (dbg) dbg[slot] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);

    // ---- staging requests (both roles).  x and unit tile 16's W_ih are copied as they are: global -> LDS without passing
    // registers (a lane's 16 bytes land at the wave-uniform LDS address + 16 lane).
    constexpr int kXF = (4 * NBX + kQWaves - 1) / kQWaves, kWF = (3 * NBX + kQWaves - 1) / kQWaves;
    constexpr int kHF = 4 * (kUnitTiles + 1) / kQWaves;  // 4 x 18 tile slots (17 + the zero half of k-block 8) = 9 per wave
    static_assert(4 * (kUnitTiles + 1) % kQWaves == 0, "whole tiles per wave");
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    // (kHead) the narrow head's operands first: the four hidden images of the previous stage's layer B -> LDS, and per wave the
    // weights of its chains -- 4 m-tiles x 2 NB0 n-tiles = 8 or 16 chains of 9 MFMAs over the 8 waves: chain ch = wave + 8 q is
    // (m-tile ch & 3, n-tile ch >> 2)
    constexpr int kCh = kHead ? YB : 1;
    frag_t yw[kCh][9];
    float ybias[kCh];
    if (kHead) {
        for (int i = wave; i < 4 * 9; i += kQWaves)
            __builtin_amdgcn_global_load_lds((gptr_t) ((const frag_t *) g.yh + ((size_t) (mt0 + i / 9) * 9 + i % 9) * 64 + lane),
                                             (lptr_t) (smem + kQ1OffYh + i * 1024), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < kCh; ++q) {
            const int nt = (wave + kQWaves * q) >> 2;
#pragma unroll
            for (int blk = 0; blk < 9; ++blk) yw[q][blk] = ((const frag_t *) g.yw)[((size_t) nt * 9 + blk) * 64 + lane];
            ybias[q] = g.yb[nt * 16 + colq];
        }
    }
#pragma unroll
    for (int q = 0; q < kXF; ++q) {
        const int i = wave + kQWaves * q;
        const int m = i / NBX, k = i % NBX;
        if (i < 4 * NBX && !(kHead && k < NB0))
            __builtin_amdgcn_global_load_lds(
                (gptr_t) ((k < NB0 ? (const frag_t *) g.a0 + ((size_t) (mt0 + m) * NB0 + k) * 64
                                   : (const frag_t *) g.a1 + ((size_t) (mt0 + m) * 9 + (k - NB0)) * 64) + lane),
                (lptr_t) (smem + kQ1OffXs + (m * 11 + k) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < kWF; ++q) {
        const int i = wave + kQWaves * q;
#ifndef Q1_T_NOFETCH
        if (i < 3 * NBX)
#else
        if (false)
#endif
            __builtin_amdgcn_global_load_lds((gptr_t) ((const frag_t *) g.wih + ((size_t) 48 * NBX + i) * 64 + lane),
                                             (lptr_t) (smem + kQ1OffW16x + i * 1024), 16, 0, 0);
    }
    f32x4 sh[kHF];
#pragma unroll
    for (int q = 0; q < kHF; ++q) {
        const int idx = wave + kQWaves * q;
        const int m = idx / (kUnitTiles + 1), t = idx % (kUnitTiles + 1);
        // (slot 17 loads tile 16 again and is zeroed when it is used: no branch around a load, see kns_gru.hip)
        sh[q] = ((const f32x4 *) g.hstate_in)[((size_t) (mt0 + m) * kUnitTiles + (t < kUnitTiles ? t : kUnitTiles - 1)) * 64 + lane];
    }
    // staging -> LDS.  The 18 tile slots of an image cover all of its 9 KiB (slot 17 = the zero half of k-block 8).
    auto images = [&]() {
#pragma unroll
        for (int q = 0; q < kHF; ++q) {
            const int idx = wave + kQWaves * q;
            const int m = idx / (kUnitTiles + 1), t = idx % (kUnitTiles + 1);
            unsigned w0, w1;
            q_pack(sh[q], even, w0, w1);
            // k = 271 and k = 272 of the operand are the constant 1 that multiplies the two bias rows of the packed W_hh
            // (kns_layout.h, kBiasK0): k = 271 is column 15 of tile 16 -- the high half of the words of lanes colq 14 / 15 --, k = 272
            // column 0 of the otherwise empty slot 17 -- the low half of the words of lanes colq 0 / 1
            // (as masks, not conditions: `t` is wave-uniform at run time, and hipcc turns such conditions into branches around the
            // stores, whose merged waits serialise the prologue's one round trip)
            const unsigned m17 = 0u - (unsigned) (t == kUnitTiles), m16 = (0u - (unsigned) ((t == kUnitTiles - 1) & (colq >= 14))) & 0xffff0000u;
            const unsigned one17 = colq < 2 ? kBf16One : 0u;
            w0 = (w0 & ~m17 & ~m16) | (one17 & m17) | ((kBf16One << 16) & m16);
            w1 = (w1 & ~m17 & ~m16) | (one17 & m17) | ((kBf16One << 16) & m16);
            char *img = smem + kQ1OffHs + m * kQHsBytes + q_tile_off(t) + lane_off;
            *(unsigned *) img = w0;
            *(unsigned *) (img + 16) = w1;
        }
    };
    // (kHead) y_prev = sigmoid(h_B . W_head + b_head), columns >= yvalid zero, rounded to the operand type: what gemm_head_kernel /
    // gemm_kernel<kOutASigmoid> store, written into the staged x operand instead (k-block nt / 2 of m-tile m, column half nt & 1)
    auto head = [&]() {
        if (!kHead) return;
        __syncthreads();  // the hidden images are in LDS (every wave's requests)
#pragma unroll
        for (int q = 0; q < kCh; ++q) {
            const int ch = wave + kQWaves * q, m = ch & 3, nt = ch >> 2;
            const frag_t *ya = (const frag_t *) (smem + kQ1OffYh + m * kQHsBytes) + lane;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int blk = 0; blk < 9; ++blk) acc = PBF16::mma(ya[blk * 64], yw[q][blk], acc);
            const bool pad = nt * 16 + colq >= g.yvalid;
            if (kPad) {  // the valid columns into the staged features' last block (k = 1 + column; the rest of the block stays)
                uint16_t *sc = (uint16_t *) (smem + kQ1OffXs + (m * 11 + NBX - 1) * 1024);
                if (!pad) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        sc[PBF16::off((lane >> 4) * 4 + i, 1 + nt * 16 + colq)] = PBF16::cvt(head_sigmoid<PBF16>(acc[i] + ybias[q]));
                }
                continue;
            }
            uint16_t *sc = (uint16_t *) (smem + kQ1OffXs + (m * 11 + (nt >> 1)) * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = pad ? 0.0f : head_sigmoid<PBF16>(acc[i] + ybias[q]);
                sc[PBF16::off((lane >> 4) * 4 + i, (nt & 1) * 16 + colq)] = PBF16::cvt(x);
            }
        }
    };
    // one tile of h_0 of m-tile m: fp32 state, operand words of the hidden sequence
    auto emit = [&](int m, int tile, const f32x4 &hnew) {
        ((f32x4 *) g.hstate_out)[((size_t) (mt0 + m) * kUnitTiles + tile) * 64 + lane] = hnew;
        unsigned w0, w1;
        q_pack(hnew, even, w0, w1);
        char *hs = (char *) g.hseq + (size_t) (mt0 + m) * kQHsBytes + q_tile_off(tile) + lane_off;
        *(unsigned *) hs = w0;
        *(unsigned *) (hs + 16) = w1;
    };

    if (wave < 4) {
        // ---------------------------------------------------------------------------------------- x waves
        const frag_t *wih = (const frag_t *) g.wih + (size_t) (u * 3) * NBX * 64 + lane;  // [gate][k-block] fragments of tile u
        frag_t w[3][NBX];
        auto request = [&](const int blk) {
#ifdef Q1_T_NOFETCH  // TIMING ONLY (garbage): as if the weights were already on the CU -- the bound on cross-call residency
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) asm volatile("" : "=v"(w[gt][blk]));
            (void) wih;
#else
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) w[gt][blk] = wih[(gt * NBX + blk) * 64];
#endif
        };
#pragma unroll
        for (int blk = 0; blk < kQ1Ahead && blk < NBX; ++blk) request(blk);
        const float b0 = g.bih[(u * 3 + 0) * 16 + colq], b1 = g.bih[(u * 3 + 1) * 16 + colq], b2 = g.bih[(u * 3 + 2) * 16 + colq];
        const float b16 = g.bih[(16 * 3 + (j < 3 ? j : 0)) * 16 + colq];
        f32x4 hp16 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (j == 3) hp16 = ((const f32x4 *) g.hstate_in)[((size_t) (mt0 + c) * kUnitTiles + 16) * 64 + lane];
        stamp(1);
        images();
        head();
        stamp(2);
        __syncthreads();  // barrier A
        stamp(3);
        f32x4 acc[4][3], a16 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) acc[m][gt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const frag_t *xs = (const frag_t *) (smem + kQ1OffXs) + lane;
        const frag_t *w16x = (const frag_t *) (smem + kQ1OffW16x) + (j < 3 ? j : 0) * NBX * 64 + lane;
#pragma unroll
        for (int blk = 0; blk < NBX; ++blk) {
            if (blk + kQ1Ahead < NBX) request(blk + kQ1Ahead);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const frag_t a = xs[(m * 11 + blk) * 64];
#pragma unroll
                for (int gt = 0; gt < 3; ++gt) acc[m][gt] = PBF16::mma(a, w[gt][blk], acc[m][gt]);
            }
            if (j < 3) a16 = PBF16::mma(xs[(c * 11 + blk) * 64], w16x[blk * 64], a16);  // one gate of unit tile 16, m-tile c
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(4);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            char *ring = smem + kQ1OffGi + ((m * 4 + j) * 3) * 512 + lane * 8;
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) {
                const float b = gt == 0 ? b0 : gt == 1 ? b1 : b2;
                f32x4 v = acc[m][gt];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] + b;
                *(f16x4 *) (ring + gt * 512) = PBF16::to_gi(v);
            }
        }
        if (j < 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a16[i] = a16[i] + b16;
            *(f16x4 *) (smem + kQ1OffGi16 + j * 512 + lane * 8) = PBF16::to_gi(a16);
        }
        __syncthreads();  // barrier B
        stamp(5);
        if (j == 3) {  // unit tile 16 of m-tile c
            const char *gh = smem + kQ1OffGh16 + lane * 16;
            f32x4 gacc[3];
            gacc[0] = *(const f32x4 *) gh;
            gacc[1] = *(const f32x4 *) (gh + 1024);
            gacc[2] = *(const f32x4 *) (gh + 2048);
            const char *gx = smem + kQ1OffGi16 + lane * 8;
            emit(c, 16, q_gates(gacc, *(const u32x2 *) gx, *(const u32x2 *) (gx + 512), *(const u32x2 *) (gx + 1024), hp16));
        }
        stamp(8);
        return;
    }

    // -------------------------------------------------------------------------------------------- h waves
    const frag_t *whh = (const frag_t *) g.whh + (size_t) (u * 3) * 9 * 64 + lane;  // [gate][k-block] fragments of tile u
    const frag_t *whh16 = (const frag_t *) g.whh + (size_t) (16 * 3 + (j < 3 ? j : 0)) * 9 * 64 + lane;
    frag_t w[9][3], w16[9];
    auto request = [&](const int blk) {
#ifdef Q1_T_NOFETCH
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) asm volatile("" : "=v"(w[blk][gt]));
        asm volatile("" : "=v"(w16[blk]));
        (void) whh, (void) whh16;
#else
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) w[blk][gt] = whh[(gt * 9 + blk) * 64];
        w16[blk] = whh16[blk * 64];  // (used by waves 0..2; requested by all four: no conditional load in the stream)
#endif
    };
#pragma unroll
    for (int blk = 0; blk < kQ1Ahead && blk < 9; ++blk) request(blk);
    stamp(1);
    images();
    head();
    stamp(2);
    __syncthreads();  // barrier A
    stamp(3);
    f32x4 acc[4][3], a16 = f32x4{0.f, 0.f, 0.f, 0.f};  // (b_hh rides in the operand: two rows of the packed W_hh against h's constant 1)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) acc[m][gt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const frag_t *hs = (const frag_t *) (smem + kQ1OffHs) + lane;
    // the previous state of the tiles this wave finishes after barrier B: requested once the last weights are (k-block 9 - kQ1Ahead),
    // so that it is there when the gates start and waits in registers for a few k-blocks only
    f32x4 hp[4];
#pragma unroll
    for (int blk = 0; blk < 9; ++blk) {
        if (blk + kQ1Ahead < 9) request(blk + kQ1Ahead);
        if (blk == 9 - kQ1Ahead) {
#pragma unroll
            for (int m = 0; m < 4; ++m) hp[m] = ((const f32x4 *) g.hstate_in)[((size_t) (mt0 + m) * kUnitTiles + u) * 64 + lane];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const frag_t a = hs[(m * 9 + blk) * 64];
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) acc[m][gt] = PBF16::mma(a, w[blk][gt], acc[m][gt]);
        }
        if (j < 3) a16 = PBF16::mma(hs[(c * 9 + blk) * 64], w16[blk], a16);  // one gate of unit tile 16, m-tile c
        __builtin_amdgcn_sched_barrier(0);
    }
    if (j < 3) *(f32x4 *) (smem + kQ1OffGh16 + j * 1024 + lane * 16) = a16;
    stamp(4);
    __syncthreads();  // barrier B
    stamp(5);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const char *ring = smem + kQ1OffGi + ((m * 4 + j) * 3) * 512 + lane * 8;
        const u32x2 pr = *(const u32x2 *) ring, pz = *(const u32x2 *) (ring + 512), pn = *(const u32x2 *) (ring + 1024);
        emit(m, u, q_gates(acc[m], pr, pz, pn, hp[m]));
    }
    stamp(8);
}

bool gru_quad_supported(int precision, int mtiles, int nb0) {
    return precision == kBf16 && mtiles >= 4 && mtiles % 4 == 0 && nb0 >= 0 && nb0 <= 2;
}

void launch_gru_quad(const GruQuadArgs &a, hipStream_t s) {
    const int nquads = a.mtiles / 4;
    for (int q0 = 0; q0 < nquads; q0 += 64) {  // 64 quads = 256 workgroups = one per CU
        GruQuadArgs g = a;
        g.quad0 = q0;
        const int n = nquads - q0 < 64 ? nquads - q0 : 64;
        const dim3 grid((n + 7) / 8 * 32), block(64 * kQWaves);  // 32 workgroups = 8 quads, one per XCD
        const bool head = a.yw != nullptr;  // the previous stage's narrow head rides along
        if (a.nb0 == 0 && head)  // ... into the padding of the features' last block (one k-block = two n-tiles of head)
            hipLaunchKernelGGL((gru_quad1_kernel<0, 1>), grid, block, 0, s, g);
        else if (a.nb0 == 0)
            hipLaunchKernelGGL((gru_quad1_kernel<0, 0>), grid, block, 0, s, g);
        else if (a.nb0 == 1 && head)
            hipLaunchKernelGGL((gru_quad1_kernel<1, 1>), grid, block, 0, s, g);
        else if (a.nb0 == 1)
            hipLaunchKernelGGL((gru_quad1_kernel<1, 0>), grid, block, 0, s, g);
        else if (head)
            hipLaunchKernelGGL((gru_quad1_kernel<2, 2>), grid, block, 0, s, g);
        else
            hipLaunchKernelGGL((gru_quad1_kernel<2, 0>), grid, block, 0, s, g);
    }
}

}  // namespace kns
