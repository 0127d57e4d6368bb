// kns_gruq.hip -- a whole GRU layer (input GEMM + recurrent GEMM + gates) over T frames in one launch,
// fused over CU QUADS: the pre-activations x . W_ih never leave the CU (no `gi` round trip through HBM), and a CU works on
// four independent m-tiles per step instead of being one in-order chain over a single one.
//
// Decomposition.  Four workgroups (same XCD by dispatch order: blocks b, b + 8, b + 16, b + 24) own four m-tiles = 64 streams.
// Workgroup c keeps, for ALL T steps, the columns of W_ih and W_hh of hidden units 64 c .. 64 c + 63 (unit tiles 4 c .. 4 c + 3
// = k-blocks 2 c, 2 c + 1 of the hidden state) in registers; unit tile 16 (units 256 .. 270) is kept by every workgroup (in
// LDS) and served by workgroup c for m-tile c.  A "block" is (step t, m-tile m), numbered b = 4 t + m.
// The workgroup advances in PHASES separated by one s_barrier; in phase p
//   x waves 0..3  (wave j: W_ih of unit tile 4 c + j, 3 x NBX fragments in registers) compute x . W_ih of block p + 1 -- it does
//                 not depend on h --, add b_ih, round to fp16 (the storage type of the two-kernel form's `gi`: the arithmetic
//                 is the same bit for bit) and leave it in LDS for their partner; they stage the x operand two blocks ahead
//                 (each wave a quarter of the k-blocks); waves 0..2 FILE the other workgroups' tiles of the h that block
//                 p + 1 reads into its LDS image (requested from the exchange buffer a phase earlier, tags checked now);
//                 wave 3 serves unit tile 16: its input projection where m = c (weights from LDS) and its gate math;
//   h waves 4..7  (wave j: W_hh of unit tile 4 c + j, 27 fragments in registers) do block p: 27 MFMAs against the image of
//                 h_{t-1} (waves 0..2 one gate of tile 16 as well where m = c), gates, and send h_t of their tile to the other
//                 three workgroups (granules) and to the hidden sequence in HBM; the tile enters the LDS image at the start
//                 of the next phase, when nobody reads that image any more.
// One wave of each kind shares a SIMD: the x wave's MFMAs run under its partner's gate arithmetic.
// Hand-off between workgroups (MI355X guide, "R2"): a lane's 16-byte store carries two self-tagged 8-byte granules {tag, 2 x
// bf16}; tag = launch serial << 12 | step + 1.  No flag, no fence, no drain: the data is its own flag; loads bypass L1 (sc1).
// Slots alternate with the step's parity; a slot is rewritten only after every consumer has used it (by data flow: producing
// h_{t+2} needs all of h_{t+1}, which needed every consumer's h_t to be complete).  A block's output is needed four phases
// later; it is requested two phases and checked three phases after it was produced.
// Every wait for another workgroup is bounded: on overrun the wave records a code in GruQuadArgs::err, stops polling and runs
// the remaining phases on whatever it has (the call's results are invalid and the host is told so).
// (The first form of this kernel synchronised its waves through counters in LDS instead of barriers: bit-identical results,
// 375 us per layer at the bench shape against 278 us for the two-kernel form -- its synchronisation skeleton alone, with
// MFMAs, gate arithmetic and memory traffic compiled out, took 194 us.  History: commit "fused kernel (flag-synchronised form)".)
#include "kns_device.hpp"

#include <limits.h>

#include <type_traits>

namespace kns {

constexpr int kQWaves = 8;
#ifndef KQ_QA
#define KQ_QA 4
#endif
constexpr int kQA = KQ_QA;                            // operand fragments in flight (LDS -> register) in the MFMA loops
constexpr int kQHsBytes = 9 * 1024;                   // one hidden-state operand image: 9 k-blocks in A-fragment order
constexpr int kQOffHs = 0;                            // [4 m-tiles]: h_{t-1} while a block reads it, then h_t tile by tile
constexpr int kQOffXs = 4 * kQHsBytes;                // [3][NBX] KiB (sized for NBX = 11): x of three consecutive blocks
constexpr int kQOffGi = kQOffXs + 3 * 11 * 1024;      // [2][4 pairs][3 gates][64 lanes][8 B]: fp16 pre-activations
constexpr int kQOffGh16 = kQOffGi + 2 * 4 * 1536;     // [2][3][64][16 B]  unit tile 16: fp32 recurrent accumulators
constexpr int kQOffW16x = kQOffGh16 + 2 * 3072;       // [3 gates][NBX] KiB  unit tile 16's W_ih (sized for NBX = 11)
constexpr int kQOffW16h = kQOffW16x + 3 * 11 * 1024;  // [3 gates][9] KiB    unit tile 16's W_hh
constexpr int kQLds = kQOffW16h + 27 * 1024;

// developer ablations (timing experiments, results are garbage): 1 no remote gather, 2 no x staging loads, 4 no global stores
// of h, 8 no gate math, 16 no MFMAs
#ifndef KQ_ABL
#define KQ_ABL 0
#endif
constexpr int kQGatherLimit = 1 << 18;   // polls of the exchange buffer (~0.3 s)

struct QCtx {
    unsigned long long *dbg;  // this wave's stamp rows ([block][8]) or null
    char *smem;
    unsigned *err;
    int lane, colq;
    int c;                // workgroup's place in its quad
    int mt0;              // first m-tile of the quad
    int T, mtiles, NB;    // NB = 4 T blocks
    unsigned tag_base;
    bool even;            // lane holds an even column of its unit tile
    int lane_off;         // byte offset of this lane's first packed word inside a unit tile's half k-block (second: + 16)
};

// developer instrumentation: s_memtime of one workgroup's waves at fixed points of every phase (null in production)
__device__ __forceinline__ void q_stamp(const QCtx &cx, int b, int slot) {
    if (cx.dbg && cx.lane == 0 && b >= 0 && b < cx.NB) cx.dbg[b * 8 + slot] = __builtin_amdgcn_s_memtime();
}
__device__ __forceinline__ void q_note(const QCtx &cx, int b, int slot, unsigned long long v) {
    if (cx.dbg && cx.lane == 0 && b >= 0 && b < cx.NB) cx.dbg[b * 8 + slot] = v;
}

// every LDS access of this wave has been performed; then the workgroup's barrier
__device__ __forceinline__ void q_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// the bf16 configuration's gate arithmetic -- what every bf16 recurrent kernel does (kns_device.hpp); the accumulators start
// from b_hh
__device__ __forceinline__ f32x4 q_gates(const f32x4 (&acc)[3], u32x2 vr, u32x2 vz, u32x2 vn, const f32x4 &hprev) {
    if (KQ_ABL & 8) return acc[0] + acc[1] + acc[2] + hprev;
    const unsigned pr[2] = {vr[0], vr[1]}, pz[2] = {vz[0], vz[1]}, pn[2] = {vn[0], vn[1]};
    return gate_block_bf16(pr, pz, pn, acc[0], acc[1], acc[2], hprev);
}

// h_t of unit tile u, m-tile (local) m, as this lane's two packed operand words: to the other workgroups as granules and to the
// hidden sequence in HBM.  The caller writes the words into the m-tile's LDS image (q_image_write) once nobody reads it.
__device__ __forceinline__ void q_publish(const GruQuadArgs &g, const QCtx &cx, int t, int m, int u, unsigned w0, unsigned w1) {
    if (KQ_ABL & 4) return;
    const int mt = cx.mt0 + m;
    const unsigned tag = cx.tag_base | (unsigned) (t + 1);
    const __amdgpu_buffer_rsrc_t gr =
        make_rsrc((char *) g.xchg + (((size_t) mt * 2 + (t & 1)) * 17 + u) * 1024, 1024);
#ifndef KQ_STORE_AUX
#define KQ_STORE_AUX 16
#endif
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{tag, w0, tag, w1}, gr, cx.lane * 16u, 0, KQ_STORE_AUX /* 16 = sc1: write through */);
    const __amdgpu_buffer_rsrc_t hr =
        make_rsrc((char *) g.hseq + ((size_t) t * cx.mtiles + mt) * kQHsBytes + q_tile_off(u), 512);
    __builtin_amdgcn_raw_buffer_store_b32(w0, hr, (unsigned) cx.lane_off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(w1, hr, (unsigned) cx.lane_off + 16u, 0, 0);
}
__device__ __forceinline__ void q_image_write(const QCtx &cx, int m, int u, unsigned w0, unsigned w1) {
    char *img = cx.smem + kQOffHs + m * kQHsBytes + q_tile_off(u) + cx.lane_off;
    *(unsigned *) img = w0;
    *(unsigned *) (img + 16) = w1;
}

// ---- remote tiles of h: requested from the exchange buffer (sc1: from L2, never this CU's L1), checked, filed into an image
struct QGather {
    u32x4 gr[4];
    u32x4 g16;  // unit tile 16 (x wave 0 only)
};
// the h that block bq READS (step (bq >> 2) - 1 of m-tile bq & 3), tiles tile0 + i * tstep
__device__ __forceinline__ void q_gather_load(const GruQuadArgs &g, const QCtx &cx, int bq, int tile0, int tstep, QGather &q) {
    const int ts = (bq >> 2) - 1, mq = bq & 3;
    const size_t slot = ((size_t) (cx.mt0 + mq) * 2 + (ts & 1)) * 17;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __amdgpu_buffer_rsrc_t r = make_rsrc((const char *) g.xchg + (slot + tile0 + i * tstep) * 1024, 1024);
        q.gr[i] = __builtin_amdgcn_raw_buffer_load_b128(r, cx.lane * 16u, 0, 16);
    }
}
__device__ __forceinline__ bool q_gather_valid(const QCtx &cx, int bq, const QGather &q) {
    const unsigned tag = cx.tag_base | (unsigned) (bq >> 2);  // step (bq >> 2) - 1, + 1
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) ok = ok && q.gr[i][0] == tag && q.gr[i][2] == tag;
    return __builtin_amdgcn_ballot_w64(!ok) == 0;
}
__device__ __forceinline__ void q_gather_file(const QCtx &cx, int bq, int tile0, int tstep, const QGather &q) {
    char *img = cx.smem + kQOffHs + (bq & 3) * kQHsBytes + cx.lane_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *(unsigned *) (img + q_tile_off(tile0 + i * tstep)) = q.gr[i][1];
        *(unsigned *) (img + q_tile_off(tile0 + i * tstep) + 16) = q.gr[i][3];
    }
}

// ------------------------------------------------------------------------------------------------ x waves

// 3 x NBX MFMAs of one block against the wave's register-resident W_ih tile
template <int NBX>
__device__ __forceinline__ void q_x_mma(f32x4 (&acc)[3], const bf16x8 *xa, const bf16x8 (&w)[3][NBX], int lane) {
    bf16x8 qa[kQA];
#pragma unroll
    for (int p = 0; p < kQA; ++p) qa[p] = xa[p * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int blk = 0; blk < NBX; ++blk) {
        const bf16x8 a = qa[blk % kQA];
#pragma unroll
        for (int gt = 0; gt < 3; ++gt)
            if (!(KQ_ABL & 16)) acc[gt] = PBF16::mma(a, w[gt][blk], acc[gt]);
        if (blk + kQA < NBX) qa[blk % kQA] = xa[(blk + kQA) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
    }
}
// the same for unit tile 16, whose weights live in LDS ([gate][k-block] fragments): a second pass over the staged x operand,
// rolled so that it needs a handful of registers
template <int NBX>
__device__ __forceinline__ void q_x_mma16(f32x4 (&acc)[3], const bf16x8 *xa, const bf16x8 *w16, int lane) {
#pragma nounroll
    for (int blk = 0; blk < NBX; ++blk) {
        const bf16x8 a = xa[blk * 64 + lane];
#pragma unroll
        for (int gt = 0; gt < 3; ++gt)
            if (!(KQ_ABL & 16)) acc[gt] = PBF16::mma(a, w16[(gt * NBX + blk) * 64 + lane], acc[gt]);
    }
}

template <int NB0>
__device__ __forceinline__ void q_x_wave(const GruQuadArgs &g, const QCtx &cx, const int j) {
    typedef bf16x8 frag_t;
    constexpr int NBX = 9 + NB0;
    const int lane = cx.lane, c = cx.c, u = 4 * c + j;
    const frag_t *wih = (const frag_t *) g.wih;
    frag_t w[3][NBX];
    float bi[3];
#pragma unroll
    for (int gt = 0; gt < 3; ++gt) {
#pragma unroll
        for (int blk = 0; blk < NBX; ++blk) w[gt][blk] = wih[((size_t) (u * 3 + gt) * NBX + blk) * 64 + lane];
        bi[gt] = g.bih[(u * 3 + gt) * 16 + cx.colq];
    }
    const frag_t *w16 = (const frag_t *) (cx.smem + kQOffW16x);
    // (wave 3) unit tile 16: biases, fp32 state of m-tile c
    const float b16r = g.bih[(16 * 3 + 0) * 16 + cx.colq], b16z = g.bih[(16 * 3 + 1) * 16 + cx.colq],
                b16n = g.bih[(16 * 3 + 2) * 16 + cx.colq];
    f32x4 h16 = ((const f32x4 *) g.hstate_in)[((size_t) (cx.mt0 + c) * kUnitTiles + 16) * 64 + lane];
    // The resident weights have arrived before the loop is entered -- said explicitly: hipcc's wait-count pass otherwise merges
    // "still in flight" from the loop's entry edge into the loop header and makes every phase's MFMAs wait for the
    // vector-memory operations of the phase before (vmcnt counts them all).
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)

    // this wave's share of a block's x operand: k-blocks j, j + 4, j + 8 (clamped: an extra copy of the last one is harmless)
    const int p0 = j, p1 = j + 4, p2 = j + 8 < NBX ? j + 8 : NBX - 1;
    auto piece = [&](int blk, int i) -> const frag_t * {
        const size_t mtg = (size_t) (blk >> 2) * cx.mtiles + cx.mt0 + (blk & 3);
        if (NB0 > 0 && i < NB0) return (const frag_t *) g.a0 + (mtg * NB0 + i) * 64 + lane;
        return (const frag_t *) g.a1 + (mtg * 9 + (i - NB0)) * 64 + lane;
    };
    frag_t st0, st1, st2;
    auto stage_load = [&](int blk) {
        const int bn = blk < cx.NB ? blk : cx.NB - 1;
        if (KQ_ABL & 2) {
            st0 = st1 = st2 = w[0][0];
            return;
        }
        st0 = *piece(bn, p0);
        st1 = *piece(bn, p1);
        st2 = *piece(bn, p2);
    };
    auto stage_write = [&](int blk) {
        frag_t *xn = (frag_t *) (cx.smem + kQOffXs + (blk % 3) * NBX * 1024);
        xn[p0 * 64 + lane] = st0;
        xn[p1 * 64 + lane] = st1;
        xn[p2 * 64 + lane] = st2;
    };

    u32x2 gi16[3] = {u32x2{0, 0}, u32x2{0, 0}, u32x2{0, 0}};  // (wave 3) unit tile 16's fp16 pre-activations of the current step
    // x . W_ih of block q -> the ring slot its partner reads in phase q (wave 3: and unit tile 16's where the block's m = c)
    auto project = [&](const int q) {
        const frag_t *xa = (const frag_t *) (cx.smem + kQOffXs + (q % 3) * NBX * 1024);
        f32x4 acc[3];
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
        q_x_mma<NBX>(acc, xa, w, lane);
        char *ring = cx.smem + kQOffGi + (((q & 1) * 4 + j) * 3) * 512 + lane * 8;
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) {
            f32x4 v = acc[gt];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = v[i] + bi[gt];
            *(f16x4 *) (ring + gt * 512) = PBF16::to_gi(v);
        }
        if (j == 3 && (q & 3) == c) {
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
            q_x_mma16<NBX>(acc, xa, w16, lane);
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) {
                const float b16 = gt == 0 ? b16r : gt == 1 ? b16z : b16n;
                f32x4 v = acc[gt];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] + b16;
                gi16[gt] = __builtin_bit_cast(u32x2, PBF16::to_gi(v));
            }
        }
    };

    // remote tiles of h: wave j < 3 the four tiles of workgroup (c + 1 + j) & 3, wave 0 also unit tile 16 from the workgroup
    // that serves it for the m-tile (nothing to fetch when that is this workgroup)
    const int rq = (c + 1 + j) & 3;
    QGather q = {};
    bool dead = false;  // gave up on the exchange buffer: no more polling, the call is reported as failed
    auto request = [&](int bq) {  // what h-block bq reads (redirected to something harmless where there is nothing to get)
        if ((KQ_ABL & 1) || j == 3) return;
        const int br = bq >= 4 && bq < cx.NB ? bq : 4 + (bq & 3);
        q_gather_load(g, cx, br, 4 * rq, 1, q);
        const size_t slot = ((size_t) (cx.mt0 + (br & 3)) * 2 + (((br >> 2) - 1) & 1)) * 17 + 16;
        q.g16 = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc((const char *) g.xchg + slot * 1024, 1024), lane * 16u, 0, 16);
    };
    auto file = [&](int bq) {  // requested a phase ago
        if (j == 3 || bq < 4 || bq >= cx.NB) return;
        const bool real16 = j == 0 && (bq & 3) != c;
        int spins = 0;
        if (!(KQ_ABL & 1) && !dead) {
            const unsigned tag = cx.tag_base | (unsigned) (bq >> 2);
            while (!(q_gather_valid(cx, bq, q) &&
                     (!real16 || __builtin_amdgcn_ballot_w64(q.g16[0] != tag || q.g16[2] != tag) == 0))) {
                if (++spins > kQGatherLimit) {
                    if (lane == 0) atomicCAS(cx.err, 0u, 0x30000000u | (unsigned) (j << 24) | (unsigned) bq);
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                asm volatile("" ::: "memory");
                request(bq);
            }
        }
        q_gather_file(cx, bq, 4 * rq, 1, q);
        if (real16) {
            char *img = cx.smem + kQOffHs + (bq & 3) * kQHsBytes + cx.lane_off + q_tile_off(16);
            *(unsigned *) img = q.g16[1];
            *(unsigned *) (img + 16) = q.g16[3];
        }
        q_note(cx, bq, 7, (unsigned long long) spins);
    };

    // ---- phase -1: blocks 0, 1, 2 of x are staged (kernel prologue); block 0's pre-activations, block 3's x on its way
    stage_load(3);
    project(0);
    request(1);  // (nothing real before h-block 4; keeps the loop uniform)
    q_barrier();
    for (int p = 0; p <= cx.NB; ++p) {
        q_stamp(cx, p, 0);
        // x of block p + 3: requested a phase ago, into the ring now; block p + 4 requested
        stage_write(p + 3);
        stage_load(p + 4);
        // (wave 3) unit tile 16's gate math for block p - 1, if that was this workgroup's: its recurrent accumulators were left in
        // LDS by the h waves in the phase before, nobody reads image c in this phase
        if (j == 3 && p >= 1 && ((p - 1) & 3) == c) {
            const int t = (p - 1) >> 2;
            const char *gh = cx.smem + kQOffGh16 + (t & 1) * 3 * 1024 + lane * 16;
            f32x4 acc[3];
            acc[0] = *(const f32x4 *) gh;
            acc[1] = *(const f32x4 *) (gh + 1024);
            acc[2] = *(const f32x4 *) (gh + 2048);
            h16 = q_gates(acc, gi16[0], gi16[1], gi16[2], h16);
            unsigned w0, w1;
            q_pack(h16, cx.even, w0, w1);
            q_image_write(cx, c, 16, w0, w1);
            q_publish(g, cx, t, c, 16, w0, w1);
        }
        q_stamp(cx, p, 1);
        if (p + 1 < cx.NB) project(p + 1);
        q_stamp(cx, p, 2);
        // the other workgroups' tiles of what h-block p + 1 reads (requested at the end of the phase before), then the request
        // for p + 2.  (Filing BEFORE the projection -- vector / LDS work while the partner h wave has the matrix pipe -- measured
        // slower, 405 against 354 us per layer: the requests then have had less than a phase to come back, and a global load
        // takes about 1.5 us here.)
        file(p + 1);
        q_stamp(cx, p, 3);
        asm volatile("" ::: "memory");
        request(p + 2);
        q_barrier();
    }
    if (j == 3) ((f32x4 *) g.hstate_out)[((size_t) (cx.mt0 + c) * kUnitTiles + 16) * 64 + lane] = h16;
}

// ------------------------------------------------------------------------------------------------ h waves

// 27 MFMAs of one block against the wave's register-resident W_hh tile; kW16: one gate of unit tile 16 as a fourth chain,
// its weights read from LDS through the same rolling queue
template <bool kW16>
__device__ __forceinline__ void q_h_mma(f32x4 (&acc)[3], f32x4 &a16, const bf16x8 *ha, const bf16x8 (&w)[27], const bf16x8 *w16,
                                        int lane) {
    bf16x8 qa[kQA], qw[kQA];
#pragma unroll
    for (int p = 0; p < kQA; ++p) {
        qa[p] = ha[p * 64 + lane];
        if (kW16) qw[p] = w16[p * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int blk = 0; blk < 9; ++blk) {
        const bf16x8 a = qa[blk % kQA];
#pragma unroll
        for (int gt = 0; gt < 3; ++gt)
            if (!(KQ_ABL & 16)) acc[gt] = PBF16::mma(a, w[blk * 3 + gt], acc[gt]);
        if (kW16 && !(KQ_ABL & 16)) a16 = PBF16::mma(a, qw[blk % kQA], a16);
        if (blk + kQA < 9) {
            qa[blk % kQA] = ha[(blk + kQA) * 64 + lane];
            if (kW16) qw[blk % kQA] = w16[(blk + kQA) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void q_h_wave(const GruQuadArgs &g, const QCtx &cx, const int j) {
    typedef bf16x8 frag_t;
    const int lane = cx.lane, c = cx.c, u = 4 * c + j;
    const frag_t *whh = (const frag_t *) g.whh;
    frag_t w[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) w[i] = whh[((size_t) (u * 3 + i % 3) * 9 + i / 3) * 64 + lane];
    const frag_t *w16 = (const frag_t *) (cx.smem + kQOffW16h) + (j < 3 ? j : 0) * 9 * 64;  // gate j of unit tile 16
    const float br = g.bhh[(u * 3 + 0) * 16 + cx.colq], bz = g.bhh[(u * 3 + 1) * 16 + cx.colq], bn = g.bhh[(u * 3 + 2) * 16 + cx.colq];
    const float b16 = g.bhh[(16 * 3 + (j < 3 ? j : 0)) * 16 + cx.colq];  // gate j of unit tile 16
    f32x4 hreg[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) hreg[m] = ((const f32x4 *) g.hstate_in)[((size_t) (cx.mt0 + m) * kUnitTiles + u) * 64 + lane];
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the resident weights, bias and initial state are in (see q_x_wave)
    unsigned w0p = 0, w1p = 0;  // block p - 1's tile: into its image at the start of phase p, when nobody reads that image
    q_barrier();                // (phase -1: the x waves compute block 0's pre-activations)
    for (int p = 0; p <= cx.NB; ++p) {
        q_stamp(cx, p, 0);
        if (p >= 1) q_image_write(cx, (p - 1) & 3, u, w0p, w1p);
        if (p < cx.NB) {
            const int t = p >> 2, m = p & 3;
            const frag_t *ha = (const frag_t *) (cx.smem + kQOffHs + m * kQHsBytes);  // holds h_{t-1} now
            f32x4 acc[3], a16 = f32x4{b16, b16, b16, b16};  // the recurrent chains start from b_hh
            acc[0] = f32x4{br, br, br, br};
            acc[1] = f32x4{bz, bz, bz, bz};
            acc[2] = f32x4{bn, bn, bn, bn};
            const bool with16 = (m == c) && (j < 3);
            if (with16)
                q_h_mma<true>(acc, a16, ha, w, w16, lane);
            else
                q_h_mma<false>(acc, a16, ha, w, w16, lane);
            q_stamp(cx, p, 1);
            if (with16) *(f32x4 *) (cx.smem + kQOffGh16 + ((t & 1) * 3 + j) * 1024 + lane * 16) = a16;
            const char *ring = cx.smem + kQOffGi + (((p & 1) * 4 + j) * 3) * 512 + lane * 8;
            const u32x2 pr = *(const u32x2 *) ring, pz = *(const u32x2 *) (ring + 512), pn = *(const u32x2 *) (ring + 1024);
            // (m is a runtime value: select the register, do not index the array)
            const f32x4 hprev = m == 0 ? hreg[0] : m == 1 ? hreg[1] : m == 2 ? hreg[2] : hreg[3];
            const f32x4 hnew = q_gates(acc, pr, pz, pn, hprev);
            if (m == 0) hreg[0] = hnew;
            if (m == 1) hreg[1] = hnew;
            if (m == 2) hreg[2] = hnew;
            if (m == 3) hreg[3] = hnew;
            q_stamp(cx, p, 2);
            q_pack(hnew, cx.even, w0p, w1p);
            q_publish(g, cx, t, m, u, w0p, w1p);
            q_stamp(cx, p, 3);
        }
        q_barrier();
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) ((f32x4 *) g.hstate_out)[((size_t) (cx.mt0 + m) * kUnitTiles + u) * 64 + lane] = hreg[m];
}

// ------------------------------------------------------------------------------------------------ kernel

template <int NB0>
__global__ __launch_bounds__(64 * kQWaves, 2) void gru_quad_kernel(GruQuadArgs g) {
    constexpr int NBX = 9 + NB0;
    __shared__ __attribute__((aligned(16))) char smem[kQLds];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int nquads = g.mtiles >> 2;

    QCtx cx;
    cx.smem = smem;
    cx.dbg = (g.dbg && bid == g.dbg_block) ? g.dbg + (size_t) wave * 4 * g.T * 8 : nullptr;
    cx.err = g.err;
    cx.lane = lane;
    cx.colq = lane & 15;
    cx.c = (bid >> 3) & 3;
    cx.T = g.T;
    cx.mtiles = g.mtiles;
    cx.NB = 4 * g.T;
    cx.tag_base = g.serial << 12;
    cx.even = (lane & 1) == 0;
    {
        const int row0 = (lane >> 4) * 4 + (cx.even ? 0 : 2), kk0 = cx.colq & ~1;
        cx.lane_off = (row0 + 16 * (kk0 >> 3)) * 16 + (kk0 & 7) * 2;
    }
    // one quad per four workgroups: blocks b, b + 8, b + 16, b + 24 of a group of 32 (one XCD by dispatch order).  Larger
    // batches are covered by further launches (launch_gru_quad), so every workgroup of a launch is resident at once and a
    // quad never waits for a workgroup that has not been dispatched.
    const int qq = g.quad0 + (bid >> 5) * 8 + (bid & 7);
    if (qq >= nquads) return;
    cx.mt0 = 4 * qq;

    // ---- prologue: unit tile 16's weights (LDS-resident for the whole launch), zeroed operand images (k-block 8's upper half
    // stays zero for good), initial h, x of blocks 0, 1, 2
    for (int i = wave; i < 3 * NBX; i += kQWaves)
        ((bf16x8 *) (smem + kQOffW16x))[i * 64 + lane] = ((const bf16x8 *) g.wih)[((size_t) 48 * NBX + i) * 64 + lane];
    for (int i = wave; i < 27; i += kQWaves)
        ((bf16x8 *) (smem + kQOffW16h))[i * 64 + lane] = ((const bf16x8 *) g.whh)[((size_t) 48 * 9 + i) * 64 + lane];
    for (int i = tid; i < 4 * kQHsBytes / 16; i += 64 * kQWaves) ((uint4 *) (smem + kQOffHs))[i] = uint4{0, 0, 0, 0};
    __syncthreads();
    for (int idx = wave; idx < 4 * kUnitTiles; idx += kQWaves) {
        const int m = idx / kUnitTiles, u = idx % kUnitTiles;
        const f32x4 hv = ((const f32x4 *) g.hstate_in)[((size_t) (cx.mt0 + m) * kUnitTiles + u) * 64 + lane];
        unsigned w0, w1;
        q_pack(hv, cx.even, w0, w1);
        q_image_write(cx, m, u, w0, w1);  // h_{-1}
    }
    for (int i = wave; i < 3 * NBX; i += kQWaves) {  // blocks 0, 1, 2 (= m-tiles 0, 1, 2 of step 0)
        const int blk = i / NBX, k = i % NBX;
        const bf16x8 *src = k < NB0 ? (const bf16x8 *) g.a0 + ((size_t) (cx.mt0 + blk) * NB0 + k) * 64
                                    : (const bf16x8 *) g.a1 + ((size_t) (cx.mt0 + blk) * 9 + (k - NB0)) * 64;
        ((bf16x8 *) (smem + kQOffXs))[i * 64 + lane] = src[lane];
    }
    __syncthreads();
    if (wave < 4)
        q_x_wave<NB0>(g, cx, wave);
    else
        q_h_wave(g, cx, wave - 4);
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
//              x waves: x . W_ih (waves 0..2 also one gate of unit tile 16's, m-tile c);  h waves: h . W_hh from b_hh (waves
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

template <int NB0, bool kHead>  // kHead: the y part of x is computed here (the previous stage's narrow head), not read
__global__ __launch_bounds__(64 * kQWaves, 2) void gru_quad1_kernel(GruQuadArgs g) {
    typedef bf16x8 frag_t;
    constexpr int NBX = 9 + NB0;
    static_assert(!kHead || NB0 > 0, "a head feeds a y part");
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
    constexpr int kCh = kHead ? NB0 : 1;
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
        if (i < 3 * NBX)
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
            if (t == kUnitTiles) w0 = w1 = 0u;
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
            uint16_t *sc = (uint16_t *) (smem + kQ1OffXs + (m * 11 + (nt >> 1)) * 1024);
            const bool pad = nt * 16 + colq >= g.yvalid;
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
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) w[gt][blk] = wih[(gt * NBX + blk) * 64];
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
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) w[blk][gt] = whh[(gt * 9 + blk) * 64];
        w16[blk] = whh16[blk * 64];  // (used by waves 0..2; requested by all four: no conditional load in the stream)
    };
#pragma unroll
    for (int blk = 0; blk < kQ1Ahead && blk < 9; ++blk) request(blk);
    const float b0 = g.bhh[(u * 3 + 0) * 16 + colq], b1 = g.bhh[(u * 3 + 1) * 16 + colq], b2 = g.bhh[(u * 3 + 2) * 16 + colq];
    const float b16 = g.bhh[(16 * 3 + (j < 3 ? j : 0)) * 16 + colq];
    stamp(1);
    images();
    head();
    stamp(2);
    __syncthreads();  // barrier A
    stamp(3);
    f32x4 acc[4][3], a16 = f32x4{b16, b16, b16, b16};
#pragma unroll
    for (int m = 0; m < 4; ++m) {  // the recurrent chains start from b_hh
        acc[m][0] = f32x4{b0, b0, b0, b0};
        acc[m][1] = f32x4{b1, b1, b1, b1};
        acc[m][2] = f32x4{b2, b2, b2, b2};
    }
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
        if (a.T == 1) {  // one step: no exchange, no phases (gru_quad1_kernel)
            const bool head = a.yw != nullptr && a.nb0 > 0;  // the previous stage's narrow head rides along
            if (a.nb0 == 0)
                hipLaunchKernelGGL((gru_quad1_kernel<0, false>), grid, block, 0, s, g);
            else if (a.nb0 == 1 && head)
                hipLaunchKernelGGL((gru_quad1_kernel<1, true>), grid, block, 0, s, g);
            else if (a.nb0 == 1)
                hipLaunchKernelGGL((gru_quad1_kernel<1, false>), grid, block, 0, s, g);
            else if (head)
                hipLaunchKernelGGL((gru_quad1_kernel<2, true>), grid, block, 0, s, g);
            else
                hipLaunchKernelGGL((gru_quad1_kernel<2, false>), grid, block, 0, s, g);
            continue;
        }
        if (a.nb0 == 0)
            hipLaunchKernelGGL(gru_quad_kernel<0>, grid, block, 0, s, g);
        else if (a.nb0 == 1)
            hipLaunchKernelGGL(gru_quad_kernel<1>, grid, block, 0, s, g);
        else
            hipLaunchKernelGGL(gru_quad_kernel<2>, grid, block, 0, s, g);
    }
}

}  // namespace kns
