// kns_gru.hip -- the GRU layers: recurrent halves streaming and resident (8 waves), whole layers of one frame
// (low-latency kernel) and of several frames as a wavefront over (layer, frame).
#include "kns_device.hpp"

namespace kns {

// ------------------------------------------------------------------------------------------------ recurrent GRU

// Weights streamed from L2 every step (fp32 parity path; bf16 only as an A/B switch).  W waves per workgroup share the
// 17 unit tiles round-robin; the B fragments of a tile are fetched four k-blocks (twelve fragments) ahead of their MFMAs, and with W = 8 two waves per SIMD cover each other's latencies.
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
    // bf16 configuration: the operand's k = 271, 272 are the constant 1 against the two bias rows of the packed W_hh (kns_layout.h,
    // kBiasK0); k = 271 is rewritten with every image of tile 16 (operand_of below), k = 272 stays as set here, in both buffers
    if (P::kPrec == kBf16 && tid < 64) {
        const int k = kBiasK0 + ((tid >> 4) & 1);
        ((elem_t *) hbuf[tid >> 5])[(k / P::KB) * 64 * P::EPL + P::off(tid & 15, k % P::KB)] = (elem_t) kBf16One;
    }
    auto operand_of = [&](float h, int u) -> elem_t {
        elem_t e = P::cvt(h);
        if constexpr (P::kPrec == kBf16) {  // (a mask, not a condition: no branch around the stores)
            const uint32_t one_here = 0u - (uint32_t) ((u == kUnitTiles - 1) & (colq == 15));
            e = (elem_t) (((uint32_t) e & ~one_here) | (kBf16One & one_here));
        }
        return e;
    };
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int u = wave + W * q;
        if (u < kUnitTiles) {
            const int k = u * 16 + colq;
            elem_t *dst = (elem_t *) hbuf[0] + (k / P::KB) * 64 * P::EPL;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[P::off(rowq + i, k % P::KB)] = operand_of(hreg[q][i], u);
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
                const float br = g.bhh[(u * 3 + 0) * 16 + colq];
                const float bz = g.bhh[(u * 3 + 1) * 16 + colq];
                const float bn = g.bhh[(u * 3 + 2) * 16 + colq];
                f32x4 acc[3];
#pragma unroll
                for (int gt = 0; gt < 3; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};  // (bf16: b_hh rides in the operands, kns_layout.h)
                // B fragments two k-blocks ahead of their MFMAs, rotated through registers (a rolled loop: unrolling all 51
                // fp32 k-block/gate pairs makes hipcc materialise an address pair per load and spill)
                const frag_t *wu = whh + (size_t) u * 3 * NBH * 64 + lane;
                constexpr int kAhead = 4;  // k-blocks in flight per wave: 12 fragments = 12 KiB per wave, 96 KiB per CU
                frag_t bq[kAhead][3];
#pragma unroll
                for (int p = 0; p < kAhead; ++p)
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) bq[p][gt] = wu[(gt * NBH + (p < NBH ? p : NBH - 1)) * 64];
#pragma nounroll
                for (int blk = 0; blk < NBH; ++blk) {
                    const frag_t ab = ha[blk * 64 + lane];
                    frag_t bc[3];
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) {
                        bc[gt] = bq[0][gt];
#pragma unroll
                        for (int p = 0; p + 1 < kAhead; ++p) bq[p][gt] = bq[p + 1][gt];
                    }
                    const int nb = blk + kAhead < NBH ? blk + kAhead : NBH - 1;
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) bq[kAhead - 1][gt] = wu[(gt * NBH + nb) * 64];
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) acc[gt] = P::mma(ab, bc[gt], acc[gt]);
                }
                f32x4 ir = P::from_gi(gir), iz = P::from_gi(giz), in = P::from_gi(gin);
                const int k = u * 16 + colq;
                elem_t *dst = (elem_t *) hbuf[cur ^ 1] + (k / P::KB) * 64 * P::EPL;
                if (P::kPrec == kBf16) {  // the bf16 configuration's gate arithmetic, identical in every bf16 kernel
#pragma unroll
                    for (int i = 0; i < 4; ++i) hreg[q][i] = gate_elem_bf16(ir[i], iz[i], in[i], acc[0][i], acc[1][i], acc[2][i], hreg[q][i]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) dst[P::off(rowq + i, k % P::KB)] = operand_of(hreg[q][i], u);
                } else {
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

// ---- low-latency GRU layer: input GEMM + recurrent GEMM + gates of one frame, one wavefront per (unit tile, m-tile).
// Arithmetic is, operation for operation, what the chunked path does (same MFMA, same k order, gi rounded to its storage
// type before the gates, same gate formulas per precision), so a stream's samples do not depend on which path ran.
// kHead: the y part is the previous stage's narrow head, computed here (every workgroup for its m-tile) instead of read
// kPad (with kHead, NB0 == 0): the head's at most kYPadMax values go INTO the features' last k-block, columns 1 ... yvalid of block
// NBH - 1 ([features ; y_prev] sharing a k-block, kns_layout.h) instead of being a y part in front of x
// (the body is shared by the one-layer launch below and by gru_wave_kernel, which runs several layers' workgroups in one launch;
// u, mt: the workgroup's unit tile and m-tile; the four LDS areas are the caller's)
template <class P, int NB0, bool kHead, bool kPad = false>  // NB0: k-blocks of the y part of the layer input (everything static: no branch around a load or an MFMA)
__device__ __forceinline__ void gru_small_body(const GruSmallArgs &g, const int u, const int mt, char *hbuf, char *hspare, char *ybuf,
                                               f32x4 (*xch)[3][64]) {
    static_assert(!kPad || (kHead && NB0 == 0 && P::kPrec == kBf16), "head into the padding: a fused head, no y part, bf16 layout");
    // One workgroup per (unit tile, m-tile), one wave per gate: each wave streams only its gate's weights (a third of the
    // tile's), eight k-blocks of operands requested before the MFMAs that use them; the three accumulator pairs meet
    // in LDS and wave 0 does the gate math.  Every accumulator still sums its k-blocks in ascending order, so the result is
    // bit-identical to the chunked kernels.
    typedef typename P::frag_t frag_t;
    typedef typename P::elem_t elem_t;
    constexpr int NBH = P::NBH;
    const int lane = threadIdx.x & 63;
    const int gt = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // this wave's gate: r, z, n
    const int colq = lane & 15, rowq = (lane >> 4) * 4;
    constexpr int nb = NB0 + NBH;

    // Everything the wave needs from memory is requested before anything is waited for -- h_{t-1}, the operand blocks, this
    // gate's W_ih and W_hh fragments, the biases: ONE memory round trip per launch (with the staging of h, the two weight queues
    // and the biases one after the other it was five, and a one-frame call is fifteen such launches).
    constexpr int kHT = (kUnitTiles + 2) / 3;  // tiles of h_{t-1} this wave converts
    const f32x4 hown = ((const f32x4 *) g.hstate_in)[((size_t) mt * kUnitTiles + u) * 64 + lane];
    f32x4 hv[kHT];
#pragma unroll
    for (int q = 0; q < kHT; ++q) {
        const int v = gt + 3 * q;
        hv[q] = ((const f32x4 *) g.hstate_in)[((size_t) mt * kUnitTiles + (v < kUnitTiles ? v : kUnitTiles - 1)) * 64 + lane];
    }
    const frag_t *wih = (const frag_t *) g.wih + (size_t) (u * 3 + gt) * nb * 64 + lane;
    const frag_t *whh = (const frag_t *) g.whh + (size_t) (u * 3 + gt) * NBH * 64 + lane;
    const frag_t *a0 = (const frag_t *) g.a0 + (size_t) mt * NB0 * 64 + lane;
    const frag_t *a1 = (const frag_t *) g.a1 + (size_t) mt * NBH * 64 + lane;
    // (kHead) NB0 k-blocks of NPB n-tiles = 1 .. 4 chains of NBH MFMAs, chain c on wave c mod 3; a wave whose slot is past the
    // last chain repeats it (same values to the same words: no branch around the loads or the MFMAs)
    constexpr int kChains = kHead ? (kPad ? P::NPB : NB0 * P::NPB) : 1, kCw = (kChains + 2) / 3;
    frag_t ya[kHead ? NBH : 1], yw[kCw][kHead ? NBH : 1];
    float ybias[kCw];
    if (kHead) {
#pragma unroll
        for (int p = 0; p < NBH; ++p) ya[p] = ((const frag_t *) g.yh)[((size_t) mt * NBH + p) * 64 + lane];
#pragma unroll
        for (int q = 0; q < kCw; ++q) {
            const int c = gt + 3 * q < kChains ? gt + 3 * q : kChains - 1;
#pragma unroll
            for (int p = 0; p < NBH; ++p) yw[q][p] = ((const frag_t *) g.yw)[((size_t) c * NBH + p) * 64 + lane];
            ybias[q] = g.yb[c * 16 + colq];
        }
    }
    frag_t xa[nb], wi[nb], wh[NBH];
#pragma unroll
    for (int p = 0; p < nb; ++p) {
        if (!(kHead && p < NB0)) xa[p] = p < NB0 ? a0[(size_t) p * 64] : a1[(size_t) (p - NB0) * 64];
        wi[p] = wih[(size_t) p * 64];
    }
#pragma unroll
    for (int p = 0; p < NBH; ++p) wh[p] = whh[(size_t) p * 64];
    const float bi = g.bih[(u * 3 + gt) * 16 + colq];
    const float br = g.bhh[(u * 3 + 0) * 16 + colq], bz = g.bhh[(u * 3 + 1) * 16 + colq], bn = g.bhh[(u * 3 + 2) * 16 + colq];
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the loads to their uses to save registers: five round trips again)

    // h_{t-1}: fp32 C-fragments -> operand-typed A-fragments in LDS (the 17 tiles shared out over the three waves; the columns
    // past tile 16 -- the upper half of the bf16 layout's last k-block -- are written as zeros by the wave that has a slot free)
#pragma unroll
    for (int q = 0; q < kHT; ++q) {  // (no branch: a slot past the image -- fp32 layout, tile 17 -- writes into a spare block instead)
        const int v = gt + 3 * q;
        const int k = v * 16 + colq;
        elem_t *dst = v * 16 < NBH * P::KB ? (elem_t *) hbuf + (k / P::KB) * 64 * P::EPL : (elem_t *) hspare;
        const uint32_t keep = v < kUnitTiles ? 0xffffffffu : 0u;  // (a mask, not a branch: slot 17 is the zero tile)
        // bf16 configuration: k = 271 (column 15 of tile 16) and k = 272 (column 0 of the empty slot 17) are the constant 1 against
        // the two bias rows of the packed W_hh (kns_layout.h, kBiasK0).  As a lane MASK, not a condition: with `&&` / `||` hipcc
        // turned the wave-uniform part into branches around every store, and their merged waits cost the launch a round trip
        // (4.1 -> 5.4 us per layer at 64 streams)
        const uint32_t one_here = 0u - (uint32_t) (((v == kUnitTiles - 1) & (colq == 15)) | ((v == kUnitTiles) & (colq == 0)));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            elem_t e = P::cvt(u2f(f2u(hv[q][i]) & keep));
            if constexpr (P::kPrec == kBf16) e = (elem_t) (((uint32_t) e & ~one_here) | (kBf16One & one_here));
            dst[P::off(rowq + i, k % P::KB)] = e;
        }
    }
    static_assert(3 * kHT * 16 >= NBH * P::KB, "the three waves' tile slots cover the operand image");

    if (kHead) {  // y_prev = sigmoid(h_B . W_head + b_head), columns >= yvalid zero, rounded to the operand type (what the head GEMM stores)
#pragma unroll
        for (int q = 0; q < kCw; ++q) {
            const int c = gt + 3 * q < kChains ? gt + 3 * q : kChains - 1;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < NBH; ++p) acc = P::mma(ya[p], yw[q][p], acc);
            elem_t *sc = (elem_t *) ybuf + (c / P::NPB) * 64 * P::EPL;
            const bool pad = c * 16 + colq >= g.yvalid;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = pad ? 0.0f : head_sigmoid<P>(acc[i] + ybias[q]);
                sc[P::off(rowq + i, (c % P::NPB) * 16 + colq)] = P::cvt(x);
            }
        }
    }
    f32x4 acci = f32x4{0.f, 0.f, 0.f, 0.f}, acch = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!kHead) {
#pragma unroll
        for (int p = 0; p < nb; ++p) acci = P::mma(xa[p], wi[p], acci);
    }
    __syncthreads();  // hbuf (and ybuf) complete
#pragma unroll
    for (int p = 0; p < NBH; ++p) acch = P::mma(((const frag_t *) hbuf)[p * 64 + lane], wh[p], acch);
    if (kHead && kPad) {
        // the head's values sit in ybuf as an A block of their own (k = column); they belong at k = 1 + column of the features'
        // last block: both are the first lane group's 16 bytes (k 0..7 of row lane), so lanes 0..15 shift them up by one element
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 yf = __builtin_bit_cast(u32x4, ((const frag_t *) ybuf)[lane]);
        u32x4 xf = __builtin_bit_cast(u32x4, xa[nb - 1]);
        // 16-bit elements e0..e7 in four dwords; merged element i (1 <= i <= yvalid) = y element i - 1
        u32x4 sh;  // y shifted up by one element
        sh[0] = yf[0] << 16;
        sh[1] = (yf[0] >> 16) | (yf[1] << 16);
        sh[2] = (yf[1] >> 16) | (yf[2] << 16);
        sh[3] = (yf[2] >> 16) | (yf[3] << 16);
        if (lane < 16) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                // mask of the elements of dword d that take the head's value: elements 2 d, 2 d + 1 with 1 <= element <= yvalid
                const unsigned lo = (2 * d >= 1 && 2 * d <= g.yvalid) ? 0x0000ffffu : 0u, hi = (2 * d + 1 <= g.yvalid) ? 0xffff0000u : 0u;
                xf[d] = (xf[d] & ~(lo | hi)) | (sh[d] & (lo | hi));
            }
        }
        const frag_t xlast = __builtin_bit_cast(frag_t, xf);
#pragma unroll
        for (int p = 0; p < nb; ++p) acci = P::mma(p == nb - 1 ? xlast : xa[p], wi[p], acci);
    } else if (kHead) {  // the input-side chain after the barrier: its first NB0 blocks are the head's output
#pragma unroll
        for (int p = 0; p < nb; ++p) acci = P::mma(p < NB0 ? ((const frag_t *) ybuf)[p * 64 + lane] : xa[p], wi[p], acci);
    }
    {
        f32x4 v = acci;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] + bi;
        xch[0][gt][lane] = P::from_gi(P::to_gi(v));
        xch[1][gt][lane] = acch;
    }
    __syncthreads();
    if (gt != 0) return;
    f32x4 gin[3], gh[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        gin[q] = xch[0][q][lane];
        gh[q] = xch[1][q][lane];
    }
    f32x4 hnew;
    if (P::kPrec == kBf16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) hnew[i] = gate_elem_bf16(gin[0][i], gin[1][i], gin[2][i], gh[0][i], gh[1][i], gh[2][i], hown[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float r = kns_sigmoid(gin[0][i] + (gh[0][i] + br));
            float z = kns_sigmoid(gin[1][i] + (gh[1][i] + bz));
            float n = kns_tanh(__builtin_fmaf(r, gh[2][i] + bn, gin[2][i]));
            hnew[i] = __builtin_fmaf(z, hown[i] - n, n);
        }
    }
    ((f32x4 *) g.hstate_out)[((size_t) mt * kUnitTiles + u) * 64 + lane] = hnew;
    const int k = u * 16 + colq;
    elem_t *hs = (elem_t *) g.hseq + ((size_t) mt * NBH + k / P::KB) * 64 * P::EPL;
#pragma unroll
    for (int i = 0; i < 4; ++i) hs[P::off(rowq + i, k % P::KB)] = P::cvt(hnew[i]);
}

template <class P, int NB0, bool kHead, bool kPad = false>
__global__ __launch_bounds__(192) void gru_small_kernel(GruSmallArgs g) {
    __shared__ __attribute__((aligned(16))) char hbuf[P::NBH * 1024];
    __shared__ __attribute__((aligned(16))) char hspare[1024];
    __shared__ __attribute__((aligned(16))) char ybuf[(NB0 > 0 ? NB0 : 1) * 1024];  // (kHead) the y part as A fragments
    __shared__ f32x4 xch[2][3][64];  // [input | recurrent][gate][lane]
    gru_small_body<P, NB0, kHead, kPad>(g, blockIdx.x, blockIdx.y, hbuf, hspare, ybuf, xch);
}

// ---- calls of several frames as a WAVEFRONT over (layer, frame).  Layer l of frame t needs layer l - 1 of frame t and layer l of
// frame t - 1, so the items of one anti-diagonal -- one per pipeline stage: the eight GRU layers and the three narrow heads between
// the stages -- are independent and run side by side in ONE launch: T + 10 launches per call instead of 8 T, each with enough
// workgroups for the chip.  One layer per XCD (gru_wave_kernel below).  A layer workgroup owns (unit tile, group of m-tiles) and has
// four waves: three MFMA waves (gates r, z, n) and one that requests operands and does the gate arithmetic (details in the body);
// the operands of an m-tile -- y, x and h_{t-1}, all in operand form -- go from memory straight into LDS.  The arithmetic is
// gru_small_body's, operation for operation (same chains, same k order, gi rounded to its storage type, same gate formulas), so
// the PCM does not depend on the route.  Measured and not adopted: DESIGN.md section 6 (one barrier per m-tile with an LDS flag:
// races; a five-wave bf16 form: fewer workgroups per CU, slower).
template <class P, int NB0>
__device__ __forceinline__ void gru_wave_layer(const GruWaveItem &it, const int u, const int m0, const int m1, const int role, char *obuf,
                                               f32x4 *xch, const bool dbg_stamp) {
    typedef typename P::frag_t frag_t;
    typedef typename P::elem_t elem_t;
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    constexpr int NBH = P::NBH, nb = NB0 + NBH, nop = nb + NBH;  // operand blocks of an m-tile: y, x, h_{t-1}
    constexpr int kBufBytes = (2 * P::NBH + 3) * 1024;
    const GruSmallArgs &g = it.g;
    const int lane = threadIdx.x & 63;
    const int colq = lane & 15, rowq = (lane >> 4) * 4;
    // Four waves: roles 0..2 are the gates r, z, n -- each keeps its gate's W_ih and W_hh fragments in registers for the whole group
    // of m-tiles and runs the two MFMA chains of an m-tile, nothing else; role 3 requests the NEXT m-tile's operand blocks and does
    // the gate arithmetic of the m-tile BEFORE, both under those MFMAs.  Two barriers per m-tile: B1 (the operand blocks are in LDS,
    // the exchange area is free), B2 (the pre-activations are in the exchange area; the other operand buffer is free); between B2
    // and the next B1 role 3 only takes the pre-activations into registers and waits for the blocks it requested.
    if (role == 3) {
        // An m-tile's operand blocks go from memory straight into LDS (no registers, no conversion: h_{t-1} is read in operand form --
        // the hidden sequence's slot of frame t - 1, or the call's converted state for its first frame), into the buffer the MFMAs
        // in flight do not read.
        auto request = [&](int mt, char *buf) {
#pragma unroll
            for (int p = 0; p < nop; ++p) {
                const frag_t *src = p < NB0 ? (const frag_t *) g.a0 + ((size_t) mt * NB0 + p) * 64
                                  : p < nb  ? (const frag_t *) g.a1 + ((size_t) mt * NBH + (p - NB0)) * 64
                                            : (const frag_t *) it.hprev + ((size_t) mt * NBH + (p - nb)) * 64;
                __builtin_amdgcn_global_load_lds((gptr_t) (src + lane), (lptr_t) (buf + p * 1024), 16, 0, 0);
            }
        };
        auto landed = [&](char *buf) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (P::kPrec == kBf16) {
                // k = 271 and k = 272 of the h operand are the constant 1 against the two bias rows of the packed W_hh (kns_layout.h,
                // kBiasK0): elements 15 and 16 of rows 0 .. 15 of h's last block, written once it has landed
                if (lane < 32) ((elem_t *) (buf + (nop - 1) * 1024))[P::off(lane & 15, 15 + (lane >> 4))] = (elem_t) kBf16One;
            }
        };
        // the exchange area, bf16: [input | recurrent][gate] accumulators; fp32 (LDS is the resource there: two workgroups per CU
        // leave 4 KiB): the sums the gate formulas start with, made by the MFMA waves -- r: gi + (gh + b_r), z: gi + (gh + b_z),
        // n: gi and gh + b_n -- same operations in the same order as gru_small_body's
        constexpr int kXW = P::kPrec == kBf16 ? 6 : 4;
        f32x4 pre[kXW], hown = ((const f32x4 *) g.hstate_in)[((size_t) m0 * kUnitTiles + u) * 64 + lane];
        auto gates = [&](int mt) {
            f32x4 hnew;
            if constexpr (P::kPrec == kBf16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) hnew[i] = gate_elem_bf16(pre[0][i], pre[1][i], pre[2][i], pre[3][i], pre[4][i], pre[5][i], hown[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float r = kns_sigmoid(pre[0][i]);
                    float z = kns_sigmoid(pre[1][i]);
                    float n = kns_tanh(__builtin_fmaf(r, pre[3][i], pre[2][i]));
                    hnew[i] = __builtin_fmaf(z, hown[i] - n, n);
                }
            }
            ((f32x4 *) g.hstate_out)[((size_t) mt * kUnitTiles + u) * 64 + lane] = hnew;
            const int k = u * 16 + colq;
            elem_t *hs = (elem_t *) g.hseq + ((size_t) mt * NBH + k / P::KB) * 64 * P::EPL;
#pragma unroll
            for (int i = 0; i < 4; ++i) hs[P::off(rowq + i, k % P::KB)] = P::cvt(hnew[i]);
        };
        auto take = [&]() {  // the pre-activations of the m-tile the MFMA waves have just finished; then the exchange area is theirs again
#pragma unroll
            for (int q = 0; q < kXW; ++q) pre[q] = xch[q * 64 + lane];
        };
        request(m0, obuf);
        landed(obuf);
        int cur = 0;
        for (int mt = m0; mt < m1; ++mt) {
            KNS_WSTAMP(0);
            __syncthreads();  // B1
            KNS_WSTAMP(1);
            cur ^= 1;
            if (mt + 1 < m1) request(mt + 1, obuf + cur * kBufBytes);
            if (mt > m0) {
                gates(mt - 1);
                hown = ((const f32x4 *) g.hstate_in)[((size_t) mt * kUnitTiles + u) * 64 + lane];
            }
            KNS_WSTAMP(2);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B2 (not __syncthreads: the requested blocks stay in flight)
            KNS_WSTAMP(3);
            take();
            landed(obuf + cur * kBufBytes);
            KNS_WSTAMP(4);
        }
        __syncthreads();  // (the MFMA waves' last barrier)
        gates(m1 - 1);
        return;
    }
    const int gt = role;
    frag_t wi[nb], wh[NBH];
    {
        const frag_t *wih = (const frag_t *) g.wih + (size_t) (u * 3 + gt) * nb * 64 + lane;
        const frag_t *whh = (const frag_t *) g.whh + (size_t) (u * 3 + gt) * NBH * 64 + lane;
#pragma unroll
        for (int p = 0; p < nb; ++p) wi[p] = wih[(size_t) p * 64];
#pragma unroll
        for (int p = 0; p < NBH; ++p) wh[p] = whh[(size_t) p * 64];
    }
    const float bi = g.bih[(u * 3 + gt) * 16 + colq], bh = g.bhh[(u * 3 + gt) * 16 + colq];
    int cur = 0;
    for (int mt = m0; mt < m1; ++mt) {
        char *buf = obuf + cur * kBufBytes;
        KNS_WSTAMP(0);
        __syncthreads();  // B1
        KNS_WSTAMP(1);
        cur ^= 1;
        // the two chains side by side (neither waits for its own previous MFMA), the operand fragments of block p + 1 read from LDS
        // under the MFMAs of block p; each chain still sums its k-blocks in ascending order
        f32x4 acci = f32x4{0.f, 0.f, 0.f, 0.f}, acch = f32x4{0.f, 0.f, 0.f, 0.f};
        const frag_t *bl = (const frag_t *) buf + lane;
#pragma unroll
        for (int p = 0; p < NB0; ++p) acci = P::mma(bl[p * 64], wi[p], acci);
        frag_t xn = bl[NB0 * 64], hn = bl[nb * 64];
#pragma unroll
        for (int p = 0; p < NBH; ++p) {
            const frag_t xc = xn, hc = hn;
            if (p + 1 < NBH) {
                xn = bl[(NB0 + p + 1) * 64];
                hn = bl[(nb + p + 1) * 64];
            }
            P::mma2(xc, wi[NB0 + p], acci, hc, wh[p], acch);
        }
        {
            f32x4 v = acci;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = v[i] + bi;
            KNS_WSTAMP(2);
            if constexpr (P::kPrec == kBf16) {
                xch[gt * 64 + lane] = P::from_gi(P::to_gi(v));
                xch[(3 + gt) * 64 + lane] = acch;
            } else {
                f32x4 hb;
#pragma unroll
                for (int i = 0; i < 4; ++i) hb[i] = acch[i] + bh;
                if (gt == 2) {
                    xch[2 * 64 + lane] = v;
                    xch[3 * 64 + lane] = hb;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) hb[i] = v[i] + hb[i];
                    xch[gt * 64 + lane] = hb;
                }
            }
        }
        __syncthreads();  // B2
        KNS_WSTAMP(3);
    }
    __syncthreads();  // role 3's last B1
}

// the state of a call's first frame in operand form: fp32 tiles [layer][m-tile][17][64][4] -> A-packed [layer][m-tile][NBH] blocks
// (the padding columns stay as allocated: zero)
template <class P>
__global__ __launch_bounds__(64) void gru_wave_prev_kernel(const float *hstate, void *hprev) {
    typedef typename P::elem_t elem_t;
    const int lane = threadIdx.x, v = blockIdx.x;
    const size_t mtl = blockIdx.y;  // layer * mtiles + m-tile
    const int colq = lane & 15, rowq = (lane >> 4) * 4;
    const f32x4 h = ((const f32x4 *) hstate)[(mtl * kUnitTiles + v) * 64 + lane];
    const int k = v * 16 + colq;
    elem_t *dst = (elem_t *) hprev + (mtl * P::NBH + k / P::KB) * 64 * P::EPL;
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[P::off(rowq + i, k % P::KB)] = P::cvt(h[i]);
}

void launch_gru_wave_prev(const float *hstate, void *hprev, int layers_x_mtiles, int precision, hipStream_t s) {
    dim3 grid(kUnitTiles, layers_x_mtiles);
    if (precision == kBf16)
        hipLaunchKernelGGL(gru_wave_prev_kernel<PBF16>, grid, dim3(64), 0, s, hstate, hprev);
    else
        hipLaunchKernelGGL(gru_wave_prev_kernel<PF32>, grid, dim3(64), 0, s, hstate, hprev);
}

// a narrow head as an item of the wavefront: y = sigmoid(h_B . W_head + b_head) of one m-tile, rounded to the operand type, columns
// >= yvalid zero -- what gemm_head_kernel stores; chain c (an n-tile of 16 columns, NBH k-blocks) on wave c mod 4.  y_nb > 0: into
// the y operand [m-tiles][y_nb] blocks; pad: into columns y_kk0 ... of block y_blk of the features (bf16, at most kYPadMax values)
template <class P>
__device__ __forceinline__ void gru_wave_head(const GruWaveItem &it, const int mt) {
    typedef typename P::frag_t frag_t;
    typedef typename P::elem_t elem_t;
    constexpr int NBH = P::NBH;
    const GruSmallArgs &g = it.g;
    const int lane = threadIdx.x & 63;
    const int gt = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int colq = lane & 15, rowq = (lane >> 4) * 4;
    for (int c = gt; c < it.chains; c += 4) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        frag_t ya[NBH], yw[NBH];
#pragma unroll
        for (int p = 0; p < NBH; ++p) ya[p] = ((const frag_t *) g.yh)[((size_t) mt * NBH + p) * 64 + lane];
#pragma unroll
        for (int p = 0; p < NBH; ++p) yw[p] = ((const frag_t *) g.yw)[((size_t) c * NBH + p) * 64 + lane];
        const float ybias = g.yb[c * 16 + colq];
#pragma unroll
        for (int p = 0; p < NBH; ++p) acc = P::mma(ya[p], yw[p], acc);
        const int col = c * 16 + colq;
        const bool past = col >= g.yvalid;
        if (it.pad) {
            elem_t *dst = (elem_t *) it.yout + ((size_t) mt * it.y_nb + it.y_blk) * 64 * P::EPL;
            if (!past) {
#pragma unroll
                for (int i = 0; i < 4; ++i) dst[P::off(rowq + i, it.y_kk0 + col)] = P::cvt(head_sigmoid<P>(acc[i] + ybias));
            }
        } else {
            elem_t *dst = (elem_t *) it.yout + ((size_t) mt * it.y_nb + c / P::NPB) * 64 * P::EPL;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                dst[P::off(rowq + i, (c % P::NPB) * 16 + colq)] = P::cvt(past ? 0.0f : head_sigmoid<P>(acc[i] + ybias));
        }
    }
}

template <class P>
__global__ __launch_bounds__(256, 2) void gru_wave_kernel(GruWaveArgs w) {
    __shared__ __attribute__((aligned(16))) char obuf[2 * (2 * P::NBH + 3) * 1024];  // two operand buffers of [y | x | h] blocks
    __shared__ f32x4 xch[(P::kPrec == kBf16 ? 6 : 4) * 64];
    // Workgroups are dealt to the eight XCDs round-robin by their linear index, and each XCD has its own L2: every workgroup of a
    // layer runs on ONE XCD (layer l on XCD l), so an XCD pulls one layer's weights per launch (1.8 MB in fp32) and serves its
    // workgroups from L2 -- dealt across the chip, every XCD's 4 MB L2 would see all eight layers' 14 MB and keep none.
    // (while the pipeline fills and drains fewer than eight layers are in a launch: four or fewer share the XCDs 2, 4 or 8 to a layer,
    // XCD x taking every parts-th workgroup of its layer)
    const int xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
    const int li = w.layer_item[xcd], hi = w.head_item[xcd];
    if (xslot >= w.xcd_wgs) {  // a head: one workgroup per m-tile
        if (hi >= 0 && xslot - w.xcd_wgs < w.item[hi].g.mtiles) gru_wave_head<P>(w.item[hi], xslot - w.xcd_wgs);
        return;
    }
    const int slot = xslot * w.parts + w.layer_part[xcd];
    if (li < 0 || slot >= w.layer_wgs) return;
    const GruWaveItem &it = w.item[li];
    const int mtiles = it.g.mtiles;
    const int u = slot % kUnitTiles, grp = slot / kUnitTiles;
    const int m0 = grp * w.mgroup, m1 = m0 + w.mgroup < mtiles ? m0 + w.mgroup : mtiles;
    // (the gate wave sits on a different SIMD from workgroup to workgroup, so that the CU's four matrix pipes share the MFMA waves)
    const int role = __builtin_amdgcn_readfirstlane(((threadIdx.x >> 6) + slot) & 3);
    const bool dbg = w.stamp && xcd == 3 && slot == 1;  // (KNS_TIMING builds)
    switch (__builtin_amdgcn_readfirstlane(it.g.nb0)) {
        case 0: gru_wave_layer<P, 0>(it, u, m0, m1, role, obuf, xch, dbg); break;
        case 1: gru_wave_layer<P, 1>(it, u, m0, m1, role, obuf, xch, dbg); break;
        case 2: gru_wave_layer<P, 2>(it, u, m0, m1, role, obuf, xch, dbg); break;
        default: gru_wave_layer<P, 3>(it, u, m0, m1, role, obuf, xch, dbg); break;
    }
}

void launch_gru_wave(const GruWaveArgs &w, int precision, int mtiles, hipStream_t s) {
    bool heads = false;
    for (int x = 0; x < 8; ++x) heads = heads || w.head_item[x] >= 0;
    dim3 grid(8 * (w.xcd_wgs + (heads ? mtiles : 0)));
    if (precision == kBf16)
        hipLaunchKernelGGL(gru_wave_kernel<PBF16>, grid, dim3(256), 0, s, w);
    else
        hipLaunchKernelGGL(gru_wave_kernel<PF32>, grid, dim3(256), 0, s, w);
}

void launch_gru_small(const GruSmallArgs &a, hipStream_t s) {
    dim3 grid(kUnitTiles, a.mtiles);
    auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(192), 0, s, a); };
    const bool head = a.yw != nullptr;  // the previous stage's narrow head rides along
    if (a.precision == kBf16) {
        switch (a.nb0) {
            case 0: head ? go(gru_small_kernel<PBF16, 0, true, true>) : go(gru_small_kernel<PBF16, 0, false>); break;
            case 1: head ? go(gru_small_kernel<PBF16, 1, true>) : go(gru_small_kernel<PBF16, 1, false>); break;
            default: head ? go(gru_small_kernel<PBF16, 2, true>) : go(gru_small_kernel<PBF16, 2, false>); break;
        }
    } else {
        switch (a.nb0) {
            case 0: go(gru_small_kernel<PF32, 0, false>); break;  // (fp32 keeps its front-end and every y part in front of x)
            case 1: head ? go(gru_small_kernel<PF32, 1, true>) : go(gru_small_kernel<PF32, 1, false>); break;
            case 2: head ? go(gru_small_kernel<PF32, 2, true>) : go(gru_small_kernel<PF32, 2, false>); break;
            default: head ? go(gru_small_kernel<PF32, 3, true>) : go(gru_small_kernel<PF32, 3, false>); break;
        }
    }
}

// ---- 8-wave form of the resident recurrent kernel.  Measured on MI355X: a single wave per SIMD executes its MFMAs and
// its gate VALU one after the other (a unit tile costs 432 + ~650 cycles), while two waves on one SIMD overlap them.
// Eight waves of <= 256 registers hold the 459 KiB of W_hh as: wave w owns unit tiles w and w + 8; 41 of its 54 fragments in VGPRs
// (round 4: the first 14 of tile w and all 27 of tile w + 8), 13 in LDS, tile 16 in LDS: 328 KiB of registers + 131 KiB of LDS.
// A fragments are re-read from LDS per k-block.
constexpr int kR8Waves = 8;
#ifndef R8C_Q
#define R8C_Q 3
#endif
// 41 of a wave's 54 fragments fit its registers; 13 live in LDS and go through a 3-deep register queue in their tile's MFMA loop.
// Which tile carries them (R8_REG0 = register-resident fragments of the FIRST tile), A/B through bench.py on one box, us per launch:
// 27 (rounds 1-3: the second tile carries all 13) 155.3 | 23: 160.2 | 20 (even split) 157.4 | 17: 155.8 | 14 (the first tile carries
// all 13) 154.0.  The stamps' reading that the second MFMA window is bound by the LDS pipe (8 waves x 31 reads) did not hold: an
// even split is slower, not faster; the first window, where all eight waves start together behind the barrier, absorbs the queue best.
#ifndef R8_REG0
#define R8_REG0 15
#endif
#ifndef R8_REGS
#define R8_REGS 42  // (rounds 1-3 and early round 4: 41; the 42nd frees 8 KiB of LDS for the narrow head's weights, below)
#endif
constexpr int kR8RegFrags0 = R8_REG0;                  // fragments of the first tile kept in registers ...
constexpr int kR8RegFrags1 = R8_REGS - R8_REG0 < 27 ? R8_REGS - R8_REG0 : 27;  // ... and of the second
static_assert(kR8RegFrags1 <= 27 && kR8RegFrags0 <= 27, "fragments per tile");
constexpr int kR8LdsFrags0 = 27 - kR8RegFrags0, kR8LdsFrags1 = 27 - kR8RegFrags1;  // the rest, in LDS (12 per wave in all)
constexpr int kR8Lds = 2 * PBF16::NBH * 1024 + kR8Waves * (kR8LdsFrags0 + kR8LdsFrags1) * 1024 + 27 * 1024;
constexpr int kR8HeadLds = PBF16::NBH * 1024;          // (kYHead) the narrow head's weight fragments, n-tile 0

// MFMAs of one unit tile whose fragments [first_lds, 27) live in LDS at wl[(i - first_lds)] (i = k_block * 3 + gate) and
// the rest in wreg[i]; LDS fragments go through a kQ-deep register queue
// kChain: a fourth accumulator rides along, one link per k-block with the same A fragment -- w16[blk * kChainStride * 64]
// The operand's A fragments are requested D k-blocks AHEAD of the MFMAs that multiply them (round 6: with the read issued right in
// front of its three MFMAs every k-block exposed an LDS round trip; -1.5 ... -2 % per launch, profiles/r06_antiphase.txt section 2)
#ifndef R8_APF
#define R8_APF 2  // gru_resident8_kernel<false> (six of the bench's eight recurrent launches): fits 256 registers without a spill
#endif
#ifndef R8_APF_Y
#define R8_APF_Y 0  // gru_resident8_kernel<true>: the head's chain and epilogue leave no registers (a distance of 1 spills inside the step)
#endif
template <int kFirstLds, int kQ, int kNReg, bool kChain = false, int kChainStride = 3, int D = 0>
__device__ __forceinline__ void r8_tile_mma(f32x4 (&acc)[3], const bf16x8 *ha, const bf16x8 (&wreg)[kNReg], const bf16x8 *wl,
                                            int lane, f32x4 *a16 = nullptr, const bf16x8 *w16 = nullptr) {
    constexpr int N = 27, NB = PBF16::NBH;
    bf16x8 qb[kQ], qc, aq[D + 1];
    // (the order of these first requests decides hipcc's register allocation of the whole step: A first with D > 0, last with D = 0 --
    // the other order spills 20-32 bytes inside the step in either form)
    if (D > 0) {
#pragma unroll
        for (int p = 0; p <= D; ++p) aq[p] = ha[p * 64 + lane];
    }
#pragma unroll
    for (int p = 0; p < kQ; ++p)
        if (kFirstLds + p < N) qb[p] = wl[p * 64 + lane];
    if (kChain) qc = w16[0];
    if (D == 0) aq[0] = ha[lane];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int blk = i / 3;
        if (i % 3 == 0 && blk > 0 && blk + D < NB) aq[(blk + D) % (D + 1)] = ha[(blk + D) * 64 + lane];
        const bf16x8 a = aq[blk % (D + 1)];
        bf16x8 b;
        if (i < kFirstLds) {
            b = wreg[i < kNReg ? i : 0];
        } else {
            const int j = i - kFirstLds;
            b = qb[j % kQ];
            if (i + kQ < N) qb[j % kQ] = wl[(j + kQ) * 64 + lane];
        }
        acc[i % 3] = PBF16::mma(a, b, acc[i % 3]);
        if (kChain && i % 3 == 2) {  // one link of unit tile 16's k chain per k-block, same A fragment
            const bf16x8 c = qc;
            if (blk + 1 < NB) qc = w16[(blk + 1) * kChainStride * 64];
            *a16 = PBF16::mma(a, c, *a16);
        }
    }
}

__device__ __forceinline__ f16x4 buf_load_gi(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, kGiStreamAux));
}

// kYHead: the stage's narrow head rides along (GruArgs::yw ...): wave 0 -- the wave with the most slack before the step's barrier
// (profiles/r04_recurrent_stamps.txt) -- carries its nine MFMAs through its first tile's loop as a fourth accumulator, on the same A
// fragments (the image of h_{t-1} it multiplies anyway), then the sigmoid and sixteen 2-byte stores: y_{t-1} leaves one step late,
// y_{T-1} after the loop.  The chain is k-ascending from 0 with the bias after, like gemm_head_kernel's: the same bits.
template <bool kYHead>
__global__ __launch_bounds__(64 * kR8Waves, 2) void gru_resident8_kernel(GruArgs g) {
    typedef PBF16 P;
    typedef P::frag_t frag_t;
    constexpr int NBH = P::NBH;
    constexpr int kApf = kYHead ? R8_APF_Y : R8_APF;  // A fragments requested this many k-blocks ahead
    __shared__ __attribute__((aligned(16))) char smem[kR8Lds + 3 * 1024 + 16 + (kYHead ? kR8HeadLds : 0)];
    frag_t *wlh = (frag_t *) (smem + kR8Lds + 3 * 1024 + 16);  // (kYHead) [9][64]: the head's n-tile 0
    char *hbuf0 = smem, *hbuf1 = smem + NBH * 1024;
    frag_t *wl1 = (frag_t *) (smem + 2 * NBH * 1024);                                   // [8 waves][13][64]: first tile's, then second tile's
    frag_t *wl16 = (frag_t *) (smem + 2 * NBH * 1024 + kR8Waves * (kR8LdsFrags0 + kR8LdsFrags1) * 1024);  // [27][64], i = blk * 3 + gate
    f32x4 *acc16 = (f32x4 *) (smem + kR8Lds);  // [3 gates][64 lanes]: unit tile 16's accumulators, handed across waves

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x;
    const int colq = lane & 15, rowq = (lane >> 4) * 4;
    const frag_t *whh = (const frag_t *) g.whh;
    const int u0 = wave, u1 = wave + 8, u2 = 16;
    // Unit tile 16 (the 17th) would make one wave's serial chain 3 tiles long while the others wait at the barrier.
    // The first wave of each SIMD gets the SIMD's issue slots first and is through its two tiles ~1 000 cycles before its
    // partner (per-wave stamps, tools/timing.py), so tile 16 is done in that slack: waves 1, 2, 3 carry its 27 MFMAs (one
    // gate each, the full k chain in one accumulator, so the arithmetic is unchanged) through their second tile's k loop
    // as a fourth accumulator, the accumulators cross LDS behind a step-count flag, and waves 0..3 each do the gate math of
    // one of the four rows a lane owns.  One barrier per step.
#ifndef R8_CHAIN0
#define R8_CHAIN0 1
#endif
#ifndef R8_ROW0
#define R8_ROW0 0
#endif
    const int g16 = wave - R8_CHAIN0;  // gate whose tile-16 MFMAs this wave computes (waves R8_CHAIN0 .. R8_CHAIN0 + 2)
    const bool c16 = wave >= R8_CHAIN0 && wave <= R8_CHAIN0 + 2;
    const bool q16 = wave >= R8_ROW0 && wave < R8_ROW0 + 4;  // this wave finishes row (lane >> 4) * 4 + (wave - R8_ROW0) of tile 16
    // flags accessed with explicit ds instructions: a volatile access or a workgroup fence would make hipcc drain every
    // outstanding global load of the wave (s_waitcnt vmcnt(0)) first
    const unsigned flag16 = (unsigned) (uintptr_t) (smem + kR8Lds + 3 * 1024);  // [3 gates]: step count of acc16's content

    // ---- prologue: everything is requested before anything is waited for (one memory round trip per launch; with the LDS-resident
    // fragments, the biases, the state and the first pre-activations fetched one group after the other it was eight).  The
    // fragments that live in LDS go there directly (global -> LDS, a lane's 16 bytes at the wave-uniform address + 16 lane).
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    frag_t *wl0w = wl1 + wave * (kR8LdsFrags0 + kR8LdsFrags1) * 64, *wl1w = wl0w + kR8LdsFrags0 * 64;
#pragma unroll
    for (int i = kR8RegFrags0; i < 27; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t) (whh + ((size_t) (u0 * 3 + i % 3) * NBH + i / 3) * 64 + lane),
                                         (lptr_t) (wl0w + (i - kR8RegFrags0) * 64), 16, 0, 0);
#pragma unroll
    for (int i = kR8RegFrags1; i < 27; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t) (whh + ((size_t) (u1 * 3 + i % 3) * NBH + i / 3) * 64 + lane),
                                         (lptr_t) (wl1w + (i - kR8RegFrags1) * 64), 16, 0, 0);
    for (int i = wave; i < 27; i += kR8Waves)
        __builtin_amdgcn_global_load_lds((gptr_t) (whh + ((size_t) (u2 * 3 + i % 3) * NBH + i / 3) * 64 + lane),
                                         (lptr_t) (wl16 + i * 64), 16, 0, 0);
    if (kYHead) {
        for (int i = wave; i < NBH; i += kR8Waves)
            __builtin_amdgcn_global_load_lds((gptr_t) ((const frag_t *) g.yw + (size_t) i * 64 + lane), (lptr_t) (wlh + i * 64), 16, 0, 0);
    }
    const float ybias = kYHead ? g.yb[colq] : 0.0f;
    frag_t w0[kR8RegFrags0], w1[kR8RegFrags1];
#pragma unroll
    for (int i = 0; i < kR8RegFrags0; ++i) w0[i] = whh[((size_t) (u0 * 3 + i % 3) * NBH + i / 3) * 64 + lane];
#pragma unroll
    for (int i = 0; i < kR8RegFrags1; ++i) w1[i] = whh[((size_t) (u1 * 3 + i % 3) * NBH + i / 3) * 64 + lane];
    f32x4 hreg[2];
    hreg[0] = ((const f32x4 *) g.hstate_in)[((size_t) mt * kUnitTiles + u0) * 64 + lane];
    hreg[1] = ((const f32x4 *) g.hstate_in)[((size_t) mt * kUnitTiles + u1) * 64 + lane];
    const int e16 = q16 ? wave - R8_ROW0 : 0;  // element of the f32x4 this wave owns in tile 16
    float h16 = g.hstate_in[(((size_t) mt * kUnitTiles + u2) * 64 + lane) * 4 + e16];
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
    for (int i = tid; i < 2 * NBH * 64; i += 64 * kR8Waves) ((uint4 *) smem)[i] = uint4{0, 0, 0, 0};
    if (tid < 4) ((int *) (smem + kR8Lds + 3 * 1024))[tid] = 0;
    __syncthreads();
    // the operand's k = 271, 272 are the constant 1 against the two bias rows of the packed W_hh (kns_layout.h, kBiasK0): no bias
    // fetch and no accumulator splat in the step.  k = 271 (column 15 of tile 16) is rewritten by put_h16 every step, k = 272 is
    // past every tile and stays as set here, in both buffers.
    if (tid < 64) {
        const int k = kBiasK0 + ((tid >> 4) & 1);
        ((uint16_t *) ((tid >> 5) ? hbuf1 : hbuf0))[(k / P::KB) * 64 * P::EPL + P::off(tid & 15, k % P::KB)] = (uint16_t) kBf16One;
    }
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
        dst[0] = colq == 15 ? (uint16_t) kBf16One : f2bf(h);  // (column 15 = k 271: not a hidden unit, the bias rows' constant 1)
    };
    put_h(hbuf0, u0, hreg[0]);
    put_h(hbuf0, u1, hreg[1]);
    if (q16) put_h16(hbuf0, h16);
    __syncthreads();

    // Every wave issues the same vector-memory operations every step and none of them sits inside a branch (the hidden
    // sequence copy is unconditional -- at t = 0 it writes h_{-1} into slot 0, which the same lanes overwrite with h_0 one
    // step later --, block 8 goes out in eighths, and tile 16's pre-activations are requested by all waves although only
    // waves 0..3 use them): with conditional loads and stores in the loop hipcc's first vmcnt wait of a step also covered
    // the pre-activations requested last in the previous step.
    const unsigned lane8 = lane * 8u;
    // The bases of the two per-step streams are ADVANCED by their stride instead of being recomputed from t (hipcc does not
    // strength-reduce the 64-bit (t * mtiles + mt) * bytes products: ~40 scalar instructions per step in a loop whose waves issue
    // an instruction every ~5 cycles): the hidden-sequence slot of step t is max(t - 1, 0), the pre-activation slot min(t + 1, T - 1)
    const size_t hs_stride = (size_t) g.mtiles * NBH * 1024, gi_stride = (size_t) g.mtiles * kGateTiles * 512;
    const char *hs_base = (const char *) g.hseq + (size_t) mt * NBH * 1024;
    const char *gn_base = (const char *) g.gi + (size_t) mt * kGateTiles * 512 + (g.T > 1 ? gi_stride : 0);
    auto publish = [&](const frag_t *src, const char *slot_base) {
        const __amdgpu_buffer_rsrc_t hs = make_rsrc(slot_base, NBH * 1024);
        const unsigned i0 = wave * 64 + lane, i1 = 8 * 64 + wave * 8 + (lane & 7);
        const frag_t x0 = src[i0];
        buf_store_frag(hs, i0 * 16u, x0);
        const frag_t x1 = src[i1];
        buf_store_frag(hs, lane < 8 ? i1 * 16u : 0x7fffff00u, x1);  // lanes 8..63: past the descriptor's end, dropped
    };
    // (kYHead, wave 0) y of the hidden vector whose image the chain just multiplied -> frame `frame` of the destination matrix
    const size_t y_frame_bytes = (size_t) g.mtiles * g.y_nb * 1024;
    // (in a wave-0 branch, against the rule that no vector-memory operation of the step sits inside one: the unconditional form --
    // every wave issuing the stores through a descriptor that is empty for waves 1..7 -- costs registers the kernel does not have
    // (256 VGPRs + 24 B of scratch) and measured 171.7 us per launch against 156.5)
    auto emit_y = [&](const f32x4 &ya, int frame) {
        uint16_t *dst = (uint16_t *) ((char *) g.yout + (size_t) frame * y_frame_bytes + ((size_t) mt * g.y_nb + g.y_blk) * 1024);
        if (colq < g.yvalid) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[P::off(rowq + i, g.y_kk0 + colq)] = P::cvt(head_sigmoid<PBF16>(ya[i] + ybias));
        }
    };
    for (int t = 0; t < g.T; ++t) {
        KNS_STAMP(0);
        KNS_STAMP_AT(9, 8);  // steady-state step length = (stamp 10 - stamp 9) / 16
        KNS_STAMP_AT(10, 24);
        const char *hc = (t & 1) ? hbuf1 : hbuf0;
        char *hn = (t & 1) ? hbuf0 : hbuf1;
        const frag_t *ha = (const frag_t *) hc;
        publish(ha, hs_base);  // LDS holds h_{t-1}
        const __amdgpu_buffer_rsrc_t gnext = make_rsrc(gn_base, kGateTiles * 512);
        hs_base += t > 0 ? hs_stride : 0;
        gn_base += t + 2 < g.T ? gi_stride : 0;

        auto gates = [&](const int q, f32x4 (&acc)[3]) {
            // the fp16 pre-activations enter the gate arithmetic through v_fma_mix_f32 (an f16 operand of an f32 fma): x + t as
            // fma(x, 1, t) and fma(r, q, x) round once, exactly like the add / fma on the converted value
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 vr = __builtin_bit_cast(u32x2, gi[q][0]), vz = __builtin_bit_cast(u32x2, gi[q][1]),
                        vn = __builtin_bit_cast(u32x2, gi[q][2]);
            const unsigned pr[2] = {vr[0], vr[1]}, pz[2] = {vz[0], vz[1]}, pn[2] = {vn[0], vn[1]};
            const int u = q ? u1 : u0;
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) gi[q][gt] = buf_load_gi(gnext, lane8, (u * 3 + gt) * 512u);
            const f32x4 hnew = gate_block_bf16(pr, pz, pn, acc[0], acc[1], acc[2], hreg[q]);
            hreg[q] = hnew;
            put_h(hn, u, hnew);
        };
        // the chains start from the inline constant 0: b_hh rides in the operands (two rows of the packed W_hh against h's constant 1)
        auto acc_init = [&](f32x4 (&acc)[3], const int) {
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
        };
        KNS_STAMP(1);
        f32x4 acc[3];
        acc_init(acc, u0);
        if (kYHead && wave == 0) {  // ... with the head's chain on h_{t-1} (at t = 0: h_{-1}, into frame 0's slot, rewritten at t = 1)
            f32x4 ya = f32x4{0.f, 0.f, 0.f, 0.f};
            r8_tile_mma<kR8RegFrags0, R8C_Q, kR8RegFrags0, true, 1, kApf>(acc, ha, w0, wl0w, lane, &ya, wlh + lane);
            emit_y(ya, t > 0 ? t - 1 : 0);
        } else {
            r8_tile_mma<kR8RegFrags0, R8C_Q, kR8RegFrags0, false, 3, kApf>(acc, ha, w0, wl0w, lane);
        }
        KNS_STAMP(2);
        gates(0, acc);
        KNS_STAMP(3);
        acc_init(acc, u1);
        if (c16) {  // waves 1, 2, 3 also carry one gate of unit tile 16 (k-blocks in order in one accumulator) through this loop
            f32x4 a16 = f32x4{0.f, 0.f, 0.f, 0.f};
            r8_tile_mma<kR8RegFrags1, R8C_Q, kR8RegFrags1, true, 3, kApf>(acc, ha, w1, wl1w, lane, &a16, wl16 + g16 * 64 + lane);
            acc16[g16 * 64 + lane] = a16;
            // LDS operations of one wave complete in order: whoever sees the flag sees the accumulators
            asm volatile("ds_write_b32 %0, %1" ::"v"(flag16 + g16 * 4), "v"(t + 1) : "memory");
        } else {
            r8_tile_mma<kR8RegFrags1, 3, kR8RegFrags1, false, 3, kApf>(acc, ha, w1, wl1w, lane);
        }
        KNS_STAMP(4);
        gates(1, acc);
        KNS_STAMP(5);
        KNS_STAMP(6);
        {  // (requested by every wave, used by waves 0..3)
            const float xr = (float) gi16[0][e16], xz = (float) gi16[1][e16], xn = (float) gi16[2][e16];
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) gi16[gt] = buf_load_gi(gnext, lane8, (u2 * 3 + gt) * 512u);
            if (q16) {  // waves 0..3: row e16 of every lane's four rows of unit tile 16
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                i32x4 f;
                do {
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(flag16) : "memory");
                } while (__builtin_amdgcn_readfirstlane(f[0] + f[1] + f[2]) != 3 * (t + 1));
                const float ar = ((const float *) acc16)[(0 * 64 + lane) * 4 + e16];
                const float az = ((const float *) acc16)[(1 * 64 + lane) * 4 + e16];
                const float an = ((const float *) acc16)[(2 * 64 + lane) * 4 + e16];
                h16 = gate_elem_bf16(xr, xz, xn, ar, az, an, h16);
                put_h16(hn, h16);
            }
        }
        KNS_STAMP(7);
        __syncthreads();
        KNS_STAMP(8);
    }
    if (kYHead && wave == 0) {  // the head of the last hidden vector
        const frag_t *hl = (const frag_t *) ((g.T & 1) ? hbuf1 : hbuf0);
        f32x4 ya = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < NBH; ++blk) ya = P::mma(hl[blk * 64 + lane], wlh[blk * 64 + lane], ya);
        emit_y(ya, g.T - 1);
    }
    publish((const frag_t *) ((g.T & 1) ? hbuf1 : hbuf0), (const char *) g.hseq + ((size_t) (g.T - 1) * g.mtiles + mt) * NBH * 1024);
    ((f32x4 *) g.hstate_out)[((size_t) mt * kUnitTiles + u0) * 64 + lane] = hreg[0];
    ((f32x4 *) g.hstate_out)[((size_t) mt * kUnitTiles + u1) * 64 + lane] = hreg[1];
    if (q16) g.hstate_out[(((size_t) mt * kUnitTiles + u2) * 64 + lane) * 4 + e16] = h16;
}

// (Round 6 measured a form WITHOUT the step's barrier -- per-tile step counters in LDS, the first MFMA loop waiting in front of k-blocks 0
// and 6, which runs the two waves of a SIMD one window apart by construction: bit-identical, 171-175 us per launch against 155, MFMA / VALU
// co-execution 16.7 % against 18.5 %; last commit that holds it: b283d69, record: profiles/r06_antiphase.txt.)
// (Round 5 measured a form of the resident kernel that runs TWO m-tiles per workgroup -- the consumer half of a producer/consumer CU
// pair -- as a timing variant, gru_r8x2_kernel: 4 113-4 492 ticks per m-tile-step against 4 714, on half the chip; not adopted.  The
// source left the tree with the experiment: last commit that holds it is a6273b0; record: profiles/r05_r8x2_consumer.txt.  Likewise the
// gates as LDS table lookups instead of v_exp_f32 / v_rcp_f32 (-DKNS_GATE_LUT: 158 -> 196 us per launch; last commit ba172d7,
// profiles/r05_gate_lut.txt).)
void launch_gru(const GruArgs &a, hipStream_t s) {
    const bool stream_weights = (a.dev & kDevGruStream) != 0;  // A/B switch (developer build only)
    if (a.precision == kBf16 && !stream_weights && a.yw)
        hipLaunchKernelGGL(gru_resident8_kernel<true>, dim3(a.mtiles), dim3(64 * kR8Waves), 0, s, a);
    else if (a.precision == kBf16 && !stream_weights)
        hipLaunchKernelGGL(gru_resident8_kernel<false>, dim3(a.mtiles), dim3(64 * kR8Waves), 0, s, a);
    else if (a.precision == kBf16)
        hipLaunchKernelGGL((gru_kernel<PBF16, 8>), dim3(a.mtiles), dim3(512), 0, s, a);
    else
        hipLaunchKernelGGL((gru_kernel<PF32, 8>), dim3(a.mtiles), dim3(512), 0, s, a);
}

#ifdef KNS_TIMING
void read_timing(unsigned long long *out) { (void) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kns_timing), sizeof(unsigned long long) * 8 * 16); }
#endif


}  // namespace kns
