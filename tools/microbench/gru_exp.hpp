// gru_exp.hpp -- experimental variants of the resident recurrent kernel (developer harness only; included by gru_bench.hip
// after kns_gru.hip).  Flags: 1 = pinned operand queues in the MFMA loops, 2 = static s_setprio(1) for waves 0..3,
// 4 = static s_setprio(1) for waves 4..7, 8 = plain (non-packed) f32 gate math, 16 = s_setprio(1) around MFMA phases of waves 0..3
#pragma once

template <int kFirstLds, int kQ, int kNReg, bool kChain, bool kPin>
__device__ __forceinline__ void x_tile_mma(f32x4 (&acc)[3], const bf16x8 *ha, const bf16x8 (&wreg)[kNReg], const bf16x8 *wl,
                                            int lane, f32x4 *a16 = nullptr, const bf16x8 *w16 = nullptr) {
    constexpr int N = 27;
    bf16x8 qb[kQ], qc;
#pragma unroll
    for (int p = 0; p < kQ; ++p)
        if (kFirstLds + p < N) qb[p] = wl[p * 64 + lane];
    if (kChain) qc = w16[0];
    if (!kPin) {
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
            if (kChain && i % 3 == 2) {
                const bf16x8 c = qc;
                if (i / 3 + 1 < PBF16::NBH) qc = w16[(i / 3 + 1) * 3 * 64];
                *a16 = PBF16::mma(a, c, *a16);
            }
        }
        return;
    }
    // pinned form: A fragments two k-blocks ahead in three rotating registers, LDS weight fragments kQ MFMAs ahead; a
    // scheduling fence after every MFMA keeps hipcc from sinking the reads down to their uses
    bf16x8 aq[3];
    aq[0] = ha[lane];
    aq[1] = ha[64 + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int blk = i / 3;
        if (i % 3 == 0 && blk + 2 < PBF16::NBH) aq[(blk + 2) % 3] = ha[(blk + 2) * 64 + lane];
        bf16x8 b;
        if (i < kFirstLds) {
            b = wreg[i < kNReg ? i : 0];
        } else {
            const int j = i - kFirstLds;
            b = qb[j % kQ];
            if (i + kQ < N) qb[j % kQ] = wl[(j + kQ) * 64 + lane];
        }
        acc[i % 3] = PBF16::mma(aq[blk % 3], b, acc[i % 3]);
        if (kChain && i % 3 == 2) {
            const bf16x8 c = qc;
            if (blk + 1 < PBF16::NBH) qc = w16[(blk + 1) * 3 * 64];
            *a16 = PBF16::mma(aq[blk % 3], c, *a16);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ f16x4 x_buf_load_gi(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}

template <int F>
__global__ __launch_bounds__(64 * kR8Waves, 2) void gru_x_kernel(GruArgs g) {
    constexpr bool kPin = (F & 1) != 0;
    typedef PBF16 P;
    typedef P::frag_t frag_t;
    constexpr int NBH = P::NBH;
    __shared__ __attribute__((aligned(16))) char smem[kR8Lds + 3 * 1024 + 16];
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
    // The first wave of each SIMD gets the SIMD's issue slots first and is through its two tiles ~1 000 cycles before its
    // partner (per-wave stamps, tools/timing.py), so tile 16 is done in that slack: waves 1, 2, 3 carry its 27 MFMAs (one
    // gate each, the full k chain in one accumulator, so the arithmetic is unchanged) through their second tile's k loop
    // as a fourth accumulator, the accumulators cross LDS behind a step-count flag, and waves 0..3 each do the gate math of
    // one of the four rows a lane owns.  One barrier per step.
    const int g16 = wave - 1;         // gate whose tile-16 MFMAs this wave computes (waves 1..3)
    const bool c16 = wave >= 1 && wave <= 3;
    const bool q16 = wave < 4;        // this wave finishes row (lane >> 4) * 4 + wave of tile 16
    // flags accessed with explicit ds instructions: a volatile access or a workgroup fence would make hipcc drain every
    // outstanding global load of the wave (s_waitcnt vmcnt(0)) first
    const unsigned flag16 = (unsigned) (uintptr_t) (smem + kR8Lds + 3 * 1024);  // [3 gates]: step count of acc16's content

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
    if (tid < 4) ((int *) (smem + kR8Lds + 3 * 1024))[tid] = 0;
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
    _Float16 gs16[3];
    {
        const _Float16 *gp = (const _Float16 *) g.gi + (size_t) mt * kGateTiles * 256;
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) gs16[gt] = gp[((u2 * 3 + gt) * 64 + lane) * 4 + e16];
    }
    __syncthreads();

    // Every wave issues the same vector-memory operations every step and none of them sits inside a branch (the hidden
    // sequence copy is unconditional -- at t = 0 it writes h_{-1} into slot 0, which the same lanes overwrite with h_0 one
    // step later --, block 8 goes out in eighths, and tile 16's pre-activations are requested by all waves although only
    // waves 0..3 use them): with conditional loads and stores in the loop hipcc's first vmcnt wait of a step also covered
    // the pre-activations requested last in the previous step.
    const unsigned lane8 = lane * 8u;
    auto publish = [&](const frag_t *src, int slot) {
        const __amdgpu_buffer_rsrc_t hs = make_rsrc((const frag_t *) g.hseq + ((size_t) slot * g.mtiles + mt) * NBH * 64, NBH * 1024);
        const unsigned i0 = wave * 64 + lane, i1 = 8 * 64 + wave * 8 + (lane & 7);
        const frag_t x0 = src[i0];
        buf_store_frag(hs, i0 * 16u, x0);
        const frag_t x1 = src[i1];
        buf_store_frag(hs, lane < 8 ? i1 * 16u : 0x7fffff00u, x1);  // lanes 8..63: past the descriptor's end, dropped
    };
    if ((F & 2) && wave < 4) __builtin_amdgcn_s_setprio(1);
    if ((F & 4) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    for (int t = 0; t < g.T; ++t) {
        KNS_STAMP(0);
        KNS_STAMP_AT(9, 8);  // steady-state step length = (stamp 10 - stamp 9) / 16
        KNS_STAMP_AT(10, 24);
        const char *hc = (t & 1) ? hbuf1 : hbuf0;
        char *hn = (t & 1) ? hbuf0 : hbuf1;
        const frag_t *ha = (const frag_t *) hc;
        if (!(F & 256) && !(F & 4096)) publish(ha, t > 0 ? t - 1 : 0);  // LDS holds h_{t-1}
        const __amdgpu_buffer_rsrc_t gnext = make_rsrc(
            (const P::gi_t *) g.gi + ((size_t) (t + 1 < g.T ? t + 1 : t) * g.mtiles + mt) * kGateTiles * 64, kGateTiles * 512);

        auto gates = [&](const int q, const int u, f32x4 (&acc)[3]) {
            if (F & 32) {  // ablation: no gate math (results are garbage)
                if (!(F & 128)) {
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) gi[q][gt] = x_buf_load_gi(gnext, lane8, (u * 3 + gt) * 512u);
                }
                f32x4 hn2 = acc[0] + acc[1] + acc[2] + P::from_gi(gi[q][0]);
                hreg[q] = hn2;
                put_h(hn, u, hn2);
                return;
            }
            const f32x4 ir = P::from_gi(gi[q][0]), iz = P::from_gi(gi[q][1]), in = P::from_gi(gi[q][2]);
#pragma unroll
            for (int gt = 0; gt < 3; ++gt)
                if (!(F & 128)) gi[q][gt] = x_buf_load_gi(gnext, lane8, (u * 3 + gt) * 512u);
            const float br = lbias[(u * 3 + 0) * 16 + colq], bz = lbias[(u * 3 + 1) * 16 + colq],
                        bn = lbias[(u * 3 + 2) * 16 + colq];
            const f32x2 vbr = {br, br}, vbz = {bz, bz}, vbn = {bn, bn};
            f32x4 hnew;
            if (F & 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((ir[i] + (acc[0][i] + br)) * -1.44269504088896341f));
                    const float z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((iz[i] + (acc[1][i] + bz)) * -1.44269504088896341f));
                    const float rr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((r * (acc[2][i] + bn) + in[i]) * 2.88539008177792681f));
                    const float n = 1.0f - (rr + rr);
                    hnew[i] = z * (hreg[q][i] - n) + n;
                }
            } else {
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
            }
            hreg[q] = hnew;
            put_h(hn, u, hnew);
        };

        if (((F >> 16) & 15) && wave >= 4) __builtin_amdgcn_s_sleep((F >> 16) & 15);
        if (F & 8192) {
            // restructured step: unit tile 16's k-chains ride through the FIRST tile's loop of waves 1..3 (all of that tile's
            // weights are in registers, so the chain's LDS reads are the only ones besides A), which makes its accumulators
            // available a whole phase before waves 0..3 need them; those waves finish tile 16 inside their second gate block
            f32x4 acc[3];
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c16) {
                f32x4 a16 = f32x4{0.f, 0.f, 0.f, 0.f};
                x_tile_mma<27, 1, 27, true, kPin>(acc, ha, w0, wl16, lane, &a16, wl16 + g16 * 64 + lane);
                acc16[g16 * 64 + lane] = a16;
                asm volatile("ds_write_b32 %0, %1" ::"v"(flag16 + g16 * 4), "v"(t + 1) : "memory");
            } else {
                x_tile_mma<27, 1, 27, false, kPin>(acc, ha, w0, wl16, lane);
            }
            gates(0, u0, acc);
            if ((F & 4096) && !(F & 16384)) publish(ha, t > 0 ? t - 1 : 0);
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
            x_tile_mma<kR8RegFrags1, 3, kR8RegFrags1, false, kPin>(acc, ha, w1, wl1w, lane);
            if ((F & 4096) && (F & 16384)) publish(ha, t > 0 ? t - 1 : 0);
            // this wave's row of unit tile 16: one fp16 per gate and lane (requested by every wave, used by waves 0..3)
            const float xr = (float) gs16[0], xz = (float) gs16[1], xn = (float) gs16[2];
#pragma unroll
            for (int gt = 0; gt < 3; ++gt)
                gs16[gt] = __builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(gnext, lane8 + e16 * 2u, (u2 * 3 + gt) * 512u, 0));
            if (q16) {
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                i32x4 f;
                do {
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(flag16) : "memory");
                } while (__builtin_amdgcn_readfirstlane(f[0] + f[1] + f[2]) != 3 * (t + 1));
                const float ar = ((const float *) acc16)[(0 * 64 + lane) * 4 + e16];
                const float az = ((const float *) acc16)[(1 * 64 + lane) * 4 + e16];
                const float an = ((const float *) acc16)[(2 * 64 + lane) * 4 + e16];
                const float br = lbias[(u2 * 3 + 0) * 16 + colq], bz = lbias[(u2 * 3 + 1) * 16 + colq],
                            bn = lbias[(u2 * 3 + 2) * 16 + colq];
                gates(1, u1, acc);
                const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((xr + (ar + br)) * -1.44269504088896341f));
                const float z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((xz + (az + bz)) * -1.44269504088896341f));
                const float rr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((r * (an + bn) + xn) * 2.88539008177792681f));
                const float n = 1.0f - (rr + rr);
                h16 = z * (h16 - n) + n;
                put_h16(hn, h16);
            } else {
                gates(1, u1, acc);
            }
        } else {
        KNS_STAMP(1);
        f32x4 acc[3];
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if ((F & 16) && wave < 4) __builtin_amdgcn_s_setprio(1);
        if (!(F & 64)) x_tile_mma<27, 1, 27, false, kPin>(acc, ha, w0, wl16, lane);
        if ((F & 16) && wave < 4) __builtin_amdgcn_s_setprio(0);
        KNS_STAMP(2);
        gates(0, u0, acc);
        if (F & 4096) publish(ha, t > 0 ? t - 1 : 0);
        KNS_STAMP(3);
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if ((F & 1024) && c16) {  // ablation: no chain MFMAs, flags still raised
            x_tile_mma<kR8RegFrags1, 3, kR8RegFrags1, false, kPin>(acc, ha, w1, wl1w, lane);
            acc16[g16 * 64 + lane] = acc[0];
            asm volatile("ds_write_b32 %0, %1" ::"v"(flag16 + g16 * 4), "v"(t + 1) : "memory");
        } else if (c16 && !(F & 512)) {  // waves 1, 2, 3 also carry one gate of unit tile 16 (k-blocks in order in one accumulator) through this loop
            f32x4 a16 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!(F & 64)) x_tile_mma<kR8RegFrags1, R8C_Q, kR8RegFrags1, true, kPin>(acc, ha, w1, wl1w, lane, &a16, wl16 + g16 * 64 + lane);
            acc16[g16 * 64 + lane] = a16;
            // LDS operations of one wave complete in order: whoever sees the flag sees the accumulators
            asm volatile("ds_write_b32 %0, %1" ::"v"(flag16 + g16 * 4), "v"(t + 1) : "memory");
        } else {
            if ((F & 16) && wave < 4) __builtin_amdgcn_s_setprio(1);
            if (!(F & 64)) x_tile_mma<kR8RegFrags1, 3, kR8RegFrags1, false, kPin>(acc, ha, w1, wl1w, lane);
            if ((F & 16) && wave < 4) __builtin_amdgcn_s_setprio(0);
        }
        KNS_STAMP(4);
        gates(1, u1, acc);
        KNS_STAMP(5);
        KNS_STAMP(6);
        {  // (requested by every wave, used by waves 0..3)
            const float xr = (float) gi16[0][e16], xz = (float) gi16[1][e16], xn = (float) gi16[2][e16];
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) gi16[gt] = x_buf_load_gi(gnext, lane8, (u2 * 3 + gt) * 512u);
            if (q16 && !(F & 512) && !(F & 2048)) {  // waves 0..3: row e16 of every lane's four rows of unit tile 16
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                i32x4 f;
                do {
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(flag16) : "memory");
                } while (__builtin_amdgcn_readfirstlane(f[0] + f[1] + f[2]) != 3 * (t + 1));
                const float ar = ((const float *) acc16)[(0 * 64 + lane) * 4 + e16];
                const float az = ((const float *) acc16)[(1 * 64 + lane) * 4 + e16];
                const float an = ((const float *) acc16)[(2 * 64 + lane) * 4 + e16];
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
        }
        }
        KNS_STAMP(7);
        __syncthreads();
        KNS_STAMP(8);
    }
    publish((const frag_t *) ((g.T & 1) ? hbuf1 : hbuf0), g.T - 1);
    ((f32x4 *) g.hstate_out)[((size_t) mt * kUnitTiles + u0) * 64 + lane] = hreg[0];
    ((f32x4 *) g.hstate_out)[((size_t) mt * kUnitTiles + u1) * 64 + lane] = hreg[1];
    if (q16) g.hstate_out[(((size_t) mt * kUnitTiles + u2) * 64 + lane) * 4 + e16] = h16;
}


// ------------------------------------------------------------------------------------------------------------------
// 16-wave form: four waves per SIMD (128 registers each) so that a wave waiting for an LDS operand, a transcendental or
// the step barrier leaves its SIMD to three others -- latency is hidden by the hardware's wave interleave instead of by
// software queues the register budget has no room for.  Wave w owns unit tile w: its first kReg B-fragments in VGPRs,
// the rest in LDS; unit tile 16 lives in LDS, its three k-chains ride on waves 4..6, its gate math on waves 0..3.
template <int kReg, int F>
__global__ __launch_bounds__(1024, 4) void gru_r16_kernel(GruArgs g) {
    typedef PBF16 P;
    typedef P::frag_t frag_t;
    constexpr int NBH = P::NBH, kW = 16, kLdsFrags = 27 - kReg;
    constexpr int kOffW = 2 * NBH * 1024, kOffW16 = kOffW + kW * kLdsFrags * 1024, kOffAcc = kOffW16 + 27 * 1024,
                  kOffFlag = kOffAcc + 3 * 1024, kTotal = kOffFlag + 16;
    static_assert(kTotal <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char smem[kTotal];
    char *hbuf0 = smem, *hbuf1 = smem + NBH * 1024;
    frag_t *wl16 = (frag_t *) (smem + kOffW16);  // [27][64], i = blk * 3 + gate
    f32x4 *acc16 = (f32x4 *) (smem + kOffAcc);   // [3 gates][64 lanes]
    const unsigned flag16 = (unsigned) (uintptr_t) (smem + kOffFlag);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x;
    const int colq = lane & 15, rowq = (lane >> 4) * 4;
    const frag_t *whh = (const frag_t *) g.whh;
    const int u0 = wave, u2 = 16;
    const bool c16 = wave >= 4 && wave <= 6;  // carries gate (wave - 4) of unit tile 16 through its k loop
    const int g16 = c16 ? wave - 4 : 0;
    const bool q16 = wave < 4;                // finishes row (lane >> 4) * 4 + wave of unit tile 16
    const int e16 = q16 ? wave : 0;

    // ---- prologue: weights
    frag_t wr[kReg];
#pragma unroll
    for (int i = 0; i < kReg; ++i) wr[i] = whh[((size_t) (u0 * 3 + i % 3) * NBH + i / 3) * 64 + lane];
    frag_t *wlw = (frag_t *) (smem + kOffW) + wave * kLdsFrags * 64;
    for (int i = kReg; i < 27; ++i) wlw[(i - kReg) * 64 + lane] = whh[((size_t) (u0 * 3 + i % 3) * NBH + i / 3) * 64 + lane];
    for (int i = wave; i < 27; i += kW) wl16[i * 64 + lane] = whh[((size_t) (u2 * 3 + i % 3) * NBH + i / 3) * 64 + lane];
    const float br = g.bhh[(u0 * 3 + 0) * 16 + colq], bz = g.bhh[(u0 * 3 + 1) * 16 + colq], bn = g.bhh[(u0 * 3 + 2) * 16 + colq];
    const float br16 = g.bhh[(u2 * 3 + 0) * 16 + colq], bz16 = g.bhh[(u2 * 3 + 1) * 16 + colq],
                bn16 = g.bhh[(u2 * 3 + 2) * 16 + colq];

    f32x4 hreg = ((const f32x4 *) g.hstate_in)[((size_t) mt * kUnitTiles + u0) * 64 + lane];
    float h16 = g.hstate_in[(((size_t) mt * kUnitTiles + u2) * 64 + lane) * 4 + e16];
    for (int i = tid; i < 2 * NBH * 64; i += 64 * kW) ((uint4 *) smem)[i] = uint4{0, 0, 0, 0};
    if (tid < 4) ((int *) (smem + kOffFlag))[tid] = 0;
    __syncthreads();
    auto put_h = [&](char *buf, int u, const f32x4 &h) {
        const int k = u * 16 + colq;
        uint16_t *dst = (uint16_t *) buf + (k / P::KB) * 64 * P::EPL + P::off(rowq, k % P::KB);
        const uint32_t lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{h[0], h[1]}, bf16x2));
        const uint32_t hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{h[2], h[3]}, bf16x2));
        dst[0] = (uint16_t) lo;
        dst[8] = (uint16_t) (lo >> 16);
        dst[16] = (uint16_t) hi;
        dst[24] = (uint16_t) (hi >> 16);
    };
    auto put_h16 = [&](char *buf, float h) {
        const int k = u2 * 16 + colq;
        uint16_t *dst = (uint16_t *) buf + (k / P::KB) * 64 * P::EPL + P::off(rowq + e16, k % P::KB);
        dst[0] = f2bf(h);
    };
    put_h(hbuf0, u0, hreg);
    if (q16) put_h16(hbuf0, h16);

    P::gi_t gi[3], gi16[3];
    {
        const P::gi_t *gp = (const P::gi_t *) g.gi + (size_t) mt * kGateTiles * 64;
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) {
            gi[gt] = gp[(u0 * 3 + gt) * 64 + lane];
            gi16[gt] = gp[(u2 * 3 + gt) * 64 + lane];
        }
    }
    __syncthreads();

    const unsigned lane8 = lane * 8u;
    if ((F & 2) && wave < 4) __builtin_amdgcn_s_setprio(1);
    for (int t = 0; t < g.T; ++t) {
        KNS_STAMP(0);
        KNS_STAMP_AT(9, 8);
        KNS_STAMP_AT(10, 24);
        const char *hc = (t & 1) ? hbuf1 : hbuf0;
        char *hn = (t & 1) ? hbuf0 : hbuf1;
        const frag_t *ha = (const frag_t *) hc;
        {  // LDS holds h_{t-1}: waves 0..8 publish one k-block each as the next layer's A operand
            const int slot = t > 0 ? t - 1 : 0;
            const __amdgpu_buffer_rsrc_t hs = make_rsrc((const frag_t *) g.hseq + ((size_t) slot * g.mtiles + mt) * NBH * 64,
                                                        wave < NBH ? NBH * 1024 : 0);
            const unsigned i0 = (wave < NBH ? wave : 0) * 64 + lane;
            buf_store_frag(hs, i0 * 16u, ha[i0]);
        }
        const __amdgpu_buffer_rsrc_t gnext = make_rsrc(
            (const P::gi_t *) g.gi + ((size_t) (t + 1 < g.T ? t + 1 : t) * g.mtiles + mt) * kGateTiles * 64, kGateTiles * 512);
        f32x4 acc[3];
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c16) {
            f32x4 a16 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 27; ++i) {
                const frag_t a = ha[(i / 3) * 64 + lane];
                const frag_t b = i < kReg ? wr[i < kReg ? i : 0] : wlw[(i - kReg) * 64 + lane];
                acc[i % 3] = P::mma(a, b, acc[i % 3]);
                if (i % 3 == 2) a16 = P::mma(a, wl16[((i / 3) * 3 + g16) * 64 + lane], a16);
            }
            acc16[g16 * 64 + lane] = a16;
            asm volatile("ds_write_b32 %0, %1" ::"v"(flag16 + g16 * 4), "v"(t + 1) : "memory");
        } else {
#pragma unroll
            for (int i = 0; i < 27; ++i) {
                const frag_t a = ha[(i / 3) * 64 + lane];
                const frag_t b = i < kReg ? wr[i < kReg ? i : 0] : wlw[(i - kReg) * 64 + lane];
                acc[i % 3] = P::mma(a, b, acc[i % 3]);
            }
        }
        {
            const f32x4 ir = P::from_gi(gi[0]), iz = P::from_gi(gi[1]), in = P::from_gi(gi[2]);
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) gi[gt] = x_buf_load_gi(gnext, lane8, (u0 * 3 + gt) * 512u);
            f32x4 hnew;
            if (F & 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((ir[i] + (acc[0][i] + br)) * -1.44269504088896341f));
                    const float z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((iz[i] + (acc[1][i] + bz)) * -1.44269504088896341f));
                    const float rr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((r * (acc[2][i] + bn) + in[i]) * 2.88539008177792681f));
                    const float n = 1.0f - (rr + rr);
                    hnew[i] = z * (hreg[i] - n) + n;
                }
            } else {
                const f32x2 vbr = {br, br}, vbz = {bz, bz}, vbn = {bn, bn};
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const f32x2 ar = {acc[0][2 * p], acc[0][2 * p + 1]}, az = {acc[1][2 * p], acc[1][2 * p + 1]},
                                an = {acc[2][2 * p], acc[2][2 * p + 1]};
                    const f32x2 xr = {ir[2 * p], ir[2 * p + 1]}, xz = {iz[2 * p], iz[2 * p + 1]}, xn = {in[2 * p], in[2 * p + 1]};
                    const f32x2 r = fast_sigmoid2(xr + (ar + vbr));
                    const f32x2 z = fast_sigmoid2(xz + (az + vbz));
                    const f32x2 n = fast_tanh2(r * (an + vbn) + xn);
                    const f32x2 hp = {hreg[2 * p], hreg[2 * p + 1]};
                    const f32x2 h = z * (hp - n) + n;
                    hnew[2 * p] = h[0];
                    hnew[2 * p + 1] = h[1];
                }
            }
            hreg = hnew;
            put_h(hn, u0, hnew);
        }
        {  // unit tile 16: pre-activations requested by every wave (zero-length descriptor for the others), math on waves 0..3
            const float xr = (float) gi16[0][e16], xz = (float) gi16[1][e16], xn = (float) gi16[2][e16];
            const __amdgpu_buffer_rsrc_t g16r = make_rsrc(
                (const P::gi_t *) g.gi + ((size_t) (t + 1 < g.T ? t + 1 : t) * g.mtiles + mt) * kGateTiles * 64, q16 ? kGateTiles * 512 : 0);
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) gi16[gt] = x_buf_load_gi(g16r, lane8, (u2 * 3 + gt) * 512u);
            if (q16) {
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                i32x4 f;
                do {
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(flag16) : "memory");
                } while (__builtin_amdgcn_readfirstlane(f[0] + f[1] + f[2]) != 3 * (t + 1));
                const float ar = ((const float *) acc16)[(0 * 64 + lane) * 4 + e16];
                const float az = ((const float *) acc16)[(1 * 64 + lane) * 4 + e16];
                const float an = ((const float *) acc16)[(2 * 64 + lane) * 4 + e16];
                const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((xr + (ar + br16)) * -1.44269504088896341f));
                const float z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((xz + (az + bz16)) * -1.44269504088896341f));
                const float rr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((r * (an + bn16) + xn) * 2.88539008177792681f));
                const float n = 1.0f - (rr + rr);
                h16 = z * (h16 - n) + n;
                put_h16(hn, h16);
            }
        }
        KNS_STAMP(7);
        __syncthreads();
        KNS_STAMP(8);
    }
    {
        const __amdgpu_buffer_rsrc_t hs = make_rsrc((const frag_t *) g.hseq + ((size_t) (g.T - 1) * g.mtiles + mt) * NBH * 64,
                                                    wave < NBH ? NBH * 1024 : 0);
        const unsigned i0 = (wave < NBH ? wave : 0) * 64 + lane;
        buf_store_frag(hs, i0 * 16u, ((const frag_t *) ((g.T & 1) ? hbuf1 : hbuf0))[i0]);
    }
    ((f32x4 *) g.hstate_out)[((size_t) mt * kUnitTiles + u0) * 64 + lane] = hreg;
    if (q16) g.hstate_out[(((size_t) mt * kUnitTiles + u2) * 64 + lane) * 4 + e16] = h16;
}
