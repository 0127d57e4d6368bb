// kns_layout.h -- HBM data layouts of the KNS-v1 engine, shared by host packing code and kernels.
//
// Every activation matrix between kernels lives in HBM in the register-fragment order of the gfx950 MFMA that
// consumes or produces it, so that each lane's access is one aligned 16-byte (or 8-byte) word and a wave's
// access is one contiguous 1 KiB (512 B) line group:
//
//   A-packed  (MFMA A operand, rows = stream-frames):  [m_tile][k_block][64 lanes][16 B]
//   B-packed  (MFMA B operand, weights):               [n_tile][k_block][64 lanes][16 B]
//   C-packed  (MFMA C/D fragment, 16x16 tile):         [m_tile][n_tile][64 lanes][4 values]
//
//   bf16 (v_mfma_f32_16x16x32_bf16): k_block = 32 k, lane l element i  <->  row/col = l & 15, k = (l >> 4) * 8 + i
//   fp32 (v_mfma_f32_16x16x4_f32 x4): k_block = 16 k, lane l element q <->  row/col = l & 15, k = q * 4 + (l >> 4)
//   C/D (both):                       lane l value i                   <->  row = (l >> 4) * 4 + i, col = l & 15
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define KNS_HD __host__ __device__ inline
#else
#define KNS_HD inline
#endif

namespace kns {

constexpr int kFrame = 256;
constexpr int kNfft = 512;
constexpr int kBins = 257;
constexpr int kHidden = 271;
constexpr int kUnitTiles = 17;           // ceil(271 / 16)
constexpr int kGateTiles = 3 * kUnitTiles;  // 51 n-tiles of a GRU matrix, ordered [unit_tile][gate r,z,n]
constexpr int kStages = 4;
constexpr int kMaskTiles = 17;           // ceil(257 / 16)
constexpr int kGruLayers = 2 * kStages;
// bf16 configuration: constants folded into the GRU weights and biases at pack time (r and z columns; n columns)
constexpr float kGateScaleRZ = -1.44269504088896341f;  // -log2(e)
constexpr float kGateScaleN = 2.88539008177792681f;    // 2 log2(e)
// bf16 configuration: b_hh rides in the recurrent GEMM's operands -- rows kBiasK0 and kBiasK0 + 1 of the packed W_hh hold the bias
// as a bf16 pair (hi = bf16(b), lo = bf16(b - hi): 16 significant bits) and the hidden-state operand carries the constant 1 at
// those two k (they are padding of the last k-block, 271 -> 288).  The accumulators start from the inline constant 0: no
// per-step bias fetch, no accumulator splats (round 4)
constexpr int kBiasK0 = kHidden;          // = 271: column 15 of unit tile 16; kBiasK0 + 1 = 272: the first column past tile 16
constexpr unsigned kBf16One = 0x3f80u;
// bf16 configuration with the folded front-end: a fed-forward head of at most kYPadMax values rides BEHIND the features in their
// last k-block -- columns kBins ... kBins + d - 1 of the 288 (k = 1 ... d of block 8: the first lane group of the block) -- instead of
// costing a k-block of its own in the next stage's input GEMM (stages 1 and 2: 1 and 5 values; round 4)
constexpr int kYPadMax = 7;
constexpr int kMaxFrontTaps = 5;         // KNS-v1.1: the front-end may see the last N <= 5 feature frames (the reference file has N = 5)

enum Precision { kFp32 = 0, kBf16 = 1 };

// geometry of one precision mode
struct PrecInfo {
    int kb;    // logical k per packed block
    int epl;   // operand elements per lane per block (16 B / sizeof(elem))
    int esz;   // operand element size in bytes
    int npb;   // n-tiles that make up one k-block when an output becomes the next A operand (kb / 16)
    int gisz;  // bytes per C-packed lane entry of the pre-activation buffer (4 values)
};

KNS_HD PrecInfo prec_info(int precision) {
    return precision == kBf16 ? PrecInfo{32, 8, 2, 2, 8} : PrecInfo{16, 4, 4, 1, 16};
}

KNS_HD int ceil_div(int a, int b) { return (a + b - 1) / b; }

// element offset (in operand elements) inside one 64-lane block of an A- or B-packed operand
KNS_HD int pack_off_bf16(int rc, int kk) { return (rc + 16 * (kk >> 3)) * 8 + (kk & 7); }
KNS_HD int pack_off_f32(int rc, int kk) { return (rc + 16 * (kk & 3)) * 4 + (kk >> 2); }
KNS_HD int pack_off(int precision, int rc, int kk) {
    return precision == kBf16 ? pack_off_bf16(rc, kk) : pack_off_f32(rc, kk);
}

// value offset inside one C-packed 16x16 tile (64 lanes x 4 values)
KNS_HD int cpack_off(int row, int col) { return ((row >> 2) * 16 + col) * 4 + (row & 3); }

}  // namespace kns
