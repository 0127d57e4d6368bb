// kns_kernels.h -- launch interface of the gfx950 kernels (kns_stft.hip, kns_gemm.hip, kns_gru.hip).  Host code only sees PODs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "kns_layout.h"

namespace kns {

// A kernel that asks for more dynamic LDS than the default limit needs the attribute once PER DEVICE (a process may hold
// engines on several GPUs): remembered per kernel instantiation and device.
template <class Kernel>
inline void allow_dynamic_lds(Kernel kernel, size_t bytes) {
    static unsigned long long done = 0;  // bit d: set on device d (launches of one engine come from one thread at a time)
    int dev = 0;
    (void) hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done & bit) return;
    (void) hipFuncSetAttribute((const void *) kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
    done |= bit;
}


// Developer switches (kernel A/B selection, tuning knobs, probing modes) exist only in the -DKNS_DEV build
// (lib/libpv_koala_dev.so, used by tests/ and tools/); the product library reads none of them.
#ifdef KNS_DEV
inline const char *dev_env(const char *name) { return getenv(name); }
#else
inline const char *dev_env(const char *) { return nullptr; }
#endif
// kernel-family overrides carried in the launch arguments (set by the engine from its developer switches; 0 in the product)
enum DevVariant { kDevGruStream = 1, kDevGemmGeneric = 2, kDevGemmNoWsr = 4 };

// ---- analysis: int16 frames -> spectrum + normalised log-power features (SURVEY 8a rows a2+a3)
struct AnalysisArgs {
    const int16_t *pcm;       // [B][T*256] row-major (caller layout)
    const int16_t *hist_in;   // [Bpad][256] last frame of the previous call
    int16_t *hist_out;        // [Bpad][256] last frame of this call
    const float *window;      // [512]
    const float *twiddle;     // [512][2] exp(-2 pi i k / 512)
    const float *mean;        // [257]
    const float *scale;       // [257]
    float *spec;              // [T][Bpad][256][2] packed half spectrum (bin 0 = {X0.re, X256.re}); bin k = c + 16 k2 sits at
                              // complex slot ((k2 >> 1) * 16 + c) * 2 + (k2 & 1): the STFT kernels' lane order
    void *feat;               // A-packed [T*mtiles][nbf] blocks
    int B, Bpad, T, nbf, precision;
    int seg;         // frames per workgroup (a workgroup walks a time segment of its 16 streams)
    int write_spec;  // 0: the synthesis kernel rebuilds the spectrum from the PCM, nothing is stored
    // optional (one-frame calls, front-end over several frames): `feat` is the LAST slot of a feature history of hist_slots + 1
    // frames [slot][mtiles][nbf]; before it is written the workgroup rolls its m-tile's history by one frame (slots 1 .. hist_slots
    // -> 0 .. hist_slots - 1), so that afterwards the slots are the front-end's taps, oldest first -- no copy launches
    void *feat_hist = nullptr;
    int hist_slots = 0;
    // a SLICE of frames of a longer call (the layer pipeline of mid-size batches, kns_engine.cpp): `pcm` points at the slice's first frame,
    // rows are `pitch` frames apart (0: T), the frame in front of the slice is read from the row itself instead of `hist_in`, and only the
    // call's last slice leaves the history behind
    int pitch = 0, prev_in_pcm = 0, write_hist = 1;
};
void launch_analysis(const AnalysisArgs &a, hipStream_t s);

// ---- synthesis: mask x spectrum -> iFFT -> window -> overlap-add -> int16
struct SynthesisArgs {
    const float *spec;      // as above
    const float *mask;      // C-packed [T*mtiles][17][64][4]: fp32, or fp16 (mask_fp16: the bf16 configuration's mask storage type)
    const float *window;    // [512]
    const float *twiddle;   // [512][2]
    const float *tail_in;   // [Bpad][256] overlap-add state left by the previous call
    float *tail_out;        // [Bpad][256] state after this call (ping-pong with tail_in)
    int16_t *out;           // [B][T*256]
    int B, Bpad, T;
    int seg;                // frames per workgroup; segments after the first replay one frame to rebuild the tail
    const int16_t *pcm;     // recompute != 0: the call's input [B][T*256] ...
    const int16_t *hist_in; // ... and the history the analysis kernel started from, [Bpad][256]
    int recompute;          // rebuild each frame's spectrum from its PCM instead of reading `spec`
    int mask_fp16 = 0;
    // optional (one-frame calls, bf16, stored spectrum): the mask head sigmoid(h . W_mask + b_mask) inside this launch -- four
    // further waves compute the workgroup's mask tile into LDS while the STFT waves fetch their operands; `mask` is not read then
    // a slice of frames of a longer call: `pcm` / `out` point at the slice's first frame, rows `pitch` frames apart (0: T); recompute: the
    // frame in front of the slice comes from the `pcm` row itself instead of `hist_in`
    int pitch = 0, prev_in_pcm = 0;
    const void *mask_h = nullptr;    // A-packed hidden sequence of the last stage's layer B [mtiles][9]
    const void *mask_w = nullptr;    // B-packed [17][9]
    const float *mask_b = nullptr;   // [17 * 16]
};
void launch_synthesis(const SynthesisArgs &a, hipStream_t s);

// ---- GEMM over all stream-frames: out = act(A . W + bias), A from up to two A-packed sources
enum GemmOut {
    kOutGi = 0,        // C-packed pre-activations (fp32 or fp16), no activation
    kOutMask = 1,      // C-packed, sigmoid: fp32 (fp32 configuration) or fp16 (bf16 configuration: half the bytes of the mask hand-off)
    kOutAPlain = 2,    // A-packed operand type, no activation
    kOutASigmoid = 3,  // A-packed operand type, sigmoid
};
struct GemmArgs {
    const void *a0;   // [mtiles][nb0] blocks (may be null when nb0 == 0)
    const void *a1;   // [mtiles][nb1] blocks
    const void *w;    // B-packed [ntiles][nb0 + nb1] blocks
    const float *bias;  // [ntiles * 16]
    void *out;
    int nb0, nb1;
    int mtiles, ntiles;
    int n_valid;  // logical output width; columns >= n_valid are written as 0 in A-packed outputs
    int out_kind, precision;
    int dev = 0;  // DevVariant bits
    // front-end over several stacked feature frames (KNS-v1.1): the a1 part of A is `taps` blocks of nb1 k-blocks, block i
    // read from a1 + i * tap_stride bytes (the same feature buffer, one frame further on): K = nb0 + taps * nb1 k-blocks
    int taps = 1;
    size_t tap_stride = 0;
    // kOutASigmoid only, optional: instead of whole A-packed blocks of their own, the n_valid (<= kYPadMax) output columns are
    // written INTO block pad_blk of an existing A-packed matrix with pad_nb blocks per m-tile, at columns pad_kk0 ... of that block
    // (2-byte elements; everything else of the block is left alone) -- a narrow head's values in the padding of the feature matrix
    int pad_nb = 0, pad_blk = 0, pad_kk0 = 0;
    // workgroups of the weight-stationary kernels (0: one per CU, 256).  The layer pipeline of mid-size batches launches them narrower,
    // beside the recurrent launches of other layers that hold most of the chip: a grid that does not fit waits for whole CUs
    int grid = 0;
};
void launch_gemm(const GemmArgs &a, hipStream_t s);

// ---- recurrent half of one GRU layer over T frames
struct GruArgs {
    const void *gi;     // C-packed [T][mtiles][51] tiles of (x . W_ih + b_ih)
    const void *whh;    // B-packed [51][nbh] blocks
    const float *bhh;   // [51 * 16]
    const float *hstate_in;  // C-packed fp32 [mtiles][17][64][4]: hidden state before this call
    float *hstate_out;       // the same after this call (ping-pong with hstate_in: the small-batch kernel reads whole
                             // state tiles in every workgroup, so it may not be updated in place)
    void *hseq;         // A-packed [T][mtiles][nbh] blocks, out
    int T, mtiles, precision;
    int dev = 0;  // DevVariant bits
    // optional (bf16 resident kernel): the stage's NARROW head (at most kYPadMax values) computed inside this launch, step by step,
    // from the hidden vectors while they are in LDS -- y_t = sigmoid(h_t . W_head + b_head) written into columns y_kk0 ... of block
    // y_blk of an A-packed matrix with y_nb blocks per m-tile and T x mtiles m-tiles (the feature matrix' padding, kns_layout.h): no
    // head launch, no second pass over the hidden sequence
    const void *yw = nullptr;    // B-packed head weights, n-tile 0: [nbh] blocks
    const float *yb = nullptr;   // [16]
    void *yout = nullptr;
    int yvalid = 0, y_nb = 0, y_blk = 0, y_kk0 = 0;
};
void launch_gru(const GruArgs &a, hipStream_t s);

// ---- a whole GRU layer (input GEMM + recurrent step + gates) for ONE frame of a few m-tiles: the low-latency path.
// One wavefront per (unit tile, m-tile): 17 x mtiles workgroups each stream only their 3 n-tiles of W_ih and W_hh.
struct GruSmallArgs {
    const void *a0;     // A-packed y_prev [mtiles][nb0] (null when nb0 == 0)
    const void *a1;     // A-packed e or previous layer's h [mtiles][nbh]
    const void *wih;    // B-packed [51][nb0 + nbh]
    const float *bih;   // [51 * 16]
    const void *whh;    // B-packed [51][nbh]
    const float *bhh;   // [51 * 16]
    const float *hstate_in;
    float *hstate_out;
    void *hseq;         // A-packed [mtiles][nbh], out (this frame's h as the next kernel's A operand)
    int nb0, mtiles, precision;
    // optional: the narrow head of the PREVIOUS stage computed inside this launch (y_prev = sigmoid(h_B . W_head + b_head) of the
    // workgroup's m-tile, every workgroup for itself); a0 is not read then
    const void *yh = nullptr;    // A-packed hidden sequence of the previous stage's layer B [mtiles][nbh]
    const void *yw = nullptr;    // B-packed head weights [n-tiles][nbh]
    const float *yb = nullptr;
    int yvalid = 0;              // head width (columns >= yvalid are zero)
};
void launch_gru_small(const GruSmallArgs &a, hipStream_t s);

// the wavefront form of multi-frame calls (kns_gru.hip, gru_wave_kernel): the items of one anti-diagonal of (pipeline stage, frame)
struct GruWaveItem {
    GruSmallArgs g;     // a GRU layer of one frame: as for launch_gru_small (a0 = the y operand of that frame, read from memory)
    const void *hprev = nullptr;  // A-packed h_{t-1} of this layer [m-tiles][NBH]: the hidden sequence's slot of frame t - 1, or the converted state
    // a narrow head of one frame instead (chains > 0): g.yh / yw / yb / yvalid, g.mtiles; its n-tiles = chains
    int chains = 0;
    int pad = 0;        // 1: the values go into columns y_kk0 ... of block y_blk of `yout` = the features [m-tiles][y_nb] (bf16)
    void *yout = nullptr;  // else: A-packed y operand [m-tiles][y_nb]
    int y_nb = 0, y_blk = 0, y_kk0 = 0;
};
constexpr int kWaveItems = 11;  // eight layers + three heads
struct GruWaveArgs {
    GruWaveItem item[kWaveItems];
    int layer_item[8];  // per XCD: the layer item its workgroups run (-1: none in this launch)
    int head_item[8];   // per XCD: a head item behind the layer's workgroups (-1: none)
    int layer_part[8];  // per XCD: which of the `parts` shares of its layer's workgroups it runs
    int parts;          // XCDs per layer in this launch: 1, 2, 4 or 8
    int layer_wgs;      // workgroups of a layer item = 17 x ceil(mtiles / mgroup)
    int xcd_wgs;        // layer workgroups per XCD = ceil(layer_wgs / parts)
    int mgroup;         // m-tiles per workgroup of a layer item
    int stamp = 0;      // -DKNS_TIMING builds: this launch writes the s_memtime stamps
};
void launch_gru_wave(const GruWaveArgs &w, int precision, int mtiles, hipStream_t s);
// the recurrent state [layers x mtiles][17 tiles] as A-packed operand blocks [layers x mtiles][NBH] (a wavefront call's frame 0)
void launch_gru_wave_prev(const float *hstate, void *hprev, int layers_x_mtiles, int precision, hipStream_t s);

// ---- a whole GRU layer of ONE frame in one launch: input GEMM + recurrent GEMM + gates fused over CU quads (kns_gruq.hip).
// Four workgroups on one XCD own four m-tiles (64 streams); workgroup c pulls the columns of W_ih AND W_hh that belong to
// hidden units 64 c .. 64 c + 63 (and serves unit tile 16 for m-tile c) and computes its quarter of h' for all four m-tiles.
struct GruQuadArgs {
    const void *a0;     // A-packed y_prev [mtiles][nb0] (null when nb0 == 0)
    const void *a1;     // A-packed e or the previous layer's hidden sequence [mtiles][9]
    const void *wih;    // B-packed [51][nb0 + 9]
    const float *bih;   // [51 * 16]
    const void *whh;    // B-packed [51][9] (with the bias rows, kns_layout.h kBiasK0)
    const float *bhh;   // [51 * 16] (unused: the bias rides in whh)
    const float *hstate_in;   // C-packed fp32 [mtiles][17][64][4]
    float *hstate_out;
    void *hseq;         // A-packed [mtiles][9], out
    int nb0, mtiles;
    int quad0 = 0;      // first quad of this launch (set by launch_gru_quad: 64 quads per launch)
    unsigned long long *dbg = nullptr;  // developer build: [8 waves][4][8] s_memtime stamps of workgroup `dbg_block`
    int dbg_block = 0;
    // optional: the narrow head of the PREVIOUS stage computed inside this launch instead of in a launch of its own -- y_prev =
    // sigmoid(h_B . W_head + b_head) of the quad's four m-tiles, straight into the staged x operand; a0 is not read then
    const void *yh = nullptr;    // A-packed hidden sequence of the previous stage's layer B [mtiles][9]
    const void *yw = nullptr;    // B-packed head weights [2 nb0][9]
    const float *yb = nullptr;   // [2 nb0 * 16]
    int yvalid = 0;              // head width (columns >= yvalid are zero)
};
// true when the shape is one the fused kernel takes (bf16, m-tiles in whole quads)
bool gru_quad_supported(int precision, int mtiles, int nb0);
void launch_gru_quad(const GruQuadArgs &a, hipStream_t s);

// ---- state reset of selected streams
struct ResetArgs {
    int16_t *hist;   // [Bpad][256] (both ping-pong copies are cleared)
    int16_t *hist2;
    float *tail;     // [Bpad][256] (both ping-pong copies are cleared)
    float *tail2;
    float *hstate;   // [8][mtiles][17][64][4] (both ping-pong copies are cleared)
    float *hstate2;
    const uint8_t *mask;  // [Bpad] device copy, or null for all
    int Bpad;
    // front-end context (KNS-v1.1, front_taps > 1): the features of the last front_taps - 1 frames, A-packed
    // [frames][mtiles][nbf] blocks; a reset stream's rows are set to the feature of a silent frame (`silent`: one A-packed
    // m-tile [nbf] blocks whose 16 rows are all that feature)
    void *fhist = nullptr;
    const void *silent = nullptr;
    int fhist_frames = 0, nbf = 0;
};
void launch_reset(const ResetArgs &a, hipStream_t s);
// last node of a captured one-frame replay: ++*counter (device memory), published to *host_word (page-locked host memory)
void launch_frame_done(unsigned *counter, unsigned *host_word, hipStream_t s);

}  // namespace kns
