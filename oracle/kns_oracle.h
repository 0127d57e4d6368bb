/*
 * kns_oracle.h -- CPU ORACLE for the KNS-v1 streaming noise-suppression spec.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under koala_amd/ (the product) may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg use it, as the checker.
 *
 * PARITY STATUS: "parity unpinned" against the reference's pv_koala_process samples.  The reference
 * engine (include/pv_koala.h:65-80 -> lib/linux/x86_64/libpv_koala.so) is a closed, licence-gated
 * binary: no source to restate, no golden PCM in its tests, and pv_koala_init cannot succeed here
 * (SURVEY.md section 8c).  What IS pinned against the reference: the ABI constants and error fixtures
 * (tests/golden/abi_fixtures.json, captured from the shipped .so) and the reference's own acceptance
 * envelope (binding/python/test_koala.py:71-129) on resources/audio_samples/{test,noise}.wav.
 *
 * This file restates the frozen spec "KNS-v1" (DESIGN.md section 2), which follows the observable contract:
 *   - 16 kHz mono int16, 256-sample frames            include/pv_koala.h:26-33, pv_koala_frame_length()
 *   - fixed delay between input and output streams    include/pv_koala.h:92-100 (KNS-v1: 256 samples)
 *   - reset == freshly created                        include/pv_koala.h:82-90
 *   - 257 spectral bins / 2 normalisation tables / 4 cascaded 2-layer GRU(271) stages with heads
 *     of width 1,5,40,257                             lib/common/koala_params.pv byte layout
 */
#ifndef KNS_ORACLE_H
#define KNS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KNS_FRAME 256
#define KNS_NFFT 512
#define KNS_BINS 257
#define KNS_H 271
#define KNS_STAGES 4
#define KNS_G3 (3 * KNS_H)
/* Oracle-only extension: the front-end may see the last N feature frames (KNS1 header word 11; w_in then has N * 257 rows,
 * oldest frame first).  The reference's model file has such a front-end (N = 5, koala_params.pv record [1285, 271]); KNS-v1
 * and the GPU engine have N = 1.  Used by tools/pv_hypotheses.py to find out whether the missing context is what keeps
 * imported weights from behaving. */
#define KNS_MAX_FRONT_TAPS 5
#define KNS_MAX_BLOCK 64 /* streams that share one pass over the weights in kns_oracle_process */

/* precision modes: which rounding points of the GPU pipeline are emulated */
#define KNS_PREC_FP32 0 /* no rounding anywhere; GEMMs are k-ordered fmaf chains                    */
#define KNS_PREC_BF16 1 /* GEMM operands (weights+activations) rounded to bf16, input-side pre-activations to fp16 */

typedef struct kns_params kns_params_t;
typedef struct kns_oracle kns_oracle_t;

/* optional taps of one frame's intermediates (any pointer may be NULL) */
typedef struct {
    float *spectrum; /* [257][2] re,im of the analysis STFT                    */
    float *features; /* [257]                                                   */
    float *embed;    /* [271]  front-end output e                               */
    float *heads;    /* [1+5+40+257] y_1..y_4 (y_4 is the mask)                  */
    float *hidden;   /* [8][271] hidden states after this frame (A1,B1,A2,...)   */
} kns_taps_t;

/* returns 0 on success, negative on failure (-1 io, -2 format) */
int kns_params_load(const char *path, int precision, kns_params_t **out);
void kns_params_free(kns_params_t *p);
int kns_params_head_dim(const kns_params_t *p, int stage);

/* an oracle instance = `num_streams` independent streams sharing one parameter set */
int kns_oracle_create(const kns_params_t *p, int num_streams, kns_oracle_t **out);
void kns_oracle_delete(kns_oracle_t *o);
void kns_oracle_reset(kns_oracle_t *o, const uint8_t *stream_mask /* NULL = all */);
int kns_oracle_delay_sample(void);

/* pcm, enhanced: [num_streams][num_frames*256] row-major int16.  num_threads<=0 -> all cores. */
int kns_oracle_process(kns_oracle_t *o, int num_frames, const int16_t *pcm, int16_t *enhanced, int num_threads);

/* the same, also returning every frame's mask y_4 as [num_frames][num_streams][257] (NULL: no masks) */
int kns_oracle_process_mask(kns_oracle_t *o, int num_frames, const int16_t *pcm, int16_t *enhanced, float *mask,
                            int num_threads);

/* streams per block used by the last kns_oracle_process call (reporting only) */
int kns_oracle_last_block(void);

/* Sensitivity probe (bf16 mode only; 0 = off, also settable as KNS_ORACLE_JITTER in the environment): the oracle plays a SECOND
 * valid implementation of the tolerance-specified bf16 configuration -- the last bit of every transcendental result and of some
 * GEMM outputs moves by one ulp, seeded by (seed, stream, frame).  The PCM distance between a plain and a jittered run of one model
 * predicts that model's GPU-vs-oracle distance (tools/model_sensitivity.py, tests/test_holdout.py).  Never set in a parity test. */
void kns_oracle_set_jitter(int seed);

/* single stream (index s), one frame, with taps */
int kns_oracle_process_tap(kns_oracle_t *o, int s, const int16_t *pcm, int16_t *enhanced, kns_taps_t *taps);

/* stage-level entry points for unit parity tests of individual GPU kernels */
void kns_oracle_analysis(const kns_params_t *p, const int16_t *hist256, const int16_t *pcm256,
                         float *spectrum /*[257][2]*/, float *features /*[257]*/);
void kns_oracle_synthesis(const float *spectrum /*[257][2]*/, const float *mask /*[257]*/, float *tail256 /*in/out*/,
                          int16_t *out256);
/* the same two stages through a textbook radix-2 FFT-512 over the full complex block: NOT the spec (whose operation order
 * is the packed 16 x 16 transform of kns_oracle.c), an independent cross-check to a tolerance */
void kns_oracle_analysis_radix2(const kns_params_t *p, const int16_t *hist256, const int16_t *pcm256, float *spectrum,
                                float *features);
void kns_oracle_synthesis_radix2(const float *spectrum, const float *mask, float *tail256, int16_t *out256);
/* scalar math of the spec (exposed so tests can compare the GPU's device functions bit for bit) */
float kns_exp(float x);
float kns_log(float x);
float kns_log_fast(float x); /* the bf16 mode's feature logarithm */
float kns_sigmoid(float x);
float kns_tanh(float x);
float kns_round_bf16(float x);
float kns_round_fp16(float x);

#ifdef __cplusplus
}
#endif
#endif
