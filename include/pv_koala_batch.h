/*
 * pv_koala_batch.h -- additive batch extension of the Koala C ABI (SURVEY.md 8b "Batch extension").
 *
 * The reference ABI is one stream per handle, one frame per call (include/pv_koala.h:65-80); BASELINE.json's
 * configs need thousands of independent streams per GPU.  A batch handle is B streams advancing in lock step;
 * stream b of a batch behaves exactly like its own pv_koala_t (same samples, same delay, same reset semantics).
 * pv_koala_init/process are the B = 1 instance of the same engine.
 */
#ifndef PV_KOALA_BATCH_H
#define PV_KOALA_BATCH_H

#include <stdint.h>

#include "picovoice.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pv_koala_batch pv_koala_batch_t;

typedef enum {
    PV_KOALA_PRECISION_FP32 = 0, /* fp32 operands on the f32 MFMA path: +-1 LSB against the fp32 oracle       */
    PV_KOALA_PRECISION_BF16 = 1  /* bf16 GEMM operands, fp32 accumulate and gates, fp32 FFT (BASELINE configs[2]) */
} pv_koala_precision_t;

/* `device` accepts the grammar of pv_koala_init.  `max_frames_per_call` bounds `num_frames` of process_chunk and
 * sizes the activation workspace in HBM. */
PV_API pv_status_t pv_koala_batch_init(const char *access_key, const char *model_path, const char *device,
                                       int32_t num_streams, int32_t max_frames_per_call,
                                       pv_koala_precision_t precision, pv_koala_batch_t **object);
PV_API void pv_koala_batch_delete(pv_koala_batch_t *object);

/* One frame per stream: pcm and enhanced are [num_streams][256] row-major int16.  Pointers may be host memory
 * (staged through pinned buffers, call returns when `enhanced` is filled) or device memory on the handle's GPU
 * (kernels are enqueued on the handle's stream and the call returns without synchronising). */
PV_API pv_status_t pv_koala_batch_process(pv_koala_batch_t *object, const int16_t *pcm, int16_t *enhanced);

/* `num_frames` consecutive frames per stream: [num_streams][num_frames*256].  DEVICE pointers: `enhanced` may be the same
 * buffer as `pcm` (in-place) or overlap it partially (detected; the call then takes the stored-spectrum path).  HOST
 * pointers: `enhanced` may equal `pcm` EXACTLY in every call (in-place).  A PARTIAL overlap is fine while the call is not
 * pipelined in sub-chunks (below 4 MiB, or num_frames <= min(16, max_frames_per_call / 2)); larger host calls copy chunks out
 * while later chunks of `pcm` are still unread, so partially overlapping host buffers are rejected with PV_STATUS_RUNTIME_ERROR
 * (nothing is processed, the streams' state is unchanged). */
PV_API pv_status_t pv_koala_batch_process_chunk(pv_koala_batch_t *object, int32_t num_frames, const int16_t *pcm,
                                                int16_t *enhanced);

/* The same for a throughput-oriented host caller (a many-files batch job that double-buffers its I/O): `pcm` and `enhanced` must be
 * PAGE-LOCKED host memory (pv_koala_batch_host_alloc, hipHostMalloc, hipHostRegister; anything else is refused with
 * PV_STATUS_RUNTIME_ERROR, nothing processed).  The call enqueues its copy-in, kernels and copy-out and RETURNS; up to three such calls are
 * in flight per handle (a fourth first waits for the oldest), so one call's copies run under its neighbours' kernels -- a synchronous
 * call cannot hide its first copy-in and last copy-out.  Calls complete in order; `enhanced` of a call is valid, and `pcm` may be
 * reused, once pv_koala_batch_synchronize() or pv_koala_batch_async_wait() says the call has completed (a caller rotating over three
 * buffer pairs keeps both directions of the link and the GPU busy at once).  Every other entry
 * point of the handle first waits for the calls in flight.  `enhanced` may equal `pcm`. */
PV_API pv_status_t pv_koala_batch_process_chunk_async(pv_koala_batch_t *object, int32_t num_frames, const int16_t *pcm,
                                                      int16_t *enhanced);

/* Blocks until at most `max_in_flight` asynchronous calls of the handle are still in flight (0: all have completed; calls complete in
 * order).  The triple-buffering loop of a host caller:  for call n:  pv_koala_batch_async_wait(o, 2)  -- call n - 3 is complete: take its
 * `enhanced`, refill its `pcm` --  then pv_koala_batch_process_chunk_async(o, frames, pcm[n % 3], enhanced[n % 3]). */
PV_API pv_status_t pv_koala_batch_async_wait(pv_koala_batch_t *object, int32_t max_in_flight);

/* Resets the streams whose byte in `stream_mask[num_streams]` (host memory) is non-zero; NULL resets all. */
PV_API pv_status_t pv_koala_batch_reset(pv_koala_batch_t *object, const uint8_t *stream_mask);

PV_API pv_status_t pv_koala_batch_num_streams(const pv_koala_batch_t *object, int32_t *num_streams);
PV_API pv_status_t pv_koala_batch_delay_sample(const pv_koala_batch_t *object, int32_t *delay_sample);

/* Page-locked host memory for `pcm` / `enhanced`.  Host-pointer calls are pipelined in sub-chunks (copy-in, kernels and
 * copy-out of consecutive sub-chunks overlap); buffers obtained here -- or any other page-locked memory -- are read and
 * written by the GPU's copy engines directly, ordinary (pageable) buffers go through the handle's staging slots first.
 * Not tied to a handle; release with pv_koala_batch_host_free (NULL is accepted). */
PV_API pv_status_t pv_koala_batch_host_alloc(int64_t num_bytes, void **memory);
PV_API void pv_koala_batch_host_free(void *memory);

/* Run on a caller-provided HIP stream (a hipStream_t passed as void*; NULL = the handle's own stream).  Changing the stream first
 * waits for everything the handle has in flight (asynchronous host calls, work queued on the previous stream -- which must still
 * exist), since the next call's kernels work on the same stream state. */
PV_API pv_status_t pv_koala_batch_set_stream(pv_koala_batch_t *object, void *hip_stream);
/* Blocks until everything enqueued by this handle -- device-pointer calls, asynchronous host calls -- has finished. */
PV_API pv_status_t pv_koala_batch_synchronize(pv_koala_batch_t *object);

/* Per-kernel timing with HIP events recorded on the handle's stream (bench.py's roofline leg).
 * kernel classes: 0 analysis, 1 input-side GEMMs, 2 recurrent GRU, 3 head/front-end GEMMs, 4 synthesis. */
#define PV_KOALA_NUM_KERNEL_CLASSES 5
PV_API pv_status_t pv_koala_batch_profile_enable(pv_koala_batch_t *object, int32_t enable);
PV_API pv_status_t pv_koala_batch_profile_read(pv_koala_batch_t *object, double *milliseconds /*[5]*/,
                                               int64_t *launches /*[5]*/);

/* Debug taps of the last processed chunk, copied to host in logical (unpacked) layout; used by the parity tests.
 * what: 0 features [T][B][257], 1 spectrum [T][B][257][2], 2 mask [T][B][257], 3 hidden state [8][B][271],
 *       4 embedding e [T][B][271].  Returns the number of floats written, or a negative pv_status_t. */
PV_API int64_t pv_koala_batch_debug_read(pv_koala_batch_t *object, int32_t what, float *out, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* PV_KOALA_BATCH_H */
