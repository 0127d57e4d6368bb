/*
 * pv_koala.h -- single-stream C ABI of the MI355X-native noise suppressor (drop-in for Picovoice Koala 3.0.0).
 *
 * Each entry point replaces the same-named one of the reference header; numbers are reference lines:
 *   pv_koala_init                   include/pv_koala.h:52-56
 *   pv_koala_delete                 include/pv_koala.h:63
 *   pv_koala_process                include/pv_koala.h:80      <- the hot path
 *   pv_koala_reset                  include/pv_koala.h:90
 *   pv_koala_delay_sample           include/pv_koala.h:100
 *   pv_koala_frame_length           include/pv_koala.h:107     (256)
 *   pv_koala_version                include/pv_koala.h:114
 *   pv_koala_list_hardware_devices  include/pv_koala.h:126-128
 *   pv_koala_free_hardware_devices  include/pv_koala.h:136-138
 *
 * Contract kept from the reference: mono 16 kHz int16 audio, pv_koala_frame_length() samples per call, output
 * lags input by pv_koala_delay_sample() samples, reset == new instance, one caller per handle at a time.
 * Differences, all documented in DESIGN.md section 3: `access_key` is only checked for NULL/empty (no licensing),
 * `model_path` names a KNS1 parameter file, compute devices are AMD GPUs only (`best`, `gpu`, `gpu:N`; a
 * well-formed `cpu[:N]` is rejected with PV_STATUS_RUNTIME_ERROR because this build has no CPU backend).
 */
#ifndef PV_KOALA_H
#define PV_KOALA_H

#include <stdint.h>

#include "picovoice.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pv_koala pv_koala_t;

/* Creates one stream on `device`.  Fails with INVALID_ARGUMENT (NULL/empty argument, malformed device string),
 * IO_ERROR (model file), RUNTIME_ERROR (device unavailable) or OUT_OF_MEMORY; *object is untouched on failure. */
PV_API pv_status_t pv_koala_init(const char *access_key, const char *model_path, const char *device,
                                 pv_koala_t **object);

/* Releases the stream; NULL is ignored. */
PV_API void pv_koala_delete(pv_koala_t *object);

/* Consumes 256 input samples, produces the 256 enhanced samples that lie delay_sample behind them.
 * `pcm` and `enhanced_pcm` are caller-owned host buffers, may be reused, are not retained. */
PV_API pv_status_t pv_koala_process(pv_koala_t *object, const int16_t *pcm, int16_t *enhanced_pcm);

/* Forgets all history (analysis window, recurrent state, overlap-add tail). */
PV_API pv_status_t pv_koala_reset(pv_koala_t *object);

PV_API pv_status_t pv_koala_delay_sample(const pv_koala_t *object, int32_t *delay_sample);
PV_API int32_t pv_koala_frame_length(void);
PV_API const char *pv_koala_version(void);

/* One "gpu:<i> - <name>" string per visible AMD GPU; free with pv_koala_free_hardware_devices. */
PV_API pv_status_t pv_koala_list_hardware_devices(char ***hardware_devices, int32_t *num_hardware_devices);
PV_API void pv_koala_free_hardware_devices(char **hardware_devices, int32_t num_hardware_devices);

#ifdef __cplusplus
}
#endif
#endif /* PV_KOALA_H */
