/*
 * picovoice.h -- status codes, sample rate and the thread-local error stack of the drop-in libpv_koala.so
 * built from koala_amd/csrc (MI355X-native engine).
 *
 * Replaces, symbol for symbol, the reference's include/picovoice.h:
 *   pv_sample_rate          reference include/picovoice.h:36   (returns 16000)
 *   pv_status_t             reference include/picovoice.h:41-54 (12 codes, same numeric values)
 *   pv_status_to_string     reference include/picovoice.h:62
 *   pv_get_error_stack      reference include/picovoice.h:77-79
 *   pv_free_error_stack     reference include/picovoice.h:86
 * plus the exports the reference library carries without declaring them (called by
 * binding/python/_koala.py:156-160 and binding/web/src/koala.ts:71,484-487):
 *   pv_set_sdk  pv_get_sdk  pv_free  pv_log_enable  pv_log_disable
 */
#ifndef PICOVOICE_H
#define PICOVOICE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PV_API __attribute__((visibility("default")))

typedef enum {
    PV_STATUS_SUCCESS = 0,
    PV_STATUS_OUT_OF_MEMORY = 1,
    PV_STATUS_IO_ERROR = 2,
    PV_STATUS_INVALID_ARGUMENT = 3,
    PV_STATUS_STOP_ITERATION = 4,
    PV_STATUS_KEY_ERROR = 5,
    PV_STATUS_INVALID_STATE = 6,
    PV_STATUS_RUNTIME_ERROR = 7,
    PV_STATUS_ACTIVATION_ERROR = 8,
    PV_STATUS_ACTIVATION_LIMIT_REACHED = 9,
    PV_STATUS_ACTIVATION_THROTTLED = 10,
    PV_STATUS_ACTIVATION_REFUSED = 11
} pv_status_t;

/* 16000: the only sample rate the engine accepts. */
PV_API int32_t pv_sample_rate(void);

/* "SUCCESS", "OUT_OF_MEMORY", ... for the 12 codes above; NULL for anything else. */
PV_API const char *pv_status_to_string(pv_status_t status);

/*
 * After any call returned a status other than PV_STATUS_SUCCESS the calling THREAD may fetch, once, the
 * messages that failure left behind (at most 8, each "<build id> <8-hex location>: <text>").  The array is
 * owned by the caller and released with pv_free_error_stack.  With nothing pending the call returns
 * PV_STATUS_INVALID_STATE and a depth of 0.
 */
PV_API pv_status_t pv_get_error_stack(char ***message_stack, int32_t *message_stack_depth);
PV_API void pv_free_error_stack(char **message_stack);

/* Undeclared in the reference header but exported by its library and used by its bindings. */
PV_API void pv_set_sdk(const char *sdk);
PV_API const char *pv_get_sdk(void);
PV_API void pv_free(void *ptr);
PV_API void pv_log_enable(void);
PV_API void pv_log_disable(void);

#ifdef __cplusplus
}
#endif
#endif /* PICOVOICE_H */
