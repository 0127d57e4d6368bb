// pv_api.cpp -- the C ABI of libpv_koala.so: single-stream entry points of include/pv_koala.h + include/picovoice.h
// and the batch extension of include/pv_koala_batch.h, all driving kns::Engine (HIP, gfx950).
//
// Behaviour at the boundary follows what the reference library was observed to do:
// argument checks in the same order (NULL arguments -> model file -> device string -> device availability),
// the same status codes and message texts, a thread-local error stack of at most 8 messages drained by one
// pv_get_error_stack call, pv_koala_delete(NULL) a no-op.  There is no CPU compute path in this library: a call
// that needs the GPU and cannot reach one fails with a status and a message, it never falls back.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pv_koala.h"
#include "../../include/pv_koala_batch.h"
#include "kns_engine.h"

#ifdef KNS_TIMING
namespace kns {
void read_timing(unsigned long long *out);
}
#endif

namespace {

const char kBuildId[] = "a355c0a";  // 7 hex digits, as the reference prints in front of every message

thread_local std::vector<std::string> t_stack;

void push_error(unsigned code, const char *fmt, ...) {
    char text[768];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(text, sizeof(text), fmt, ap);
    va_end(ap);
    char line[832];
    snprintf(line, sizeof(line), "%s %08X: %s", kBuildId, code, text);
    if (t_stack.size() < 8) t_stack.push_back(line);
}

std::mutex g_sdk_mutex;
std::string g_sdk = "c";
bool g_log = false;

const char *const kStatusNames[] = {"SUCCESS",          "OUT_OF_MEMORY",           "IO_ERROR",
                                    "INVALID_ARGUMENT", "STOP_ITERATION",          "KEY_ERROR",
                                    "INVALID_STATE",    "RUNTIME_ERROR",           "ACTIVATION_ERROR",
                                    "ACTIVATION_LIMIT_REACHED", "ACTIVATION_THROTTLED", "ACTIVATION_REFUSED"};

enum DeviceKind { kDevBest, kDevCpu, kDevGpu };
struct DeviceSpec {
    DeviceKind kind;
    int index;  // gpu index or cpu thread count; -1 = unspecified
};

// grammar of the reference's `device` argument (include/pv_koala.h:42-46): best | cpu | cpu:N | gpu | gpu:N
bool parse_device(const char *text, DeviceSpec *out) {
    // an entry of pv_koala_list_hardware_devices ("gpu:0 - <name>") is accepted as it was printed
    std::string head(text);
    const size_t dash = head.find(" - ");
    if (dash != std::string::npos && dash > 0) head.resize(dash);
    const char *s = head.c_str();
    auto number = [](const char *p, int *v) {
        if (!*p) return false;
        long n = 0;
        for (; *p; ++p) {
            if (*p < '0' || *p > '9') return false;
            n = n * 10 + (*p - '0');
            if (n > 1 << 20) return false;
        }
        *v = (int) n;
        return true;
    };
    if (!strcmp(s, "best")) {
        *out = {kDevBest, -1};
        return true;
    }
    if (!strcmp(s, "cpu")) {
        *out = {kDevCpu, -1};
        return true;
    }
    if (!strcmp(s, "gpu")) {
        *out = {kDevGpu, -1};
        return true;
    }
    int v;
    if (!strncmp(s, "cpu:", 4) && number(s + 4, &v)) {
        *out = {kDevCpu, v};
        return true;
    }
    if (!strncmp(s, "gpu:", 4) && number(s + 4, &v)) {
        *out = {kDevGpu, v};
        return true;
    }
    return false;
}

// No C++ exception may cross the C ABI (the callers are ctypes / dlsym hosts: an escaping exception is std::terminate).  Every
// entry point that reaches engine code runs it through this: bad_alloc -> OUT_OF_MEMORY, anything else -> RUNTIME_ERROR.
template <class F>
pv_status_t guarded(F &&body) {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        push_error(0x65, "Failed to allocate memory.");
        return PV_STATUS_OUT_OF_MEMORY;
    } catch (const std::length_error &) {
        push_error(0x65, "Failed to allocate memory.");
        return PV_STATUS_OUT_OF_MEMORY;
    } catch (const std::exception &e) {
        push_error(0x339, "Unexpected failure: %s", e.what());
        return PV_STATUS_RUNTIME_ERROR;
    } catch (...) {
        push_error(0x339, "Unexpected failure.");
        return PV_STATUS_RUNTIME_ERROR;
    }
}

// shared front half of pv_koala_init / pv_koala_batch_init
pv_status_t open_engine_unguarded(const char *access_key, const char *model_path, const char *device, void *object,
                                  int num_streams, int max_frames, int precision, kns::Engine **engine) {
    if (!access_key) {
        push_error(0x64, "Argument `access_key` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    if (!model_path) {
        push_error(0x64, "Argument `model_path` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    if (!device) {  // the reference dereferences NULL here; an argument error is the drop-in-safe behaviour
        push_error(0x64, "Argument `device` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    if (!object) {
        push_error(0x64, "Argument `object` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    kns::Params params;
    std::string err;
    kns::LoadResult lr = kns::load_params(model_path, &params, &err);
    if (lr == kns::kLoadIo) {
        push_error(0xC9, "%s", err.c_str());
        push_error(0x136, "Koala model (.kns) could not be opened.");
        return PV_STATUS_IO_ERROR;
    }
    if (lr != kns::kLoadOk) {
        push_error(0xCA, "%s", err.c_str());
        push_error(0x136, "Koala model (.kns) could not be read.");
        return PV_STATUS_IO_ERROR;
    }
    DeviceSpec spec;
    if (!parse_device(device, &spec)) {
        push_error(0x322, "%s is not a valid device string", device);
        push_error(0x12C, "Picovoice Error.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    if (spec.kind == kDevCpu) {
        push_error(0x323, "Device `%s` is not available: this build has no CPU backend, use `best`, `gpu` or `gpu:N`.",
                   device);
        push_error(0x12C, "Picovoice Error.");
        return PV_STATUS_RUNTIME_ERROR;
    }
    const int ngpu = kns::visible_gpu_count();
    if (ngpu <= 0) {
        push_error(0x334, "Failed to communicate with device.");
        push_error(0x12C, "Picovoice Error.");
        return PV_STATUS_RUNTIME_ERROR;
    }
    const int index = spec.index < 0 ? 0 : spec.index;
    if (index >= ngpu) {
        push_error(0x335, "GPU device index `%d` is out of range. %d device(s) available.", index, ngpu);
        push_error(0x12C, "Picovoice Error.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    if (!*access_key) {
        push_error(0x190, "Failed to parse AccessKey ``.");
        push_error(0x12C, "Picovoice Error.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    bool oom = false;
    kns::Engine *e = kns::Engine::create(params, index, num_streams, max_frames, precision, &err, &oom);
    if (!e) {
        push_error(oom ? 0x65 : 0x336, "%s", err.c_str());
        push_error(0x12C, "Picovoice Error.");
        return oom ? PV_STATUS_OUT_OF_MEMORY : PV_STATUS_RUNTIME_ERROR;
    }
    *engine = e;
    return PV_STATUS_SUCCESS;
}

pv_status_t open_engine(const char *access_key, const char *model_path, const char *device, void *object, int num_streams,
                        int max_frames, int precision, kns::Engine **engine) {
    return guarded([&] { return open_engine_unguarded(access_key, model_path, device, object, num_streams, max_frames, precision, engine); });
}

int default_precision() {
    const char *p = getenv("KOALA_AMD_PRECISION");
    if (p && !strcmp(p, "bf16")) return kns::kBf16;
    return kns::kFp32;
}

}  // namespace

struct pv_koala {
    kns::Engine *engine;
};
struct pv_koala_batch {
    kns::Engine *engine;
};

extern "C" {

// ------------------------------------------------------------------------------------------------ picovoice.h

PV_API int32_t pv_sample_rate(void) { return 16000; }

PV_API const char *pv_status_to_string(pv_status_t status) {
    if ((int) status < 0 || (int) status > 11) return NULL;
    return kStatusNames[(int) status];
}

PV_API pv_status_t pv_get_error_stack(char ***message_stack, int32_t *message_stack_depth) {
    if (!message_stack || !message_stack_depth) return PV_STATUS_INVALID_ARGUMENT;
    const size_t n = t_stack.size();
    if (n == 0) {  // nothing pending: no allocation either -- a caller that sees a failure status frees nothing (found by the ASan build)
        *message_stack = NULL;
        *message_stack_depth = 0;
        return PV_STATUS_INVALID_STATE;
    }
    char **arr = (char **) calloc(n + 1, sizeof(char *));
    if (!arr) return PV_STATUS_OUT_OF_MEMORY;
    for (size_t i = 0; i < n; ++i) arr[i] = strdup(t_stack[i].c_str());
    t_stack.clear();
    *message_stack = arr;
    *message_stack_depth = (int32_t) n;
    return PV_STATUS_SUCCESS;
}

PV_API void pv_free_error_stack(char **message_stack) {
    if (!message_stack) return;
    for (char **p = message_stack; *p; ++p) free(*p);
    free(message_stack);
}

PV_API void pv_set_sdk(const char *sdk) {
    std::lock_guard<std::mutex> lock(g_sdk_mutex);
    if (sdk) g_sdk = sdk;
}

PV_API const char *pv_get_sdk(void) {
    std::lock_guard<std::mutex> lock(g_sdk_mutex);
    static thread_local std::string copy;
    copy = g_sdk;
    return copy.c_str();
}

PV_API void pv_free(void *ptr) { free(ptr); }
PV_API void pv_log_enable(void) { g_log = true; }
PV_API void pv_log_disable(void) { g_log = false; }

// ------------------------------------------------------------------------------------------------ pv_koala.h

PV_API pv_status_t pv_koala_init(const char *access_key, const char *model_path, const char *device,
                                 pv_koala_t **object) {
    t_stack.clear();
    kns::Engine *e = nullptr;
    pv_status_t st = open_engine(access_key, model_path, device, object, 1, 1, default_precision(), &e);
    if (st != PV_STATUS_SUCCESS) return st;
    pv_koala_t *o = (pv_koala_t *) malloc(sizeof(pv_koala_t));
    if (!o) {
        delete e;
        push_error(0x65, "Failed to allocate memory.");
        return PV_STATUS_OUT_OF_MEMORY;
    }
    o->engine = e;
    *object = o;
    return PV_STATUS_SUCCESS;
}

PV_API void pv_koala_delete(pv_koala_t *object) {
    if (!object) return;
    delete object->engine;
    free(object);
}

PV_API pv_status_t pv_koala_process(pv_koala_t *object, const int16_t *pcm, int16_t *enhanced_pcm) {
    t_stack.clear();
    if (!object) {
        push_error(0x64, "Argument `object` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    if (!pcm) {
        push_error(0x64, "Argument `pcm` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    if (!enhanced_pcm) {
        push_error(0x64, "Argument `enhanced_pcm` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    return guarded([&] {
        std::string err;
        if (!object->engine->process(1, pcm, enhanced_pcm, &err, /*host_pointers=*/true)) {
            push_error(0x337, "%s", err.c_str());
            push_error(0x12C, "Picovoice Error.");
            return PV_STATUS_RUNTIME_ERROR;
        }
        return PV_STATUS_SUCCESS;
    });
}

PV_API pv_status_t pv_koala_reset(pv_koala_t *object) {
    t_stack.clear();
    if (!object) return PV_STATUS_INVALID_ARGUMENT;  // the reference leaves no message for this one
    return guarded([&] {
        std::string err;
        if (!object->engine->reset(nullptr, &err)) {
            push_error(0x338, "%s", err.c_str());
            return PV_STATUS_RUNTIME_ERROR;
        }
        return PV_STATUS_SUCCESS;
    });
}

PV_API pv_status_t pv_koala_delay_sample(const pv_koala_t *object, int32_t *delay_sample) {
    t_stack.clear();
    if (!object) {
        push_error(0x64, "Argument `object` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    if (!delay_sample) {
        push_error(0x64, "Argument `delay_sample` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    *delay_sample = kns::kFrame;
    return PV_STATUS_SUCCESS;
}

PV_API int32_t pv_koala_frame_length(void) { return kns::kFrame; }

PV_API const char *pv_koala_version(void) { return "3.0.0"; }

PV_API pv_status_t pv_koala_list_hardware_devices(char ***hardware_devices, int32_t *num_hardware_devices) {
    if (!hardware_devices || !num_hardware_devices) return PV_STATUS_INVALID_ARGUMENT;
    const int n = kns::visible_gpu_count();
    char **arr = (char **) calloc((size_t) n + 1, sizeof(char *));
    if (!arr) return PV_STATUS_OUT_OF_MEMORY;
    for (int i = 0; i < n; ++i) {
        char line[320];
        snprintf(line, sizeof(line), "gpu:%d - %s", i, kns::gpu_name(i).c_str());
        arr[i] = strdup(line);
    }
    *hardware_devices = arr;
    *num_hardware_devices = n;
    return PV_STATUS_SUCCESS;
}

PV_API void pv_koala_free_hardware_devices(char **hardware_devices, int32_t num_hardware_devices) {
    if (!hardware_devices) return;
    for (int32_t i = 0; i < num_hardware_devices; ++i) free(hardware_devices[i]);
    free(hardware_devices);
}

// ------------------------------------------------------------------------------------------------ pv_koala_batch.h

PV_API pv_status_t pv_koala_batch_init(const char *access_key, const char *model_path, const char *device,
                                       int32_t num_streams, int32_t max_frames_per_call,
                                       pv_koala_precision_t precision, pv_koala_batch_t **object) {
    t_stack.clear();
    if (num_streams <= 0 || max_frames_per_call <= 0) {
        push_error(0x66, "`num_streams` and `max_frames_per_call` must be positive.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    if (precision != PV_KOALA_PRECISION_FP32 && precision != PV_KOALA_PRECISION_BF16) {
        push_error(0x66, "`precision` must be PV_KOALA_PRECISION_FP32 or PV_KOALA_PRECISION_BF16.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    kns::Engine *e = nullptr;
    pv_status_t st = open_engine(access_key, model_path, device, object, num_streams, max_frames_per_call,
                                 precision == PV_KOALA_PRECISION_BF16 ? kns::kBf16 : kns::kFp32, &e);
    if (st != PV_STATUS_SUCCESS) return st;
    pv_koala_batch_t *o = (pv_koala_batch_t *) malloc(sizeof(pv_koala_batch_t));
    if (!o) {
        delete e;
        push_error(0x65, "Failed to allocate memory.");
        return PV_STATUS_OUT_OF_MEMORY;
    }
    o->engine = e;
    *object = o;
    return PV_STATUS_SUCCESS;
}

PV_API void pv_koala_batch_delete(pv_koala_batch_t *object) {
    if (!object) return;
    delete object->engine;
    free(object);
}

PV_API pv_status_t pv_koala_batch_process_chunk(pv_koala_batch_t *object, int32_t num_frames, const int16_t *pcm,
                                                int16_t *enhanced) {
    t_stack.clear();
    if (!object) {
        push_error(0x64, "Argument `object` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    if (!pcm || !enhanced) {
        push_error(0x64, "Argument `%s` is NULL.", pcm ? "enhanced" : "pcm");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    if (num_frames <= 0 || num_frames > object->engine->max_frames()) {
        push_error(0x66, "`num_frames` %d is outside [1, %d].", num_frames, object->engine->max_frames());
        return PV_STATUS_INVALID_ARGUMENT;
    }
    return guarded([&] {
        std::string err;
        if (!object->engine->process(num_frames, pcm, enhanced, &err)) {
            push_error(0x337, "%s", err.c_str());
            push_error(0x12C, "Picovoice Error.");
            return PV_STATUS_RUNTIME_ERROR;
        }
        return PV_STATUS_SUCCESS;
    });
}

PV_API pv_status_t pv_koala_batch_process_chunk_async(pv_koala_batch_t *object, int32_t num_frames, const int16_t *pcm,
                                                      int16_t *enhanced) {
    t_stack.clear();
    if (!object) {
        push_error(0x64, "Argument `object` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    if (!pcm || !enhanced) {
        push_error(0x64, "Argument `%s` is NULL.", pcm ? "enhanced" : "pcm");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    if (num_frames <= 0 || num_frames > object->engine->max_frames()) {
        push_error(0x66, "`num_frames` %d is outside [1, %d].", num_frames, object->engine->max_frames());
        return PV_STATUS_INVALID_ARGUMENT;
    }
    return guarded([&] {
        std::string err;
        if (!object->engine->process_host_async(num_frames, pcm, enhanced, &err)) {
            push_error(0x33A, "%s", err.c_str());
            push_error(0x12C, "Picovoice Error.");
            return PV_STATUS_RUNTIME_ERROR;
        }
        return PV_STATUS_SUCCESS;
    });
}

PV_API pv_status_t pv_koala_batch_async_wait(pv_koala_batch_t *object, int32_t max_in_flight) {
    t_stack.clear();
    if (!object) {
        push_error(0x64, "Argument `object` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    if (max_in_flight < 0) {
        push_error(0x66, "`max_in_flight` %d is negative.", max_in_flight);
        return PV_STATUS_INVALID_ARGUMENT;
    }
    return guarded([&] {
        std::string err;
        if (!object->engine->async_wait(max_in_flight, &err)) {
            push_error(0x33B, "%s", err.c_str());
            return PV_STATUS_RUNTIME_ERROR;
        }
        return PV_STATUS_SUCCESS;
    });
}

PV_API pv_status_t pv_koala_batch_process(pv_koala_batch_t *object, const int16_t *pcm, int16_t *enhanced) {
    return pv_koala_batch_process_chunk(object, 1, pcm, enhanced);
}

PV_API pv_status_t pv_koala_batch_reset(pv_koala_batch_t *object, const uint8_t *stream_mask) {
    t_stack.clear();
    if (!object) {
        push_error(0x64, "Argument `object` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    return guarded([&] {
        std::string err;
        if (!object->engine->reset(stream_mask, &err)) {
            push_error(0x338, "%s", err.c_str());
            return PV_STATUS_RUNTIME_ERROR;
        }
        return PV_STATUS_SUCCESS;
    });
}

PV_API pv_status_t pv_koala_batch_num_streams(const pv_koala_batch_t *object, int32_t *num_streams) {
    t_stack.clear();
    if (!object || !num_streams) {
        push_error(0x64, "Argument `%s` is NULL.", object ? "num_streams" : "object");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    *num_streams = object->engine->num_streams();
    return PV_STATUS_SUCCESS;
}

PV_API pv_status_t pv_koala_batch_delay_sample(const pv_koala_batch_t *object, int32_t *delay_sample) {
    t_stack.clear();
    if (!object || !delay_sample) {
        push_error(0x64, "Argument `%s` is NULL.", object ? "delay_sample" : "object");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    *delay_sample = kns::kFrame;
    return PV_STATUS_SUCCESS;
}

PV_API pv_status_t pv_koala_batch_set_stream(pv_koala_batch_t *object, void *hip_stream) {
    t_stack.clear();
    if (!object) {
        push_error(0x64, "Argument `object` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    object->engine->set_stream((hipStream_t) hip_stream);
    return PV_STATUS_SUCCESS;
}

PV_API pv_status_t pv_koala_batch_host_alloc(int64_t num_bytes, void **memory) {
    t_stack.clear();
    if (!memory) {
        push_error(0x64, "Argument `memory` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    *memory = nullptr;
    if (num_bytes <= 0) {
        push_error(0x65, "`num_bytes` should be positive.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    if (kns::visible_gpu_count() <= 0) {
        push_error(0x33A, "No GPU is visible: page-locked memory needs the HIP runtime.");
        return PV_STATUS_RUNTIME_ERROR;
    }
    if (hipHostMalloc(memory, (size_t) num_bytes, hipHostMallocPortable) != hipSuccess) {
        (void) hipGetLastError();
        *memory = nullptr;
        push_error(0x33B, "Failed to allocate %lld bytes of page-locked memory.", (long long) num_bytes);
        return PV_STATUS_OUT_OF_MEMORY;
    }
    return PV_STATUS_SUCCESS;
}

PV_API void pv_koala_batch_host_free(void *memory) {
    if (memory) (void) hipHostFree(memory);
}

PV_API pv_status_t pv_koala_batch_synchronize(pv_koala_batch_t *object) {
    t_stack.clear();
    if (!object) {
        push_error(0x64, "Argument `object` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    std::string err;
    if (!object->engine->synchronize(&err)) {
        push_error(0x339, "%s", err.c_str());
        return PV_STATUS_RUNTIME_ERROR;
    }
    return PV_STATUS_SUCCESS;
}

PV_API pv_status_t pv_koala_batch_profile_enable(pv_koala_batch_t *object, int32_t enable) {
    t_stack.clear();
    if (!object) {
        push_error(0x64, "Argument `object` is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    object->engine->profile_enable(enable != 0);
    return PV_STATUS_SUCCESS;
}

PV_API pv_status_t pv_koala_batch_profile_read(pv_koala_batch_t *object, double *milliseconds, int64_t *launches) {
    t_stack.clear();
    if (!object || !milliseconds || !launches) {
        push_error(0x64, "Argument is NULL.");
        return PV_STATUS_INVALID_ARGUMENT;
    }
    std::string err;
    if (!object->engine->profile_read(milliseconds, launches, &err)) {
        push_error(0x339, "%s", err.c_str());
        return PV_STATUS_RUNTIME_ERROR;
    }
    return PV_STATUS_SUCCESS;
}

PV_API int64_t pv_koala_batch_debug_read(pv_koala_batch_t *object, int32_t what, float *out, int64_t capacity) {
    t_stack.clear();
    if (!object || !out) return -(int64_t) PV_STATUS_INVALID_ARGUMENT;
    std::string err;
    int64_t n = -1;
    try {
        n = object->engine->debug_read(what, out, capacity, &err);
    } catch (...) {
        err = "Failed to allocate memory.";
    }
    if (n < 0) {
        push_error(0x33A, "%s", n == -2 ? "capacity too small" : err.c_str());
        return -(int64_t) PV_STATUS_INVALID_ARGUMENT;
    }
    return n;
}

#ifdef KNS_TIMING
PV_API void pv_koala_debug_timing(unsigned long long *out) { kns::read_timing(out); }
#endif

}  // extern "C"
