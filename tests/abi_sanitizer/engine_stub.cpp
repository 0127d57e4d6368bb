// Stand-in for kns_engine.cpp in the sanitizer build of the C-ABI shim (tests/test_abi_sanitized.py; SURVEY.md section 5: "ASan/UBSan
// build of the shim").  Only koala_amd/csrc/pv_api.cpp is under test here -- argument checks, the thread-local error stack, string
// and list ownership -- so the engine behind it is a host-only double: parameters "load" when the file exists, a handle is a plain
// object, process() copies its input, and no HIP call reaches a GPU.  TEST INFRASTRUCTURE: never linked into the product.
#include <stdio.h>
#include <string.h>

#include "kns_engine.h"

namespace kns {

LoadResult load_params(const char *path, Params *, std::string *err) {
    FILE *f = fopen(path, "rb");
    if (!f) {
        *err = std::string("Failed to open file `") + path + "`.";
        return kLoadIo;
    }
    char magic[8] = {0};
    const size_t n = fread(magic, 1, 8, f);
    fclose(f);
    if (n != 8 || memcmp(magic, "KNS1\0\0\0\0", 8) != 0) {
        *err = std::string("`") + path + "` is not a Koala (KNS1) model file.";
        return kLoadFormat;
    }
    return kLoadOk;
}

static int g_gpus = 1;  // STUB_GPUS=0 in the environment: "no GPU visible"
int visible_gpu_count() {
    const char *e = getenv("STUB_GPUS");
    return e ? atoi(e) : g_gpus;
}
std::string gpu_name(int) { return "Stub GPU"; }

Engine *Engine::create(const Params &, int device, int num_streams, int max_frames, int precision, std::string *err, bool *oom) {
    *oom = false;
    if (getenv("STUB_OOM")) {
        *oom = true;
        *err = "Failed to allocate device memory.";
        return nullptr;
    }
    Engine *e = new Engine();
    e->device_ = device;
    e->B_ = num_streams;
    e->Tmax_ = max_frames;
    e->prec_ = precision;
    return e;
}
Engine::~Engine() {}
bool Engine::process(int T, const int16_t *pcm, int16_t *out, std::string *err, bool) {
    if (getenv("STUB_FAIL_PROCESS")) {
        *err = "HIP error: stub";
        return false;
    }
    if (getenv("STUB_THROW")) throw std::bad_alloc();
    memmove(out, pcm, (size_t) B_ * T * kFrame * 2);
    return true;
}
bool Engine::process_host_async(int T, const int16_t *pcm, int16_t *out, std::string *err) { return process(T, pcm, out, err, true); }
bool Engine::drain_async(std::string *) { return true; }
bool Engine::async_wait(int, std::string *) { return true; }
bool Engine::reset(const uint8_t *, std::string *) { return true; }
bool Engine::synchronize(std::string *) { return true; }
void Engine::set_stream(hipStream_t s) { stream_ = s ? s : own_stream_; }
size_t Engine::shared_device_bytes() const { return 0; }
void Engine::profile_enable(bool on) { profiling_ = on; }
bool Engine::profile_read(double *ms, int64_t *launches, std::string *) {
    for (int i = 0; i < kNumKernelClasses; ++i) ms[i] = 0.0, launches[i] = 0;
    return true;
}
int64_t Engine::debug_read(int, float *, int64_t, std::string *err) {
    *err = "unknown debug tap";
    return -1;
}

}  // namespace kns

// the three HIP entry points pv_api.cpp calls directly (page-locked buffers): host memory, no runtime
extern "C" {
hipError_t hipHostMalloc(void **ptr, size_t size, unsigned int) {
    *ptr = malloc(size);
    return *ptr ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *ptr) {
    free(ptr);
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
}
