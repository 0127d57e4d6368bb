// Drives every entry point of the C ABI (include/pv_koala.h, include/picovoice.h, include/pv_koala_batch.h) through its argument
// checks, error-stack and ownership paths under AddressSanitizer + UndefinedBehaviorSanitizer, with the engine stubbed out
// (engine_stub.cpp).  Exit status 0 = every expectation held and the sanitizers stayed silent.  usage: driver <model.kns> <not-a-model>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "pv_koala.h"
#include "pv_koala_batch.h"

extern "C" {
void pv_set_sdk(const char *);
const char *pv_get_sdk(void);
void pv_free(void *);
void pv_log_enable(void);
void pv_log_disable(void);
}

static int g_fail = 0;
#define EXPECT(cond)                                              \
    do {                                                          \
        if (!(cond)) {                                            \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++g_fail;                                             \
        }                                                         \
    } while (0)

// drains the thread's stack; returns its depth and the first message
static int drain(std::string *first = nullptr) {
    char **stack = nullptr;
    int32_t depth = -1;
    const pv_status_t st = pv_get_error_stack(&stack, &depth);
    if (depth > 0) {
        EXPECT(st == PV_STATUS_SUCCESS && stack != nullptr);
        for (int i = 0; i < depth; ++i) EXPECT(stack[i] != nullptr && strlen(stack[i]) > 17);  // "<7 hex> <8 hex>: text"
        if (first) *first = stack[0];
        pv_free_error_stack(stack);
    } else {
        EXPECT(st == PV_STATUS_INVALID_STATE && depth == 0 && stack == nullptr);  // nothing pending: nothing to free
    }
    return depth;
}

static void single_stream(const char *model, const char *garbage) {
    pv_koala_t *h = nullptr;
    std::string msg;
    EXPECT(pv_koala_init(nullptr, model, "best", &h) == PV_STATUS_INVALID_ARGUMENT && drain(&msg) == 1 && msg.find("`access_key`") != std::string::npos);
    EXPECT(pv_koala_init("k", nullptr, "best", &h) == PV_STATUS_INVALID_ARGUMENT && drain(&msg) == 1 && msg.find("`model_path`") != std::string::npos);
    EXPECT(pv_koala_init("k", model, nullptr, &h) == PV_STATUS_INVALID_ARGUMENT && drain(&msg) == 1 && msg.find("`device`") != std::string::npos);
    EXPECT(pv_koala_init("k", model, "best", nullptr) == PV_STATUS_INVALID_ARGUMENT && drain(&msg) == 1 && msg.find("`object`") != std::string::npos);
    EXPECT(pv_koala_init("k", "/nonexistent/x.kns", "best", &h) == PV_STATUS_IO_ERROR && drain(&msg) == 2 && msg.find("Failed to open file") != std::string::npos);
    EXPECT(pv_koala_init("k", garbage, "best", &h) == PV_STATUS_IO_ERROR && drain() == 2);
    EXPECT(pv_koala_init("k", model, "foo", &h) == PV_STATUS_INVALID_ARGUMENT && drain(&msg) == 2 && msg.find("foo is not a valid device string") != std::string::npos);
    EXPECT(pv_koala_init("k", model, "", &h) == PV_STATUS_INVALID_ARGUMENT && drain() == 2);
    EXPECT(pv_koala_init("k", model, "gpu:", &h) == PV_STATUS_INVALID_ARGUMENT && drain() == 2);
    EXPECT(pv_koala_init("k", model, "gpu:99999999999999999999", &h) == PV_STATUS_INVALID_ARGUMENT && drain() == 2);
    EXPECT(pv_koala_init("k", model, "cpu:4", &h) == PV_STATUS_RUNTIME_ERROR && drain() == 2);
    EXPECT(pv_koala_init("k", model, "gpu:7", &h) == PV_STATUS_INVALID_ARGUMENT && drain(&msg) == 2 && msg.find("out of range") != std::string::npos);
    EXPECT(pv_koala_init("", model, "best", &h) == PV_STATUS_INVALID_ARGUMENT && drain(&msg) == 2 && msg.find("AccessKey") != std::string::npos);
    // a very long device string must not overrun the message buffers
    std::string huge(5000, 'x');
    EXPECT(pv_koala_init("k", model, huge.c_str(), &h) == PV_STATUS_INVALID_ARGUMENT && drain() == 2);
    std::string huge_path = "/nonexistent/" + std::string(4000, 'p');
    EXPECT(pv_koala_init("k", huge_path.c_str(), "best", &h) == PV_STATUS_IO_ERROR && drain() == 2);
    setenv("STUB_GPUS", "0", 1);
    EXPECT(pv_koala_init("k", model, "best", &h) == PV_STATUS_RUNTIME_ERROR && drain() == 2);
    unsetenv("STUB_GPUS");
    setenv("STUB_OOM", "1", 1);
    EXPECT(pv_koala_init("k", model, "best", &h) == PV_STATUS_OUT_OF_MEMORY && drain() == 2);
    unsetenv("STUB_OOM");

    EXPECT(pv_koala_init("k", model, "gpu:0 - Stub GPU", &h) == PV_STATUS_SUCCESS && h != nullptr);
    EXPECT(drain() == 0);
    int16_t in[256], out[256];
    for (int i = 0; i < 256; ++i) in[i] = (int16_t) (i * 37);
    EXPECT(pv_koala_process(nullptr, in, out) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_process(h, nullptr, out) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_process(h, in, nullptr) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_process(h, in, out) == PV_STATUS_SUCCESS && memcmp(in, out, sizeof(in)) == 0);
    EXPECT(pv_koala_process(h, in, in) == PV_STATUS_SUCCESS);  // in place
    setenv("STUB_FAIL_PROCESS", "1", 1);
    EXPECT(pv_koala_process(h, in, out) == PV_STATUS_RUNTIME_ERROR && drain() == 2);
    unsetenv("STUB_FAIL_PROCESS");
    setenv("STUB_THROW", "1", 1);  // an exception inside the engine must come back as a status, not cross the C ABI
    EXPECT(pv_koala_process(h, in, out) == PV_STATUS_OUT_OF_MEMORY && drain() == 1);
    unsetenv("STUB_THROW");
    // a failure's messages are replaced by the next call's (the stack describes the LAST call of the thread)
    EXPECT(pv_koala_process(nullptr, in, out) == PV_STATUS_INVALID_ARGUMENT);
    EXPECT(pv_koala_process(h, in, out) == PV_STATUS_SUCCESS && drain() == 0);
    int32_t delay = -1;
    EXPECT(pv_koala_delay_sample(nullptr, &delay) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_delay_sample(h, nullptr) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_delay_sample(h, &delay) == PV_STATUS_SUCCESS && delay == 256);
    EXPECT(pv_koala_reset(nullptr) == PV_STATUS_INVALID_ARGUMENT);
    EXPECT(pv_koala_reset(h) == PV_STATUS_SUCCESS);
    pv_koala_delete(h);
    pv_koala_delete(nullptr);

    EXPECT(pv_koala_frame_length() == 256 && pv_sample_rate() == 16000 && !strcmp(pv_koala_version(), "3.0.0"));
    for (int s = -2; s < 16; ++s) EXPECT((pv_status_to_string((pv_status_t) s) != nullptr) == (s >= 0 && s <= 11));
    EXPECT(!strcmp(pv_status_to_string(PV_STATUS_SUCCESS), "SUCCESS"));
    EXPECT(pv_get_error_stack(nullptr, nullptr) != PV_STATUS_SUCCESS);
    pv_free_error_stack(nullptr);
    pv_free(nullptr);
    pv_free(malloc(8));
    EXPECT(!strcmp(pv_get_sdk(), "c"));
    pv_set_sdk("python");
    EXPECT(!strcmp(pv_get_sdk(), "python"));
    pv_set_sdk(nullptr);
    pv_set_sdk(std::string(3000, 's').c_str());
    EXPECT(strlen(pv_get_sdk()) > 0);
    pv_log_enable();
    pv_log_disable();

    char **devs = nullptr;
    int32_t ndev = -1;
    EXPECT(pv_koala_list_hardware_devices(nullptr, &ndev) == PV_STATUS_INVALID_ARGUMENT);
    EXPECT(pv_koala_list_hardware_devices(&devs, nullptr) == PV_STATUS_INVALID_ARGUMENT);
    (void) drain();
    EXPECT(pv_koala_list_hardware_devices(&devs, &ndev) == PV_STATUS_SUCCESS && ndev >= 1 && devs != nullptr);
    for (int i = 0; i < ndev; ++i) EXPECT(devs[i] != nullptr && strlen(devs[i]) > 0);
    pv_koala_free_hardware_devices(devs, ndev);
    pv_koala_free_hardware_devices(nullptr, 0);
}

static void batch(const char *model) {
    pv_koala_batch_t *b = nullptr;
    EXPECT(pv_koala_batch_init("k", model, "best", 0, 4, PV_KOALA_PRECISION_BF16, &b) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_init("k", model, "best", 3, 0, PV_KOALA_PRECISION_BF16, &b) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_init("k", model, "best", 3, 4, (pv_koala_precision_t) 7, &b) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_init("k", model, "best", 3, 4, PV_KOALA_PRECISION_BF16, nullptr) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_init("k", model, "best", 3, 4, PV_KOALA_PRECISION_FP32, &b) == PV_STATUS_SUCCESS && b != nullptr);
    std::vector<int16_t> in(3 * 4 * 256, 5), out(3 * 4 * 256, 0);
    EXPECT(pv_koala_batch_process_chunk(nullptr, 4, in.data(), out.data()) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_process_chunk(b, 4, nullptr, out.data()) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_process_chunk(b, 4, in.data(), nullptr) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_process_chunk(b, 0, in.data(), out.data()) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_process_chunk(b, 5, in.data(), out.data()) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_process_chunk(b, 4, in.data(), out.data()) == PV_STATUS_SUCCESS && out == in);
    EXPECT(pv_koala_batch_process(b, in.data(), out.data()) == PV_STATUS_SUCCESS);
    uint8_t mask[3] = {1, 0, 1};
    EXPECT(pv_koala_batch_reset(nullptr, mask) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_reset(b, mask) == PV_STATUS_SUCCESS && pv_koala_batch_reset(b, nullptr) == PV_STATUS_SUCCESS);
    int32_t n = 0, d = 0;
    EXPECT(pv_koala_batch_num_streams(b, &n) == PV_STATUS_SUCCESS && n == 3);
    EXPECT(pv_koala_batch_num_streams(b, nullptr) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_num_streams(nullptr, &n) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_delay_sample(b, &d) == PV_STATUS_SUCCESS && d == 256);
    EXPECT(pv_koala_batch_delay_sample(nullptr, &d) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_set_stream(nullptr, nullptr) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_set_stream(b, nullptr) == PV_STATUS_SUCCESS);
    EXPECT(pv_koala_batch_synchronize(nullptr) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_synchronize(b) == PV_STATUS_SUCCESS);
    double ms[5];
    int64_t launches[5];
    EXPECT(pv_koala_batch_profile_enable(nullptr, 1) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_profile_enable(b, 1) == PV_STATUS_SUCCESS);
    EXPECT(pv_koala_batch_profile_read(b, nullptr, launches) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_profile_read(b, ms, launches) == PV_STATUS_SUCCESS);
    float taps[4];
    EXPECT(pv_koala_batch_debug_read(nullptr, 0, taps, 4) < 0);
    EXPECT(pv_koala_batch_debug_read(b, 0, nullptr, 4) < 0);
    EXPECT(pv_koala_batch_debug_read(b, 99, taps, 4) < 0);
    void *pin = nullptr;
    EXPECT(pv_koala_batch_host_alloc(0, &pin) == PV_STATUS_INVALID_ARGUMENT && pin == nullptr && drain() == 1);
    EXPECT(pv_koala_batch_host_alloc(64, nullptr) == PV_STATUS_INVALID_ARGUMENT && drain() == 1);
    EXPECT(pv_koala_batch_host_alloc(64, &pin) == PV_STATUS_SUCCESS && pin != nullptr);
    memset(pin, 0, 64);
    pv_koala_batch_host_free(pin);
    pv_koala_batch_host_free(nullptr);
    pv_koala_batch_delete(b);
    pv_koala_batch_delete(nullptr);
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    single_stream(argv[1], argv[2]);
    batch(argv[1]);
    // the error stack is per thread: failures on other threads leave this one's untouched, and eight threads hammering the
    // argument checks must not race (the stack is thread_local, the SDK string is behind a mutex)
    EXPECT(pv_koala_process(nullptr, nullptr, nullptr) == PV_STATUS_INVALID_ARGUMENT);
    std::vector<std::thread> pool;
    for (int t = 0; t < 8; ++t)
        pool.emplace_back([&, t] {
            for (int i = 0; i < 200; ++i) {
                pv_koala_t *h = nullptr;
                if (pv_koala_init("k", "/nonexistent", "best", &h) != PV_STATUS_IO_ERROR || drain() != 2) ++g_fail;
                pv_set_sdk(t & 1 ? "a" : "b");
                (void) pv_get_sdk();
            }
            if (drain() != 0) ++g_fail;
        });
    for (auto &th : pool) th.join();
    EXPECT(drain() == 1);  // this thread's message from before the pool is still here
    if (g_fail) fprintf(stderr, "%d expectation(s) failed\n", g_fail);
    return g_fail ? 1 : 0;
}
