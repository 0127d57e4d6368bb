// kns_engine.h -- host-side engine behind both the single-stream and the batch C ABI.
// One Engine = B lock-stepped streams on one GPU: parameter upload (pre-packed for MFMA), per-stream state in HBM,
// activation workspace in fragment layouts, and the per-chunk kernel sequence.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <memory>
#include <string>
#include <vector>

#include "kns_kernels.h"

namespace kns {

// host copy of a KNS1 parameter file (fp32, logical layout; see koala_amd/params.py)
struct Params {
    int front_taps = 1;  // feature frames the front-end sees (KNS-v1: 1; KNS-v1.1: up to 5, w_in then has front_taps * 257 rows, oldest frame first)
    int head[kStages];
    std::vector<float> mean, scale, w_in, b_in;
    struct Stage {
        int d_in, d_out;
        std::vector<float> w_ih_a, b_ih_a, w_hh_a, b_hh_a, w_ih_b, b_ih_b, w_hh_b, b_hh_b, w_head, b_head;
    } st[kStages];
};

enum LoadResult { kLoadOk = 0, kLoadIo = 1, kLoadFormat = 2 };
LoadResult load_params(const char *path, Params *out, std::string *err);

constexpr int kNumKernelClasses = 5;
enum KernelClass { kClsAnalysis = 0, kClsGemmIn = 1, kClsGru = 2, kClsGemmHead = 3, kClsSynthesis = 4 };

class Engine {
public:
    // returns nullptr and fills *err on failure (*oom set when the failure was an allocation)
    static Engine *create(const Params &p, int device, int num_streams, int max_frames, int precision,
                          std::string *err, bool *oom);
    ~Engine();

    // bytes of device memory this handle allocated itself / bytes of the weight image it shares (first handle on a model: it built it)
    size_t own_device_bytes() const { return own_bytes_; }
    size_t shared_device_bytes() const;
    bool weights_were_cached() const { return weights_cached_; }

    int num_streams() const { return B_; }
    int max_frames() const { return Tmax_; }
    int device() const { return device_; }

    // pcm/out: [B][T*256]; host or device pointers (both of the same kind).  Host: synchronous.  Device: enqueued.
    bool process(int T, const int16_t *pcm, int16_t *out, std::string *err, bool host_pointers = false);
    // Page-locked host buffers, asynchronous: the call's copy-in, kernels and copy-out are enqueued on three streams and the function
    // returns; up to three such calls are in flight (a fourth first waits for the oldest), so the copies of one call run under the
    // kernels of its neighbours.  `drain_async` (also reached through synchronize(), and entered by every other entry point) waits
    // for all of them.
    bool process_host_async(int T, const int16_t *pcm, int16_t *out, std::string *err);
    bool drain_async(std::string *err);
    bool async_wait(int max_in_flight, std::string *err);  // until at most that many asynchronous calls are still in flight (0: all done)
    bool reset(const uint8_t *host_mask, std::string *err);
    // 0: back to the handle's own stream.  Waits for everything the handle has in flight first (asynchronous host calls and the work
    // queued on the previous stream, which must still exist: the next call's kernels touch the same state, history and tail buffers)
    void set_stream(hipStream_t s);
    bool synchronize(std::string *err);

    void profile_enable(bool on);
    bool profile_read(double *ms, int64_t *launches, std::string *err);
    int64_t debug_read(int what, float *out, int64_t capacity, std::string *err);

private:
    Engine() {}
    bool init(const Params &p, int device, int B, int Tmax, int precision, std::string *err, bool *oom);
    bool run_device(int T, const int16_t *d_pcm, int16_t *d_out, std::string *err, bool allow_recompute = true);
    void *dalloc(size_t bytes, bool zero);
    void *upload(const void *src, size_t bytes);
    void tick(int cls);
    void tock(int cls);

    // The immutable part of a handle -- tables and every packed weight matrix -- is built once per (model content, device, precision,
    // developer switches) and SHARED by all handles that are open on it (round 6: the reference's contract is one handle per stream,
    // include/pv_koala.h:26-63; a caller with N handles used to get N folds, N packings and N weight images).  Ref-counted: freed with the
    // last handle.  A handle copies the image's pointers into its own members below; only its allocations differ.
    struct WeightImage;
    std::shared_ptr<WeightImage> weights_;
    std::vector<void *> *alloc_sink_ = nullptr;  // while an image is being built: where dalloc() records its allocations
    bool build_weights(const Params &p, int precision, std::string *err);

    int device_ = 0, B_ = 0, Bpad_ = 0, Tmax_ = 0, prec_ = 0, last_T_ = 0;
    PrecInfo pi_{};
    int nbf_ = 0, nbh_ = 0, nby_[kStages] = {0, 0, 0, 0};
    hipStream_t own_stream_ = nullptr, stream_ = nullptr;
    bool alloc_failed_ = false, weights_cached_ = false;
    size_t own_bytes_ = 0;
    std::vector<void *> allocs_;

    // parameters on device
    float *d_window_ = nullptr, *d_twiddle_ = nullptr, *d_mean_ = nullptr, *d_scale_ = nullptr;
    void *w_in_ = nullptr;
    float *b_in_ = nullptr;
    struct StageDev {
        void *w_ih_a, *w_hh_a, *w_ih_b, *w_hh_b, *w_head;
        float *b_ih_a, *b_hh_a, *b_ih_b, *b_hh_b, *b_head;
        int head_tiles, head_dim;
        bool ypad;  // this stage's fed-forward input rides in the padding of the features' last k-block (bf16, folded front-end, d_in <= kYPadMax)
    } sd_[kStages]{};

    // per-stream state
    int16_t *d_hist_[2] = {nullptr, nullptr};
    int hist_cur_ = 0;
    float *d_tail_[2] = {nullptr, nullptr}, *d_hstate_[2] = {nullptr, nullptr};
    void *d_hprev_ = nullptr;  // the recurrent state as A-packed operand blocks [layer][m-tile][nbh] (kRouteWave)
    int tail_cur_ = 0, hs_cur_ = 0;
    uint8_t *d_rmask_ = nullptr;
    // front-end context (front_taps > 1): features of the last front_taps - 1 frames (A-packed, one "frame" = mtiles x nbf
    // blocks), and one m-tile of the feature of a silent frame (what the context holds before a stream began)
    int taps_ = 1;
    bool fold_ = false;  // bf16, one-frame front-end: folded into the stage-input GEMMs (no front-end launch, no embedding buffer)
    void *d_fhist_ = nullptr, *d_silent_ = nullptr;
    size_t feat_frame_bytes_ = 0;

    // activation workspace (fragment layouts)
    float *d_spec_ = nullptr, *d_mask_ = nullptr;
    void *d_feat_ = nullptr, *d_e_ = nullptr, *d_y_[kStages - 1] = {nullptr, nullptr, nullptr}, *d_gi_ = nullptr,
         *d_hseq_a_ = nullptr, *d_hseq_b_ = nullptr;

    // staging for host-pointer calls: [B][Tmax * 256] each; chunked calls use them as two slots of [B][Tc * 256]
    int16_t *d_in_ = nullptr, *d_out_ = nullptr, *h_in_ = nullptr, *h_out_ = nullptr;
    // host-pointer calls with more than one sub-chunk: copy-in, compute and copy-out run on three streams
    bool process_host_pipelined(int T, const int16_t *pcm, int16_t *out, bool pinned, std::string *err);
    std::vector<int> host_schedule(int T) const;  // its sub-chunk lengths
    std::vector<int> dev_host_sched_;             // developer override (KOALA_AMD_HOST_SCHED)
    // calls of several frames as a wavefront over (stage, frame): kns_engine.cpp, run_wave
    GruSmallArgs small_args(int mtb, const void *a0, int nb0, const void *a1, const void *wih, const float *bih, const void *whh,
                            const float *bhh, int layer, void *hseq, int t, const StageDev *head) const;
    GruWaveItem wave_item(int i, int t, int mtb) const;
    bool wave_fits() const;
    // mid-size batches: the (layer, chunk-of-frames) grid of a call as a wavefront over a few streams (kns_engine.cpp, run_device)
    static constexpr int kPipeStreams = 4, kPipeRing = 2;
    hipStream_t pipe_stream_[kPipeStreams] = {};
    hipEvent_t pipe_fork_ = nullptr, pipe_join_[kPipeStreams] = {}, pipe_ev_[kPipeRing][kGruLayers] = {}, pipe_syn_[kPipeRing] = {};
    bool pipe_ok_ = false, pipe_failed_ = false, dev_pipe_whole_stft_ = false;
    bool pipe_ready();
    void run_wave(int T, int mtb);
    // asynchronous host calls: two slots of full-size device staging (slot 0 = d_in_ / d_out_, slot 1 allocated on first use)
    int16_t *d_in2_ = nullptr, *d_out2_ = nullptr;
    bool async_ready_ = false;  // every event and both staging slots of the asynchronous host path exist
    hipEvent_t aev_in_[2] = {nullptr, nullptr}, aev_done_[2] = {nullptr, nullptr}, aev_out_[4] = {nullptr, nullptr, nullptr, nullptr};
    bool async_busy_[4] = {false, false, false, false};  // by call number mod 4: the window is three calls
    unsigned async_n_ = 0;
    hipStream_t copy_in_ = nullptr, copy_out_ = nullptr;
    hipEvent_t host_fork_ = nullptr;  // synchronous host calls on a caller's stream: the handle's own stream waits behind it
    hipEvent_t ev_in_[2] = {nullptr, nullptr}, ev_done_[2] = {nullptr, nullptr}, ev_out_[2] = {nullptr, nullptr};
    int host_chunk_ = 1;
    size_t host_pipeline_min_bytes_ = 0;

    // hipGraph of one host-pointer frame (copy-in, 23 kernels, copy-out); built on first use
    hipGraphExec_t frame_graph_[8] = {};  // one per combination of the hidden-state / history / tail ping-pong indices
    // completion word of the zero-copy one-frame replays (kns_stft.hip, frame_done_kernel): the host spins on a word in page-locked
    // memory instead of sleeping in hipStreamSynchronize, whose wake-up is what made p99 drift away from p50 on a busy host
    unsigned *d_frame_count_ = nullptr, *h_frame_word_ = nullptr;
    unsigned frame_seq_ = 0;
    bool frame_graph_signals_[8] = {};
    bool spin_wait_ = true;
    bool use_graph_ = true, no_small_ = false, no_zero_copy_ = false, no_recompute_ = false, debug_taps_ = false;
    // developer switches (all read once in init() through dev_env(): compiled out of the product library)
    int dev_variant_ = 0, dev_only_class_ = -1, dev_analysis_seg_ = 0, dev_synth_seg_ = 0, dev_small_mt_ = 0, dev_steps_mt_ = 192, dev_wave_mt_ = -1, dev_wave_group_ = 0, dev_wave_parts_ = 1, dev_pipe_chunk_ = 0, dev_pipe_mt_ = -1, dev_pipe_grid_ = 0, dev_pipe_streams_ = 0;
    // one-frame calls: GRU layers fused over CU quads (kns_gruq.hip); narrow heads / front-end / mask head inside their consumers
    bool use_quad_ = true;
    int quad_nb0_max_ = 2;
    bool fuse_head_ = true, fuse_front_ = true;
    unsigned long long *d_qdbg_ = nullptr;  // developer build, KOALA_AMD_QUAD_DBG=<block>: stamps of the LAST fused launch
    int qdbg_block_ = -1;
    // the last run_device() stored the spectrum / the features / the mask (debug_read refuses a tap that was not stored)
    bool spec_valid_ = false, feat_valid_ = false, mask_valid_ = false;
    int last_route_ = 0;  // enum Route of the last run_device() (kns_engine.cpp; reported by the developer build's debug tap 6)

    // profiling
    bool profiling_ = false;
    struct Span {
        int cls;
        hipEvent_t a, b;
    };
    std::vector<Span> spans_;
    std::vector<hipEvent_t> pool_;
    hipEvent_t pending_ = nullptr;
    double acc_ms_[kNumKernelClasses] = {0, 0, 0, 0, 0};
    int64_t acc_n_[kNumKernelClasses] = {0, 0, 0, 0, 0};
};

int visible_gpu_count();
std::string gpu_name(int device);

}  // namespace kns
