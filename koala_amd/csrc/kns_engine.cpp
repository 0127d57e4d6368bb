// kns_engine.cpp -- parameter loading/packing, HBM workspace and the per-chunk kernel sequence of the KNS-v1 engine.
//
// Sequence for one call of B streams x T frames (fp32: 23 launches, independent of T):
//   analysis -> front-end GEMM -> 4 x { input GEMM A -> recurrent A -> input GEMM B -> recurrent B -> head GEMM } -> synthesis
// (bf16 configuration, one-frame front-end: 20 -- the front-end is folded into the stage-input GEMMs and the narrow heads of stages 0
// and 1 ride in their layer-B recurrent launches; several frames of few streams: T + 14 -- the layers as a wavefront over (stage,
// frame); the dispatch table stands in front of Engine::run_device)
// which is the batched form of what one pv_koala_process call does for one stream and one frame
// (reference include/pv_koala.h:65-80; stage structure per lib/common/koala_params.pv).
#include "kns_engine.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

namespace kns {

// ------------------------------------------------------------------------------------------------ parameter file

namespace {

bool read_vec(FILE *f, std::vector<float> *v, size_t n) {
    v->resize(n);
    return fread(v->data(), sizeof(float), n, f) == n;
}

}  // namespace

LoadResult load_params(const char *path, Params *p, std::string *err) {
    FILE *f = fopen(path, "rb");
    if (!f) {
        *err = std::string("Failed to open file `") + path + "`.";
        return kLoadIo;
    }
    char magic[8];
    uint32_t hdr[14];
    bool ok = fread(magic, 1, 8, f) == 8;
    if (ok && memcmp(magic, "koala", 5) == 0) {
        // the reference's own parameter file (lib/common/koala_params.pv, magic "koala3.0.0"): closed fixed-point format
        fclose(f);
        *err = std::string("`") + path + "` is a reference Koala `.pv` model: its fixed-point format is not documented and "
               "is not supported; pass a KNS1 (.kns) parameter file.";
        return kLoadFormat;
    }
    ok = ok && fread(hdr, 4, 14, f) == 14 && memcmp(magic, "KNS1\0\0\0\0", 8) == 0;
    ok = ok && hdr[0] == 1 && hdr[1] == kNfft && hdr[2] == kFrame && hdr[3] == kBins && hdr[4] == kHidden &&
         hdr[5] == kStages && hdr[9] == kBins && hdr[10] == kFrame;
    if (!ok) {
        fclose(f);
        *err = std::string("`") + path + "` is not a Koala (KNS1) model file.";
        return kLoadFormat;
    }
    // KNS-v1.1: a front-end over the last N feature frames (header word 11; 0 and 1 both mean one).  Checked as the unsigned
    // word it is: a value with the top bit set must not become a negative count that passes the test and sizes a vector.
    p->front_taps = hdr[11] > 1 && hdr[11] <= (uint32_t) kMaxFrontTaps ? (int) hdr[11] : 1;
    if (hdr[11] > (uint32_t) kMaxFrontTaps) {
        fclose(f);
        *err = std::string("`") + path + "` has a front-end over " + std::to_string(hdr[11]) + " feature frames (at most " +
               std::to_string(kMaxFrontTaps) + " are supported).";
        return kLoadFormat;
    }
    if (hdr[12] != 0) {  // the CPU oracle's fixed-point emulation (tools/pv_hypotheses.py): not an engine feature
        fclose(f);
        *err = std::string("`") + path + "` asks for re-quantised pre-activations (KNS1 header word 12), which only the CPU oracle emulates.";
        return kLoadFormat;
    }
    const size_t G3 = 3 * kHidden;
    ok = read_vec(f, &p->mean, kBins) && read_vec(f, &p->scale, kBins) &&
         read_vec(f, &p->w_in, (size_t) p->front_taps * kBins * kHidden) && read_vec(f, &p->b_in, kHidden);
    for (int s = 0; ok && s < kStages; ++s) {
        Params::Stage &st = p->st[s];
        p->head[s] = (int) hdr[6 + s];
        st.d_in = s ? p->head[s - 1] : 0;
        st.d_out = p->head[s];
        if (st.d_out <= 0 || st.d_out > kBins) {
            ok = false;
            break;
        }
        ok = read_vec(f, &st.w_ih_a, (size_t) (st.d_in + kHidden) * G3) && read_vec(f, &st.b_ih_a, G3) &&
             read_vec(f, &st.w_hh_a, (size_t) kHidden * G3) && read_vec(f, &st.b_hh_a, G3) &&
             read_vec(f, &st.w_ih_b, (size_t) kHidden * G3) && read_vec(f, &st.b_ih_b, G3) &&
             read_vec(f, &st.w_hh_b, (size_t) kHidden * G3) && read_vec(f, &st.b_hh_b, G3) &&
             read_vec(f, &st.w_head, (size_t) kHidden * st.d_out) && read_vec(f, &st.b_head, (size_t) st.d_out);
    }
    ok = ok && fgetc(f) == EOF;
    fclose(f);
    if (!ok) {
        *err = "Failed to read parameter from file.";
        return kLoadFormat;
    }
    return kLoadOk;
}

// ------------------------------------------------------------------------------------------------ weight packing

namespace {

struct Seg {
    int k0, klen;
};
struct Tile {
    int c0, valid;
};

uint16_t to_bf16(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t) (u >> 16);
}

std::vector<Tile> dense_tiles(int n, int multiple) {
    std::vector<Tile> t;
    for (int c = 0; c < n; c += 16) t.push_back({c, n - c < 16 ? n - c : 16});
    while (t.size() % (size_t) multiple) t.push_back({0, 0});
    return t;
}

std::vector<Tile> gru_tiles() {  // [unit tile][gate]: what one lane needs for a hidden unit sits in adjacent tiles
    std::vector<Tile> t;
    for (int u = 0; u < kUnitTiles; ++u)
        for (int g = 0; g < 3; ++g) {
            int left = kHidden - u * 16;
            t.push_back({g * kHidden + u * 16, left < 16 ? left : 16});
        }
    return t;
}

// B-packed image of W[k][n] (row-major, leading dimension ldw): [n-tile][k-block][lane][16 B]
std::vector<uint8_t> pack_b(const float *W, int ldw, const std::vector<Seg> &segs, const std::vector<Tile> &tiles,
                            int precision) {
    const PrecInfo pi = prec_info(precision);
    int nb = 0;
    for (const Seg &s : segs) nb += ceil_div(s.klen, pi.kb);
    std::vector<uint8_t> out(tiles.size() * (size_t) nb * 1024, 0);
    for (size_t nt = 0; nt < tiles.size(); ++nt) {
        int blk = 0;
        for (const Seg &s : segs) {
            for (int b = 0; b < ceil_div(s.klen, pi.kb); ++b, ++blk) {
                uint8_t *dst = out.data() + (nt * nb + blk) * 1024;
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < pi.epl; ++e) {
                        const int kk = precision == kBf16 ? (lane >> 4) * 8 + e : e * 4 + (lane >> 4);
                        const int k = b * pi.kb + kk, n = lane & 15;
                        float v = 0.0f;
                        if (k < s.klen && n < tiles[nt].valid) v = W[(size_t) (s.k0 + k) * ldw + tiles[nt].c0 + n];
                        if (precision == kBf16) {
                            uint16_t h = to_bf16(v);
                            memcpy(dst + (lane * 8 + e) * 2, &h, 2);
                        } else {
                            memcpy(dst + (lane * 4 + e) * 4, &v, 4);
                        }
                    }
            }
        }
    }
    return out;
}

// row[n] = fma(a[j], W[j][n], row[n]), j ascending: one row of the front-end fold (4 x 257 x 271 x 813 fused multiply-adds per handle).
// Two bodies of the same IEEE operations: the x86-64-v3 one (vfmadd, ~0.1 s per handle) is entered only on a CPU that reports
// FMA and AVX2; the portable one calls fmaf (correct on any host, seconds per handle).  Nothing else in this file is compiled
// with wider ISA flags, and -ffp-contract=off keeps every other a * b + c of the packing code two roundings.
#if defined(__x86_64__)
__attribute__((target("fma,avx2"))) void fold_row_v3(float *row, const float *a, const float *W, int J, int N) {
    for (int j = 0; j < J; ++j) {
        const float aj = a[j];
        const float *wr = W + (size_t) j * N;
        for (int n = 0; n < N; ++n) row[n] = __builtin_fmaf(aj, wr[n], row[n]);
    }
}
#endif
void fold_row_portable(float *row, const float *a, const float *W, int J, int N) {
    for (int j = 0; j < J; ++j) {
        const float aj = a[j];
        const float *wr = W + (size_t) j * N;
        for (int n = 0; n < N; ++n) row[n] = fmaf(aj, wr[n], row[n]);
    }
}
void fold_row(float *row, const float *a, const float *W, int J, int N) {
#if defined(__x86_64__)
    static const bool v3 = __builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2");
    if (v3) return fold_row_v3(row, a, W, J, N);
#endif
    fold_row_portable(row, a, W, J, N);
}

// true if rows k >= k_from of a B-packed image with nb k-blocks per n-tile are all zero.  The bf16 recurrent kernels publish hidden
// sequences whose k = 271, 272 hold the constant 1 (the bias rows' operand, kBiasK0), and the narrow heads write their few values
// into the features' padding columns: every OTHER consumer of those operands relies on its packed weights being zero there.
bool rows_zero_from(const std::vector<uint8_t> &img, int nb, int k_from, int precision) {
    const PrecInfo pi = prec_info(precision);
    const size_t ntiles = img.size() / ((size_t) nb * 1024);
    for (size_t nt = 0; nt < ntiles; ++nt)
        for (int blk = k_from / pi.kb; blk < nb; ++blk)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < pi.epl; ++e) {
                    const int kk = precision == kBf16 ? (lane >> 4) * 8 + e : e * 4 + (lane >> 4);
                    if (blk * pi.kb + kk < k_from) continue;
                    const uint8_t *v = img.data() + (nt * nb + blk) * 1024 + (size_t) (lane * pi.epl + e) * pi.esz;
                    for (int q = 0; q < pi.esz; ++q)
                        if (v[q]) return false;
                }
    return true;
}

std::vector<float> pack_bias(const float *b, const std::vector<Tile> &tiles) {
    std::vector<float> out(tiles.size() * 16, 0.0f);
    for (size_t nt = 0; nt < tiles.size(); ++nt)
        for (int n = 0; n < tiles[nt].valid; ++n) out[nt * 16 + n] = b[tiles[nt].c0 + n];
    return out;
}

}  // namespace

int visible_gpu_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void) hipGetLastError();
        return 0;
    }
    return n;
}

std::string gpu_name(int device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        (void) hipGetLastError();
        return "unknown";
    }
    if (prop.name[0]) return prop.name;
    return std::string("AMD GPU ") + prop.gcnArchName;  // some ROCm installs leave the marketing name empty
}

// ------------------------------------------------------------------------------------------------ engine

struct Engine::WeightImage {
    int device = 0;
    size_t bytes = 0;
    std::vector<void *> allocs;
    float *window = nullptr, *twiddle = nullptr, *mean = nullptr, *scale = nullptr, *b_in = nullptr;
    void *w_in = nullptr;
    StageDev sd[kStages]{};
    int nby[kStages] = {0, 0, 0, 0};
    ~WeightImage() {
        (void) hipSetDevice(device);
        for (void *p : allocs) (void) hipFree(p);
    }
};

namespace {
// content key of a parameter set: 64-bit multiply-xor over every tensor (15 MB: ~2 ms), plus the dimensions
uint64_t params_key(const Params &p) {
    uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t) p.front_taps;
    auto mix = [&](const std::vector<float> &v) {
        const size_t n = v.size();
        h = (h ^ n) * 0xff51afd7ed558ccdull;
        const uint32_t *u = (const uint32_t *) v.data();
        size_t i = 0;
        for (; i + 1 < n; i += 2) {
            const uint64_t w = (uint64_t) u[i] | ((uint64_t) u[i + 1] << 32);
            h = (h ^ w) * 0x9e3779b97f4a7c15ull;
            h ^= h >> 29;
        }
        if (i < n) h = (h ^ u[i]) * 0x9e3779b97f4a7c15ull;
    };
    mix(p.mean), mix(p.scale), mix(p.w_in), mix(p.b_in);
    for (int s = 0; s < kStages; ++s) {
        const Params::Stage &st = p.st[s];
        h = (h ^ (uint64_t) (st.d_in * 1024 + st.d_out)) * 0xff51afd7ed558ccdull;
        mix(st.w_ih_a), mix(st.b_ih_a), mix(st.w_hh_a), mix(st.b_hh_a), mix(st.w_ih_b), mix(st.b_ih_b), mix(st.w_hh_b), mix(st.b_hh_b),
            mix(st.w_head), mix(st.b_head);
    }
    return h;
}
struct ImageKey {
    uint64_t content;
    int device, precision;
    bool operator<(const ImageKey &o) const {
        return content != o.content ? content < o.content : device != o.device ? device < o.device : precision < o.precision;
    }
};
std::mutex g_image_mutex;
std::map<ImageKey, std::weak_ptr<void>> g_images;  // (weak: an image lives as long as a handle holds it)
}  // namespace

size_t Engine::shared_device_bytes() const { return weights_ ? weights_->bytes : 0; }

void *Engine::dalloc(size_t bytes, bool zero) {
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
        (void) hipGetLastError();
        alloc_failed_ = true;
        return nullptr;
    }
    if (alloc_sink_) {
        alloc_sink_->push_back(p);
        weights_->bytes += bytes ? bytes : 16;
    } else {
        allocs_.push_back(p);
        own_bytes_ += bytes ? bytes : 16;
    }
    // (on the handle's own non-blocking stream: a legacy-stream operation would collide with another handle's graph capture)
    if (zero) (void) hipMemsetAsync(p, 0, bytes, own_stream_);
    return p;
}

void *Engine::upload(const void *src, size_t bytes) {
    void *p = dalloc(bytes, false);
    if (p) {
        (void) hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, own_stream_);
        (void) hipStreamSynchronize(own_stream_);  // `src` is usually a temporary
    }
    return p;
}

Engine *Engine::create(const Params &p, int device, int num_streams, int max_frames, int precision, std::string *err,
                       bool *oom) {
    // (the handle is owned from its first byte: an exception out of init() -- std::bad_alloc from the packing vectors -- destroys the
    // stream, events and device allocations made so far and reaches the ABI as "out of memory", not as a leak)
    std::unique_ptr<Engine> e(new Engine());
    *oom = false;
    try {
        if (!e->init(p, device, num_streams, max_frames, precision, err, oom)) return nullptr;
    } catch (const std::bad_alloc &) {
        *oom = true;
        *err = "Failed to allocate host memory.";
        return nullptr;
    }
    return e.release();
}

// Tables and packed weights of one (model, device, precision): runs once per image, under g_image_mutex, with dalloc() recording into
// the image (alloc_sink_); leaves the pointers in this handle's members, from where init() copies them into the image.
bool Engine::build_weights(const Params &p, int precision, std::string *err) {
    // ---- tables
    std::vector<float> win(kNfft), tw(2 * kNfft);
    const double pi = 3.14159265358979323846;
    for (int n = 0; n < kNfft; ++n) {
        win[n] = (float) sin(pi * (double) n / kNfft);
        tw[2 * n] = (float) cos(2.0 * pi * (double) n / kNfft);
        tw[2 * n + 1] = (float) -sin(2.0 * pi * (double) n / kNfft);
    }
    d_window_ = (float *) upload(win.data(), win.size() * 4);
    d_twiddle_ = (float *) upload(tw.data(), tw.size() * 4);
    d_mean_ = (float *) upload(p.mean.data(), kBins * 4);
    d_scale_ = (float *) upload(p.scale.data(), kBins * 4);

    // ---- weights, pre-packed into MFMA B-fragment order
    const int G3 = 3 * kHidden;
    if (!fold_) {
        auto tiles = dense_tiles(kHidden, pi_.npb);
        std::vector<Seg> segs;  // one segment per stacked feature frame, oldest first: each padded to whole k-blocks like the features
        for (int i = 0; i < p.front_taps; ++i) segs.push_back({i * kBins, kBins});
        auto w = pack_b(p.w_in.data(), kHidden, segs, tiles, precision);
        auto b = pack_bias(p.b_in.data(), tiles);
        w_in_ = upload(w.data(), w.size());
        b_in_ = (float *) upload(b.data(), b.size() * 4);
    }
    const auto gt = gru_tiles();
    bool pad_rows_ok = true;  // see rows_zero_from
    for (int s = 0; s < kStages; ++s) {
        const Params::Stage &st = p.st[s];
        StageDev &d = sd_[s];
        std::vector<Seg> segs_a;
        if (st.d_in) segs_a.push_back({0, st.d_in});
        segs_a.push_back({st.d_in, kHidden});
        // bf16 configuration: the gates are evaluated through 2^x, so the constants of
        // sigma(x) = 1 / (1 + 2^(-x log2 e)) and tanh(x) = 1 - 2 / (1 + 2^(2 x log2 e)) are folded into the r / z and n columns of
        // W_ih, W_hh and both biases BEFORE they are rounded to the operand type (one fp32 multiplication each; the oracle's
        // bf16 mode does the same): the recurrent kernels then add and exponentiate, nothing else
        auto gate_scaled = [&](const std::vector<float> &v, size_t rows) {
            std::vector<float> o(v);
            if (precision != kBf16) return o;
            for (size_t r = 0; r < rows; ++r)
                for (int c = 0; c < G3; ++c) o[r * G3 + c] = v[r * G3 + c] * (c < 2 * kHidden ? kGateScaleRZ : kGateScaleN);
            return o;
        };
        auto pk = [&](const std::vector<float> &w, const std::vector<Seg> &segs) {
            auto img = pack_b(gate_scaled(w, w.size() / G3).data(), G3, segs, gt, precision);
            // (the last segment is the hidden-sized one: its padding rows face the constant 1 of a published hidden sequence)
            int nb = 0;
            for (const Seg &sg : segs) nb += ceil_div(sg.klen, pi_.kb);
            pad_rows_ok = pad_rows_ok && rows_zero_from(img, nb, (nb - nbh_) * pi_.kb + kHidden, precision);
            return upload(img.data(), img.size());
        };
        auto pb = [&](const std::vector<float> &b) {
            auto img = pack_bias(gate_scaled(b, 1).data(), gt);
            return (float *) upload(img.data(), img.size() * 4);
        };
        // bf16 configuration: b_hh becomes rows kBiasK0, kBiasK0 + 1 of the packed W_hh -- the scaled bias as a bf16 pair, hi =
        // bf16(b), lo = bf16(b - hi) -- multiplied by the constant 1 every bf16 recurrent kernel keeps at those k of the hidden-state
        // operand (kns_layout.h; the oracle's bf16 mode appends the same two rows)
        auto pk_hh = [&](const std::vector<float> &w, const std::vector<float> &b) {
            if (precision != kBf16) return pk(w, {{0, kHidden}});
            std::vector<float> aug = gate_scaled(w, kHidden), bs = gate_scaled(b, 1);
            aug.resize((size_t) (kHidden + 2) * G3);
            for (int c = 0; c < G3; ++c) {
                const uint16_t hb = to_bf16(bs[c]);
                const uint32_t hu = (uint32_t) hb << 16;
                float hi;
                memcpy(&hi, &hu, 4);
                aug[(size_t) kBiasK0 * G3 + c] = hi;
                aug[(size_t) (kBiasK0 + 1) * G3 + c] = bs[c] - hi;  // (exact in fp32; rounded to bf16 by pack_b)
            }
            auto img = pack_b(aug.data(), G3, {{0, kHidden + 2}}, gt, precision);
            return upload(img.data(), img.size());
        };
        if (fold_) {
            // bf16 configuration, one-frame front-end: the front-end is linear and only the stage-input
            // GEMMs consume its output, so it is FOLDED into them -- with W_e the embedding rows of this stage's gate-scaled, unrounded
            // W_ih:  Wc[k][n] = sum_j w_in[k][j] W_e[j][n],  b'[n] = b_ih[n] + sum_j b_in[j] W_e[j][n]  (j ascending fmaf chains from 0, in
            // fp32), THEN rounded to bf16.  Stage input = [y_prev ; features]; the embedding is never formed.  (oracle: fold_front)
            const std::vector<float> ws = gate_scaled(st.w_ih_a, (size_t) st.d_in + kHidden), bs = gate_scaled(st.b_ih_a, 1);
            // a fed-forward head of at most kYPadMax values rides BEHIND the features and shares their last k-block (stages 1 and 2):
            // rows [features ; y_prev], ONE segment of 257 + d_in <= 288 rows; wider ones stay in front with k-blocks of their own
            d.ypad = st.d_in > 0 && st.d_in <= kYPadMax;
            const int f0 = d.ypad ? 0 : st.d_in, y0 = d.ypad ? kBins : 0;
            std::vector<float> wf((size_t) (st.d_in + kBins) * G3, 0.0f), bf(G3, 0.0f);
            memcpy(wf.data() + (size_t) y0 * G3, ws.data(), sizeof(float) * (size_t) st.d_in * G3);
            const float *we = ws.data() + (size_t) st.d_in * G3;
            for (int k = 0; k < kBins; ++k) fold_row(wf.data() + (size_t) (f0 + k) * G3, p.w_in.data() + (size_t) k * kHidden, we, kHidden, G3);
            for (int j = 0; j < kHidden; ++j)
                for (int n = 0; n < G3; ++n) bf[n] = __builtin_fmaf(p.b_in[j], we[(size_t) j * G3 + n], bf[n]);
            for (int n = 0; n < G3; ++n) bf[n] = bs[n] + bf[n];
            std::vector<Seg> segs_f;
            if (d.ypad) {
                segs_f.push_back({0, kBins + st.d_in});
            } else {
                if (st.d_in) segs_f.push_back({0, st.d_in});
                segs_f.push_back({st.d_in, kBins});
            }
            auto img = pack_b(wf.data(), G3, segs_f, gt, precision);
            {   // columns past [y ; features ; y behind] of the feature operand may hold other stages' head values
                int nb = 0;
                for (const Seg &sg : segs_f) nb += ceil_div(sg.klen, pi_.kb);
                pad_rows_ok = pad_rows_ok && rows_zero_from(img, nb, (nb - nbf_) * pi_.kb + kBins + (d.ypad ? st.d_in : 0), precision);
            }
            d.w_ih_a = upload(img.data(), img.size());
            auto bimg = pack_bias(bf.data(), gt);
            d.b_ih_a = (float *) upload(bimg.data(), bimg.size() * 4);
        } else {
            d.w_ih_a = pk(st.w_ih_a, segs_a);
            d.b_ih_a = pb(st.b_ih_a);
        }
        d.w_hh_a = pk_hh(st.w_hh_a, st.b_hh_a);
        d.w_ih_b = pk(st.w_ih_b, {{0, kHidden}});
        d.w_hh_b = pk_hh(st.w_hh_b, st.b_hh_b);
        d.b_hh_a = pb(st.b_hh_a);
        d.b_ih_b = pb(st.b_ih_b);
        d.b_hh_b = pb(st.b_hh_b);
        auto ht = dense_tiles(st.d_out, s < kStages - 1 ? pi_.npb : 1);
        auto hw = pack_b(st.w_head.data(), st.d_out, {{0, kHidden}}, ht, precision);
        pad_rows_ok = pad_rows_ok && rows_zero_from(hw, nbh_, kHidden, precision);
        auto hb = pack_bias(st.b_head.data(), ht);
        d.w_head = upload(hw.data(), hw.size());
        d.b_head = (float *) upload(hb.data(), hb.size() * 4);
        d.head_tiles = (int) ht.size();
        d.head_dim = st.d_out;
        nby_[s] = ceil_div(st.d_out, pi_.kb);
    }

    if (!pad_rows_ok) {
        *err = "Internal error: a packed weight image has non-zero padding rows.";
        return false;
    }
    return true;

}

bool Engine::init(const Params &p, int device, int B, int Tmax, int precision, std::string *err, bool *oom) {
    if (hipSetDevice(device) != hipSuccess) {
        (void) hipGetLastError();
        *err = "Failed to communicate with device.";
        return false;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        (void) hipGetLastError();
        *err = "Failed to communicate with device.";
        return false;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        *err = std::string("GPU `") + prop.gcnArchName + "` is not supported: this build carries gfx950 (MI355X) code only.";
        return false;
    }
    device_ = device;
    B_ = B;
    Bpad_ = ceil_div(B, 16) * 16;
    Tmax_ = Tmax;
    prec_ = precision;
    pi_ = prec_info(precision);
    nbf_ = ceil_div(kBins, pi_.kb);
    nbh_ = ceil_div(kHidden, pi_.kb);
    taps_ = p.front_taps;
    fold_ = precision == kBf16 && taps_ == 1;  // the front-end rides in the stage-input GEMMs (see the packing below)
    if (hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking) != hipSuccess) {
        (void) hipGetLastError();
        *err = "Failed to create a HIP stream.";
        return false;
    }
    stream_ = own_stream_;
    // developer switches: read once per handle, and only in the -DKNS_DEV build (dev_env() is a constant nullptr otherwise)
    use_graph_ = dev_env("KOALA_AMD_NO_GRAPH") == nullptr;
    no_small_ = dev_env("KOALA_AMD_NO_SMALL") != nullptr;
    no_zero_copy_ = dev_env("KOALA_AMD_NO_ZERO_COPY") != nullptr;
    no_recompute_ = dev_env("KOALA_AMD_STORE_SPECTRUM") != nullptr;  // A/B switch: spectrum through HBM in every call
    debug_taps_ = dev_env("KOALA_AMD_DEBUG_TAPS") != nullptr;        // keep every intermediate debug_read() can return
    dev_variant_ = (dev_env("KOALA_AMD_GRU_STREAM") ? kDevGruStream : 0) | (dev_env("KOALA_AMD_GEMM_GENERIC") ? kDevGemmGeneric : 0) |
                   (dev_env("KOALA_AMD_GEMM_NO_WSR") ? kDevGemmNoWsr : 0);
    auto dev_int = [](const char *name, int dflt) {
        const char *e = dev_env(name);
        return e ? atoi(e) : dflt;
    };
    dev_only_class_ = dev_int("KOALA_AMD_ONLY_CLASS", -1);  // power / clock probing: launch one kernel class only (garbage out)
    dev_analysis_seg_ = dev_int("KOALA_AMD_ANALYSIS_SEG", 0);
    dev_synth_seg_ = dev_int("KOALA_AMD_SYNTH_SEG", 0);
    dev_small_mt_ = dev_int("KOALA_AMD_SMALL_MT", 0);
    dev_steps_mt_ = dev_int("KOALA_AMD_STEPS_MT", 192);
    dev_wave_mt_ = dev_int("KOALA_AMD_WAVE_MT", -1);  // -1: the measured limits; 0: never
    dev_wave_group_ = dev_int("KOALA_AMD_WAVE_GROUP", 0);
    dev_wave_parts_ = dev_int("KOALA_AMD_WAVE_PARTS", 1);  // 0: a layer never takes more than one XCD
    dev_pipe_chunk_ = dev_int("KOALA_AMD_PIPE_CHUNK", 0);  // frames per sub-chunk of the layer pipeline of mid-size batches (0: 16)
    dev_pipe_whole_stft_ = dev_env("KOALA_AMD_PIPE_WHOLE_STFT") != nullptr;  // its analysis / synthesis as whole-call launches (A/B arm)
    dev_pipe_streams_ = dev_int("KOALA_AMD_PIPE_STREAMS", 0);  // streams of the pipeline (0: the default, at most 4)
    dev_pipe_grid_ = dev_int("KOALA_AMD_PIPE_GRID", 0);    // workgroups of its weight-stationary GEMM launches (0: the default)
    dev_pipe_mt_ = dev_int("KOALA_AMD_PIPE_MT", -1);       // up to this many m-tiles (-1: the measured limit; 0: never)
    // One-frame calls of large batches: a GRU layer as ONE launch fused over CU quads (kns_gruq.hip, gru_quad1_kernel), bit-identical
    // to the two-kernel form -- a CU pulls a quarter of W_ih and W_hh (300 KiB) per layer instead of a half of one and all of the
    // other (~740 KiB): 128 against 222 us per 4096-stream frame step.  (The multi-frame form of that decomposition measured slower
    // than input GEMM + recurrent kernel, 354 against 278 us per layer at 64 frames, and is no longer in the tree: DESIGN.md section 6.)
    use_quad_ = dev_env("KOALA_AMD_NO_QUAD") == nullptr;
    fuse_head_ = dev_env("KOALA_AMD_NO_HEAD_FUSE") == nullptr;
    fuse_front_ = dev_env("KOALA_AMD_NO_STFT_FUSE") == nullptr;  // A/B arm: front-end / mask head as launches of their own in one-frame calls  // A/B arm: narrow heads as launches of their own in one-frame calls
    quad_nb0_max_ = dev_int("KOALA_AMD_QUAD_NB0MAX", 2);
    qdbg_block_ = dev_int("KOALA_AMD_QUAD_DBG", -1);
    // host-pointer calls are cut into sub-chunks of host_chunk_ frames (two staging slots = the Tmax-sized buffers)
    host_chunk_ = Tmax_ / 2 < 1 ? 1 : (Tmax_ / 2 > 16 ? 16 : Tmax_ / 2);
    host_pipeline_min_bytes_ = (size_t) 4 << 20;  // below this a call is launch-bound: sub-chunks would only add launches
    if (const char *e = dev_env("KOALA_AMD_HOST_SCHED")) {  // sub-chunk lengths "6,8,12,...": repeated / truncated to the call's frames
        for (const char *q = e; *q;) {
            dev_host_sched_.push_back(atoi(q));
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
    }
    if (const char *e = dev_env("KOALA_AMD_HOST_CHUNK")) {  // force a sub-chunk length; 0 = never split
        const int v = atoi(e);
        if (v >= 1 && v <= Tmax_ / 2) host_chunk_ = v, host_pipeline_min_bytes_ = 0;
        if (v == 0) host_chunk_ = Tmax_;
    }
    bool ok_sync = hipStreamCreateWithFlags(&copy_in_, hipStreamNonBlocking) == hipSuccess &&
                   hipStreamCreateWithFlags(&copy_out_, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 2 && ok_sync; ++i)
        ok_sync = hipEventCreateWithFlags(&ev_in_[i], hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&ev_done_[i], hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&ev_out_[i], hipEventDisableTiming) == hipSuccess;
    if (!ok_sync) {
        (void) hipGetLastError();
        *err = "Failed to create HIP streams/events.";
        return false;
    }

    // ---- tables and weights: shared with every other handle open on the same model, device and precision
    {
        std::lock_guard<std::mutex> lock(g_image_mutex);
        const ImageKey key{params_key(p), device, precision};
        std::shared_ptr<void> held = g_images[key].lock();
        // (KOALA_AMD_NO_WEIGHT_CACHE, developer build: every handle builds its own image, as in rounds 1-5)
        if (held && !dev_env("KOALA_AMD_NO_WEIGHT_CACHE")) {
            weights_ = std::static_pointer_cast<WeightImage>(held);
            weights_cached_ = true;
        } else {
            weights_ = std::make_shared<WeightImage>();
            weights_->device = device;
            alloc_sink_ = &weights_->allocs;
            const bool built = build_weights(p, precision, err);
            alloc_sink_ = nullptr;
            if (!built) return false;
            if (alloc_failed_) {
                *oom = true;
                *err = "Failed to allocate device memory.";
                return false;
            }
            WeightImage &w = *weights_;
            w.window = d_window_, w.twiddle = d_twiddle_, w.mean = d_mean_, w.scale = d_scale_, w.w_in = w_in_, w.b_in = b_in_;
            for (int s = 0; s < kStages; ++s) w.sd[s] = sd_[s], w.nby[s] = nby_[s];
            for (auto it = g_images.begin(); it != g_images.end();)  // (entries whose last handle is gone)
                it = it->second.expired() ? g_images.erase(it) : std::next(it);
            g_images[key] = std::static_pointer_cast<void>(weights_);
        }
        const WeightImage &w = *weights_;
        d_window_ = w.window, d_twiddle_ = w.twiddle, d_mean_ = w.mean, d_scale_ = w.scale, w_in_ = w.w_in, b_in_ = w.b_in;
        for (int s = 0; s < kStages; ++s) sd_[s] = w.sd[s], nby_[s] = w.nby[s];
    }

    // ---- per-stream state
    const size_t mtb = (size_t) Bpad_ / 16;
    d_hist_[0] = (int16_t *) dalloc((size_t) Bpad_ * kFrame * 2, true);
    d_hist_[1] = (int16_t *) dalloc((size_t) Bpad_ * kFrame * 2, true);
    d_tail_[0] = (float *) dalloc((size_t) Bpad_ * kFrame * 4, true);
    d_tail_[1] = (float *) dalloc((size_t) Bpad_ * kFrame * 4, true);
    d_hstate_[0] = (float *) dalloc((size_t) kGruLayers * mtb * kUnitTiles * 1024, true);
    d_hstate_[1] = (float *) dalloc((size_t) kGruLayers * mtb * kUnitTiles * 1024, true);
    d_hprev_ = dalloc((size_t) kGruLayers * mtb * nbh_ * 1024, true);  // the state in operand form (wavefront calls, frame 0)
    d_rmask_ = (uint8_t *) dalloc((size_t) Bpad_, true);

    // ---- activation workspace
    const size_t M = mtb * (size_t) Tmax_;  // m-tiles per call
    d_spec_ = (float *) dalloc((size_t) Tmax_ * Bpad_ * 256 * 8, true);
    feat_frame_bytes_ = mtb * nbf_ * 1024;
    d_feat_ = dalloc((M + (size_t) (taps_ - 1) * mtb) * nbf_ * 1024, true);  // [context frames | the call's frames]
    if (taps_ > 1) {
        // taps_ slots: slots 1 .. taps_ - 1 hold the features of the last taps_ - 1 frames; slot 0 is where a one-frame call's roll
        // puts the oldest tap (kns_stft.hip, AnalysisArgs::feat_hist)
        d_fhist_ = dalloc((size_t) taps_ * feat_frame_bytes_, true);
        // one m-tile whose 16 rows are the feature of a silent frame, in the operand type and layout of `feat`: made by the
        // analysis kernel itself from 16 streams of zeros, so that "reset" and "silent frames" are the same bits in both
        // configurations (the bf16 one takes its logarithm from v_log_f32, which no host formula reproduces)
        d_silent_ = dalloc((size_t) nbf_ * 1024, true);
    }
    if (!fold_) d_e_ = dalloc(M * nbh_ * 1024, true);
    for (int s = 0; s < kStages - 1; ++s) d_y_[s] = dalloc(M * nby_[s] * 1024, true);
    d_gi_ = dalloc(M * kGateTiles * 64 * pi_.gisz, true);
    d_hseq_a_ = dalloc(M * nbh_ * 1024, true);
    d_hseq_b_ = dalloc(M * nbh_ * 1024, true);
    d_mask_ = (float *) dalloc(M * kMaskTiles * 1024, true);
    if (!gru_quad_supported(prec_, (int) mtb, 0)) use_quad_ = false;
    if (qdbg_block_ >= 0) d_qdbg_ = (unsigned long long *) dalloc((size_t) 8 * 4 * 8 * 8, true);
    d_in_ = (int16_t *) dalloc((size_t) B_ * Tmax_ * kFrame * 2, true);
    d_out_ = (int16_t *) dalloc((size_t) B_ * Tmax_ * kFrame * 2, true);
    if (alloc_failed_) {
        *oom = true;
        *err = "Failed to allocate device memory.";
        return false;
    }
    if (hipHostMalloc((void **) &h_in_, (size_t) B_ * Tmax_ * kFrame * 2, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **) &h_out_, (size_t) B_ * Tmax_ * kFrame * 2, hipHostMallocDefault) != hipSuccess) {
        (void) hipGetLastError();
        *oom = true;
        *err = "Failed to allocate pinned host memory.";
        return false;
    }
    d_frame_count_ = (unsigned *) dalloc(64, true);
    if (hipHostMalloc((void **) &h_frame_word_, 64, hipHostMallocDefault) != hipSuccess) {
        (void) hipGetLastError();
        h_frame_word_ = nullptr;  // (no completion word: one-frame replays wait in hipStreamSynchronize)
    } else {
        *h_frame_word_ = 0;
    }
    // KOALA_AMD_WAIT=block (product option): one-frame host calls sleep in hipStreamSynchronize instead of polling the frame's
    // completion word -- for hosts that run more streaming handles than they have cores
    {
        const char *w = getenv("KOALA_AMD_WAIT");
        spin_wait_ = dev_env("KOALA_AMD_NO_SPIN_WAIT") == nullptr && !(w && !strcmp(w, "block"));
    }
    if (taps_ > 1) {  // the front-end context of a fresh stream is silence, not zeros
        AnalysisArgs an;
        an.pcm = d_in_;  // zeros (at least one frame of B_ >= 1 streams; rows past the last stream re-read the last one)
        an.hist_in = d_hist_[0];
        an.hist_out = d_hist_[0];  // (zeros over zeros)
        an.window = d_window_;
        an.twiddle = d_twiddle_;
        an.mean = d_mean_;
        an.scale = d_scale_;
        an.spec = nullptr;
        an.feat = d_silent_;
        an.B = 1;
        an.Bpad = 16;
        an.T = 1;
        an.nbf = nbf_;
        an.precision = prec_;
        an.seg = 1;
        an.write_spec = 0;
        launch_analysis(an, own_stream_);
        std::string e2;
        if (!reset(nullptr, &e2)) {
            *err = e2;
            return false;
        }
    }
    if (hipStreamSynchronize(own_stream_) != hipSuccess) {
        *err = std::string("Device initialisation failed: ") + hipGetErrorString(hipGetLastError());
        return false;
    }
    return true;
}

Engine::~Engine() {
    if (own_stream_) (void) hipStreamSynchronize(own_stream_);
    if (copy_out_) (void) hipStreamSynchronize(copy_out_);  // (asynchronous host calls still writing into the caller's buffers)
    for (Span &s : spans_) {
        (void) hipEventDestroy(s.a);
        (void) hipEventDestroy(s.b);
    }
    for (hipEvent_t e : pool_) (void) hipEventDestroy(e);
    for (hipGraphExec_t ge : frame_graph_)
        if (ge) (void) hipGraphExecDestroy(ge);
    for (void *p : allocs_) (void) hipFree(p);
    if (h_frame_word_) (void) hipHostFree(h_frame_word_);
    if (h_in_) (void) hipHostFree(h_in_);
    if (h_out_) (void) hipHostFree(h_out_);
    for (int i = 0; i < 4; ++i)
        if (aev_out_[i]) (void) hipEventDestroy(aev_out_[i]);
    for (int i = 0; i < 2; ++i) {
        if (aev_in_[i]) (void) hipEventDestroy(aev_in_[i]);
        if (aev_done_[i]) (void) hipEventDestroy(aev_done_[i]);
        if (ev_in_[i]) (void) hipEventDestroy(ev_in_[i]);
        if (ev_done_[i]) (void) hipEventDestroy(ev_done_[i]);
        if (ev_out_[i]) (void) hipEventDestroy(ev_out_[i]);
    }
    for (int i = 0; i < kPipeStreams; ++i) {
        if (pipe_stream_[i]) (void) hipStreamSynchronize(pipe_stream_[i]), (void) hipStreamDestroy(pipe_stream_[i]);
        if (pipe_join_[i]) (void) hipEventDestroy(pipe_join_[i]);
    }
    if (pipe_fork_) (void) hipEventDestroy(pipe_fork_);
    for (int i = 0; i < kPipeRing; ++i)
        if (pipe_syn_[i]) (void) hipEventDestroy(pipe_syn_[i]);
    for (int i = 0; i < kPipeRing; ++i)
        for (int l = 0; l < kGruLayers; ++l)
            if (pipe_ev_[i][l]) (void) hipEventDestroy(pipe_ev_[i][l]);
    if (host_fork_) (void) hipEventDestroy(host_fork_);
    if (copy_in_) (void) hipStreamDestroy(copy_in_);
    if (copy_out_) (void) hipStreamDestroy(copy_out_);
    if (own_stream_) (void) hipStreamDestroy(own_stream_);
}

void Engine::tick(int) {
    if (!profiling_) return;
    hipEvent_t e;
    if (pool_.empty()) {
        (void) hipEventCreate(&e);
    } else {
        e = pool_.back();
        pool_.pop_back();
    }
    (void) hipEventRecord(e, stream_);
    pending_ = e;
}

void Engine::tock(int cls) {
    if (!profiling_) return;
    hipEvent_t e;
    if (pool_.empty()) {
        (void) hipEventCreate(&e);
    } else {
        e = pool_.back();
        pool_.pop_back();
    }
    (void) hipEventRecord(e, stream_);
    spans_.push_back({cls, pending_, e});
}

void Engine::profile_enable(bool on) {
    profiling_ = on;
    if (on) {
        for (int i = 0; i < kNumKernelClasses; ++i) {
            acc_ms_[i] = 0;
            acc_n_[i] = 0;
        }
    }
}

bool Engine::profile_read(double *ms, int64_t *launches, std::string *err) {
    if (hipStreamSynchronize(stream_) != hipSuccess) {
        *err = std::string("HIP error: ") + hipGetErrorString(hipGetLastError());
        return false;
    }
    for (Span &s : spans_) {
        float t = 0;
        (void) hipEventElapsedTime(&t, s.a, s.b);
        acc_ms_[s.cls] += t;
        acc_n_[s.cls] += 1;
        pool_.push_back(s.a);
        pool_.push_back(s.b);
    }
    spans_.clear();
    for (int i = 0; i < kNumKernelClasses; ++i) {
        ms[i] = acc_ms_[i];
        launches[i] = acc_n_[i];
    }
    return true;
}

bool Engine::synchronize(std::string *err) {
    if (async_n_ && !drain_async(err)) return false;
    if (hipStreamSynchronize(stream_) != hipSuccess) {
        *err = std::string("HIP error: ") + hipGetErrorString(hipGetLastError());
        return false;
    }
    return true;
}

bool Engine::reset(const uint8_t *host_mask, std::string *err) {
    (void) hipSetDevice(device_);
    // (the reset kernel is ordered behind every enqueued call on the handle's stream; nothing to drain)
    ResetArgs r;
    r.hist = d_hist_[0];
    r.hist2 = d_hist_[1];
    r.tail = d_tail_[0];
    r.tail2 = d_tail_[1];
    r.hstate = d_hstate_[0];
    r.hstate2 = d_hstate_[1];
    r.Bpad = Bpad_;
    r.fhist = d_fhist_;
    r.silent = d_silent_;
    r.fhist_frames = taps_ > 1 ? taps_ : 0;
    r.nbf = nbf_;
    r.mask = nullptr;
    if (host_mask) {
        std::vector<uint8_t> m((size_t) Bpad_, 0);
        memcpy(m.data(), host_mask, (size_t) B_);
        hipError_t ce = hipMemcpyAsync(d_rmask_, m.data(), (size_t) Bpad_, hipMemcpyHostToDevice, stream_);
        if (ce == hipSuccess) ce = hipStreamSynchronize(stream_);  // `m` goes out of scope
        if (ce != hipSuccess) {  // a stale mask would reset the wrong streams: report, do not launch
            (void) hipGetLastError();
            *err = std::string("HIP error: ") + hipGetErrorString(ce);
            return false;
        }
        r.mask = d_rmask_;
    }
    launch_reset(r, stream_);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        *err = std::string("HIP error: ") + hipGetErrorString(e);
        return false;
    }
    return true;
}

// the arguments of one GRU layer of ONE frame (frame t of a call) for the low-latency layer kernel and the wavefront's items
GruSmallArgs Engine::small_args(int mtb, const void *a0, int nb0, const void *a1, const void *wih, const float *bih, const void *whh,
                                const float *bhh, int layer, void *hseq, int t, const StageDev *head) const {
    GruSmallArgs g;
    const size_t frame = (size_t) t * mtb * 1024;  // bytes of one k-block column of A per frame
    if (head) {  // the narrow head of the stage before, inside this launch
        g.yh = (const char *) d_hseq_b_ + frame * nbh_;
        g.yw = head->w_head;
        g.yb = head->b_head;
        g.yvalid = head->head_dim;
    }
    g.a0 = a0 ? (const char *) a0 + frame * nb0 : nullptr;
    g.a1 = (const char *) a1 + frame * nbh_;
    g.wih = wih;
    g.bih = bih;
    g.whh = whh;
    g.bhh = bhh;
    g.hstate_in = d_hstate_[(hs_cur_ + t) & 1] + (size_t) layer * mtb * kUnitTiles * 256;
    g.hstate_out = d_hstate_[(hs_cur_ + t + 1) & 1] + (size_t) layer * mtb * kUnitTiles * 256;
    g.hseq = (char *) hseq + frame * nbh_;
    g.nb0 = nb0;
    g.mtiles = mtb;
    g.precision = prec_;
    return g;
}

// ---- calls of several frames as a WAVEFRONT over (pipeline stage, frame): the stages are the eight GRU layers with the three narrow
// heads between them; stage i of frame t needs stage i - 1 of frame t and (a layer) its own frame t - 1, so launch k runs the items with
// i + t = k side by side (kns_gru.hip, gru_wave_kernel) -- T + 10 launches of up to 8 x 17 x groups workgroups instead of 8 T launches of
// 17 x mtb (kRouteSmallSteps) or of mtb (the chunked recurrent kernels).  The per-frame slots of the two hidden-sequence buffers, of the
// y operands and of the features' padding are reused stage after stage exactly as in the layer-by-layer order: within a frame the stages
// still run one after the other.
GruWaveItem Engine::wave_item(int i, int t, int mtb) const {  // stage i (layers 3 s, 3 s + 1; head 3 s + 2) of frame t
    const int s = i / 3, r = i % 3;
    const StageDev &d = sd_[s];
    const size_t frame = (size_t) t * mtb * 1024;
    char *feat_call = (char *) d_feat_ + (size_t) (taps_ - 1) * feat_frame_bytes_;
    GruWaveItem it;
    if (r == 2) {  // the head of stage s
        it.g.yh = (const char *) d_hseq_b_ + frame * nbh_;
        it.g.yw = d.w_head;
        it.g.yb = d.b_head;
        it.g.yvalid = d.head_dim;
        it.g.mtiles = mtb;
        it.g.precision = prec_;
        if (sd_[s + 1].ypad) {  // into columns 257 ... of the features (k = 1 ... of their block 8)
            it.chains = 1;
            it.pad = 1;
            it.yout = feat_call + frame * nbf_;
            it.y_nb = nbf_;
            it.y_blk = kBins / pi_.kb;
            it.y_kk0 = kBins % pi_.kb;
        } else {
            it.chains = d.head_tiles;
            it.yout = (char *) d_y_[s] + frame * nby_[s];
            it.y_nb = nby_[s];
        }
        return it;
    }
    // h_{t-1} in operand form: the layer's hidden-sequence slot of the frame before, or the converted state (d_hprev_) for frame 0
    const void *hseq = r ? d_hseq_b_ : d_hseq_a_;
    it.hprev = t ? (const char *) hseq + (frame - (size_t) mtb * 1024) * nbh_ : (const char *) d_hprev_ + (size_t) (2 * s + r) * mtb * nbh_ * 1024;
    if (r == 1) {
        it.g = small_args(mtb, nullptr, 0, d_hseq_a_, d.w_ih_b, d.b_ih_b, d.w_hh_b, d.b_hh_b, 2 * s + 1, d_hseq_b_, t, nullptr);
    } else {
        const bool y = s && !d.ypad;
        it.g = small_args(mtb, y ? d_y_[s - 1] : nullptr, y ? nby_[s - 1] : 0, fold_ ? (const void *) feat_call : (const void *) d_e_, d.w_ih_a,
                          d.b_ih_a, d.w_hh_a, d.b_hh_a, 2 * s, d_hseq_a_, t, nullptr);
    }
    return it;
}

bool Engine::wave_fits() const {  // every head either goes into the features' padding or is a y operand of 1-3 k-blocks
    for (int s = 0; s < kStages - 1; ++s)
        if (!(sd_[s + 1].ypad ? sd_[s].head_dim <= 16 : (nby_[s] >= 1 && nby_[s] <= 3 && sd_[s].head_tiles == pi_.npb * nby_[s]))) return false;
    return true;
}

// streams and events of the layer pipeline (run_device), made on first use
bool Engine::pipe_ready() {
    if (pipe_ok_) return true;
    if (pipe_failed_) return false;
    bool ok = true;
    for (int i = 0; i < kPipeStreams && ok; ++i) {
        if (!pipe_stream_[i]) ok = hipStreamCreateWithFlags(&pipe_stream_[i], hipStreamNonBlocking) == hipSuccess;
        if (ok && !pipe_join_[i]) ok = hipEventCreateWithFlags(&pipe_join_[i], hipEventDisableTiming) == hipSuccess;
    }
    if (ok && !pipe_fork_) ok = hipEventCreateWithFlags(&pipe_fork_, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < kPipeRing && ok; ++i)
        for (int l = 0; l < kGruLayers && ok; ++l)
            if (!pipe_ev_[i][l]) ok = hipEventCreateWithFlags(&pipe_ev_[i][l], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < kPipeRing && ok; ++i)
        if (!pipe_syn_[i]) ok = hipEventCreateWithFlags(&pipe_syn_[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void) hipGetLastError();
        pipe_failed_ = true;  // (the plain chunked route stays)
        return false;
    }
    pipe_ok_ = true;
    return true;
}

void Engine::run_wave(int T, int mtb) {
    // m-tiles per workgroup (an XCD holds 64 workgroups at a time, a layer has 17 per group; measured best: 1 up to 64 streams,
    // 3 up to 256 (fp32: 512), then 6 in bf16, 4 up to 2 048 and 8 beyond in fp32 -- short groups balance the CUs, long ones pull the
    // weights less often; within ~3 % of the best of {1, 2, 3, 4, 6, 8} at every size swept, profiles/r04_wavefront.txt)
    int mgroup = mtb <= 4 ? 1 : prec_ == kBf16 ? (mtb <= 16 ? 3 : 6) : (mtb <= 32 ? 3 : mtb <= 128 ? 4 : 8);
    if (dev_wave_group_ > 0) mgroup = dev_wave_group_;
    tick(kClsGru);
    launch_gru_wave_prev(d_hstate_[hs_cur_], d_hprev_, kGruLayers * mtb, prec_, stream_);
    tock(kClsGru);
    for (int k = 0; k < T + kWaveItems - 1; ++k) {
        GruWaveArgs w;
        w.mgroup = mgroup;
#ifdef KNS_TIMING
        w.stamp = k == T / 2 + 5;
#endif
        w.layer_wgs = kUnitTiles * ((mtb + mgroup - 1) / mgroup);
        for (int x = 0; x < 8; ++x) w.layer_item[x] = w.head_item[x] = -1, w.layer_part[x] = 0;
        int n = 0, layers = 0;
        const int i0 = k < T ? 0 : k - T + 1;
        for (int i = i0; i <= k && i < kWaveItems; ++i) layers += i % 3 != 2;
        // one layer per XCD; four or fewer layers (the pipeline filling or draining) take 2, 4 or 8 XCDs each
        w.parts = dev_wave_parts_ ? (layers <= 1 ? 8 : layers <= 2 ? 4 : layers <= 4 ? 2 : 1) : 1;
        w.xcd_wgs = (w.layer_wgs + w.parts - 1) / w.parts;
        int seen = 0;
        for (int i = i0; i <= k && i < kWaveItems; ++i) {
            w.item[n] = wave_item(i, k - i, mtb);
            if (i % 3 == 2) {  // the head of stage s: beside a layer's workgroups, on the XCD(s) of the layer before it in this launch
                w.head_item[w.parts == 1 ? 2 * (i / 3) + 1 : (seen ? seen - 1 : 0) * w.parts] = n;
            } else {
                for (int q = 0; q < w.parts; ++q) {
                    const int x = w.parts == 1 ? 2 * (i / 3) + i % 3 : seen * w.parts + q;  // (all eight in flight: layer l on XCD l)
                    w.layer_item[x] = n;
                    w.layer_part[x] = q;
                }
                ++seen;
            }
            ++n;
        }
        tick(kClsGru);
        launch_gru_wave(w, prec_, mtb, stream_);
        tock(kClsGru);
    }
}

// ---- dispatch: which kernel family a call takes (mtb = m-tiles of 16 streams = ceil(B / 16); edges measured on MI355X, each
// tested from both sides by tests/test_gpu_parity.py::test_dispatch_boundaries; the developer build reports the route of the
// last call through debug tap 6, tests/test_gpu_parity.py::test_dispatch_routes)
//
//   GRU layers (8 per call)
//   | configuration | frames T | m-tiles mtb                           | route                                                        |
//   |---------------|----------|---------------------------------------|--------------------------------------------------------------|
//   | bf16          | 1        | <= 59, or not whole quads up to 192   | kRouteSmall: gru_small_kernel, one launch per layer           |
//   | bf16          | 1        | >= 60 and mtb % 4 == 0                | kRouteQuad1: gru_quad1_kernel, one launch per layer; stage    |
//   |               |          |                                       |   inputs wider than 2 k-blocks: input GEMM + recurrent kernel |
//   | bf16          | 1        | > 192 and mtb % 4 != 0                | kRouteChunked                                                 |
//   | bf16          | > 1      | <= 48 (<= 64 up to T = 4)             | kRouteWave: gru_wave_kernel, T + 10 launches: the (stage,     |
//   |               |          |                                       |   frame) items of one anti-diagonal side by side, one layer   |
//   |               |          |                                       |   per XCD, narrow heads as items of their own                 |
//   | bf16          | >= 32    | 49 .. 144                             | kRoutePipelined: the chunked kernels on TWO sub-chunks of     |
//   |               |          |                                       |   frames, the (layer, sub-chunk) grid as a wavefront over two |
//   |               |          |                                       |   streams; >= 48 frames and <= 112 m-tiles: analysis and      |
//   |               |          |                                       |   synthesis per sub-chunk too                                 |
//   | bf16          | > 1      | otherwise                             | kRouteChunked: gemm_ws2 (m-tiles in multiples of 256 x stage; |
//   |               |          |                                       |   the rest through gemm_kernel) + gru_resident8_kernel        |
//   | fp32          | 1        | <= 256                                | kRouteSmall                                                   |
//   | fp32          | 1        | > 256                                 | kRouteChunked: gemm_kernel + gru_kernel<PF32, 8>              |
//   | fp32          | > 1      | <= 256                                | kRouteWave                                                   |
//   | fp32          | > 1      | <= 192, wavefront off (developer      | kRouteSmallSteps: gru_small_kernel frame by frame (T launches |
//   |               |          |   switch, debug taps)                 |                                                              |
//   |               |          |                                       |   per layer; the chunked recurrence would occupy mtb CUs)     |
//   | fp32          | > 1      | > 192                                 | kRouteChunked                                                 |
//
//   Around them (all configurations)
//   | what                          | T = 1, bf16                                         | otherwise                            |
//   |-------------------------------|-----------------------------------------------------|--------------------------------------|
//   | spectrum                      | stored by analysis, read by synthesis (in place)    | T > 1: rebuilt from the PCM by the   |
//   |                               |                                                     | synthesis kernel (not if in == out)  |
//   | front-end GEMM                | none: folded into the stage-input GEMMs (one-frame  | bf16, T > 1: none either; fp32 and   |
//   |                               | front-end); five-frame: gemm_front5_t1_kernel       | KNS-v1.1: gemm_wsr / front5 / generic |
//   | narrow heads 1 / 5 / 40       | inside the next stage's first layer launch          | wavefront: items of their own; bf16  |
//   |                               |                                                     | chunked: 1 and 5 inside their        |
//   |                               |                                                     | layer-B recurrent launch; gemm_head  |
//   | mask head                     | inside the synthesis launch                         | gemm_wsr_kernel                      |
//   | analysis / synthesis segments | one                                                 | ~4 / ~2 workgroups per CU (>= 1 frame)  |
//   Host-pointer calls: >= 4 MiB and more than min(16, max_frames / 2) frames -> sub-chunks on three streams; T = 1 -> hipGraph replay.
enum Route { kRouteChunked = 0, kRouteSmall = 1, kRouteSmallSteps = 2, kRouteQuad1 = 3, kRouteWave = 4, kRoutePipelined = 5 };

bool Engine::run_device(int T, const int16_t *d_pcm, int16_t *d_out, std::string *err, bool allow_recompute) {
    const int mtb = Bpad_ / 16;
    last_T_ = T;

    AnalysisArgs an;
    an.pcm = d_pcm;
    // T == 1: the one workgroup that reads a stream's history also writes it (same lanes, same addresses), so the state
    // is updated in place and the launch arguments never change -- which is what lets the frame be a hipGraph
    const bool in_place = T == 1;
    an.hist_in = d_hist_[hist_cur_];
    an.hist_out = d_hist_[in_place ? hist_cur_ : hist_cur_ ^ 1];
    an.window = d_window_;
    an.twiddle = d_twiddle_;
    an.mean = d_mean_;
    an.scale = d_scale_;
    an.spec = d_spec_;
    char *feat_now = (char *) d_feat_ + (size_t) (taps_ - 1) * feat_frame_bytes_;  // behind the context frames
    an.feat = feat_now;
    char *fhist_last = (char *) d_fhist_ + feat_frame_bytes_;  // the last taps_ - 1 frames' features
    // one-frame calls of a several-frame front-end: the analysis kernel rolls the history itself and writes the frame's features
    // into its last slot; the front-end reads its taps from there (no copy launches)
    const bool roll_in_analysis = taps_ > 1 && T == 1 && fuse_front_ && !debug_taps_;
    if (roll_in_analysis) {
        an.feat_hist = d_fhist_;
        an.hist_slots = taps_ - 1;
        an.feat = (char *) d_fhist_ + (size_t) (taps_ - 1) * feat_frame_bytes_;
    } else if (taps_ > 1) {  // [features of the last taps - 1 frames | this call's]: the front-end reads taps shifted views of it
        (void) hipMemcpyAsync(d_feat_, fhist_last, (size_t) (taps_ - 1) * feat_frame_bytes_, hipMemcpyDeviceToDevice, stream_);
    }
    an.B = B_;
    an.Bpad = Bpad_;
    an.T = T;
    an.nbf = nbf_;
    an.precision = prec_;
    // The spectrum makes its round trip through HBM only where it has to: in single-frame calls (the history is updated in
    // place there, so the synthesis kernel cannot rebuild it) and when the debug taps are on.  Otherwise the synthesis
    // kernel recomputes it from the PCM.
    // (a caller whose `enhanced` overlaps its `pcm` gets the stored-spectrum form: the synthesis kernel would otherwise
    // read input frames it has already overwritten)
    const bool recompute = !in_place && !no_recompute_ && allow_recompute;
    an.write_spec = !recompute || debug_taps_;
    spec_valid_ = an.write_spec != 0;
    {   // time segments: about four workgroups per CU
        const int seg_env = dev_analysis_seg_;
        int seg = T;
        // (down to one frame per workgroup for few streams: 16 streams x 32 frames, whole call, 0.216 -> 0.192 ms)
        while (seg > 1 && (Bpad_ / 16) * ((T + seg - 1) / seg) < 1024) seg = (seg + 1) / 2;
        an.seg = seg_env > 0 ? (seg_env < T ? seg_env : T) : seg;
    }
    feat_valid_ = !roll_in_analysis;  // (otherwise the features went to the history slots)
    const int16_t *hist_before = d_hist_[hist_cur_];
    const int only = dev_only_class_;  // -1 in the product library
    // (the analysis launch itself follows the route decisions below: the layer pipeline of mid-size batches cuts it into slices of frames)

    // Time slice of the call the launches below work on, and where they go: the whole call on the handle's stream -- or, for mid-size
    // batches, one of its sub-chunks on one of the pipeline streams (run_pipelined below).  Every activation matrix is frame-major
    // ([frame][m-tile][blocks]), so a slice of frames is a contiguous slice of each of them.
    int c_t0 = 0, c_T = T, c_hs = hs_cur_, c_grid = 0;
    hipStream_t c_stream = stream_;
    auto gemm = [&](int cls, const void *a0, int nb0, const void *a1, int nb1, const void *w, const float *bias,
                    void *out, int ntiles, int n_valid, int kind, int taps = 1, int pad_nb = 0, int pad_blk = 0, int pad_kk0 = 0) {
        GemmArgs g;
        g.pad_nb = pad_nb;
        g.pad_blk = pad_blk;
        g.pad_kk0 = pad_kk0;
        g.taps = taps;
        g.tap_stride = feat_frame_bytes_;
        const size_t m0 = (size_t) c_t0 * mtb;  // first m-tile of the slice
        const size_t out_stride = kind == kOutGi ? (size_t) kGateTiles * 64 * pi_.gisz
                                  : kind == kOutMask ? (size_t) kMaskTiles * 256 * (prec_ == kBf16 ? 2 : 4)
                                                     : (size_t) (pad_nb ? pad_nb : (ntiles + pi_.npb - 1) / pi_.npb) * 1024;
        g.a0 = a0 ? (const char *) a0 + m0 * nb0 * 1024 : nullptr;
        g.a1 = (const char *) a1 + m0 * nb1 * 1024;
        g.w = w;
        g.bias = bias;
        g.out = (char *) out + m0 * out_stride;
        g.nb0 = nb0;
        g.nb1 = nb1;
        g.mtiles = mtb * c_T;
        g.ntiles = ntiles;
        g.n_valid = n_valid;
        g.out_kind = kind;
        g.precision = prec_;
        g.dev = dev_variant_;
        g.grid = c_grid;
        tick(cls);
        if (only < 0 || only == cls) launch_gemm(g, c_stream);
        tock(cls);
    };
    auto gru = [&](const void *whh, const float *bhh, int layer, void *hseq, const StageDev *head = nullptr) {
        GruArgs g;
        const size_t m0 = (size_t) c_t0 * mtb;
        if (head) {  // this stage's narrow head inside the launch, into the padding of the feature matrix (kns_kernels.h)
            g.yw = head->w_head;
            g.yb = head->b_head;
            g.yout = feat_now + (size_t) c_t0 * feat_frame_bytes_;
            g.yvalid = head->head_dim;
            g.y_nb = nbf_;
            g.y_blk = kBins / pi_.kb;
            g.y_kk0 = kBins % pi_.kb;
        }
        g.gi = (const char *) d_gi_ + m0 * kGateTiles * 64 * pi_.gisz;
        g.whh = whh;
        g.bhh = bhh;
        g.hstate_in = d_hstate_[c_hs] + (size_t) layer * mtb * kUnitTiles * 256;
        g.hstate_out = d_hstate_[c_hs ^ 1] + (size_t) layer * mtb * kUnitTiles * 256;
        g.hseq = (char *) hseq + m0 * nbh_ * 1024;
        g.T = c_T;
        g.mtiles = mtb;
        g.precision = prec_;
        g.dev = dev_variant_;
        tick(kClsGru);
        if (only < 0 || only == kClsGru) launch_gru(g, c_stream);
        tock(kClsGru);
    };

    // One frame per call: whole GRU layers (input GEMM + recurrent GEMM + gates) as single wide launches of the low-latency
    // kernel -- 17 x mtb workgroups that stream their share of both weight matrices from L2, no pre-activation round trip,
    // half the launches.  The weight-resident kernels first pull 459 KiB per CU for ONE step, so the fused form wins up to
    // 192 m-tiles in bf16 (3 072 streams: 218 vs 264 us per frame step; 512 streams: 113 vs 212; at 4 096 it loses, 256 vs
    // 222) and at every size measured in fp32 (4 096 streams: 628 vs 815 us).  Same arithmetic, bit for bit.
    const int small_env = dev_small_mt_;
    // (bf16, m-tiles in whole quads: from 60 m-tiles on the one-step quad kernel is faster -- it takes ~94 us per frame step from
    // 512 to 1 536 streams, 96 at 2 048, 106 at 4 096 (222 with the chunked kernels); the low-latency kernel below 79 us at 512
    // streams, 81 at 704, 99 at 1 024, 120 at 1 536)
    const int small_mt = small_env > 0 ? small_env : (prec_ == kBf16 ? (use_quad_ ? 59 : 192) : 256);
    const bool small = T == 1 && mtb <= small_mt && !no_small_;
    // Fewer than 256 m-tiles: the chunked recurrent kernels would occupy mtb workgroups, so the layers run frame
    // by frame through the low-latency kernel instead (17 x mtb workgroups of three waves per frame, input GEMM included):
    // in fp32 at 256 streams x 32 frames 8 x 32 launches of ~13 us beat 8 x (0.1 + 0.75) ms (2.8 vs 1.2 M frames/s; 5.5 vs
    // 3.7 M at 1024 streams, 7.2 vs 6.7 M at 3072, equal at 4096).  Frame t of a call reads the hidden
    // state from ping-pong buffer (hs_cur_ + t) & 1 and writes the other one.
    const int steps_mt = dev_steps_mt_;
    // (fp32 only: the bf16 recurrent kernel keeps its weights on chip and is faster than 8 us per frame and layer even
    // with 16 workgroups -- 11.3 vs 5.2 M frames/s at 256 streams; the fp32 one streams them: 1.2 vs 2.8 M)
    auto gru_small = [&](const void *a0, int nb0, const void *a1, const void *wih, const float *bih, const void *whh,
                         const float *bhh, int layer, void *hseq, int t = 0, const StageDev *head = nullptr) {
        const GruSmallArgs g = small_args(mtb, a0, nb0, a1, wih, bih, whh, bhh, layer, hseq, t, head);
        tick(kClsGru);
        launch_gru_small(g, stream_);
        tock(kClsGru);
    };
    // Where it wins (measured against the other routes, tools/wave_check.py, profiles/r04_wavefront.txt): fp32 at every size measured
    // (2 frames x 256 streams: 0.19 against 0.23 ms; 32 x 256: 1.13 against 2.52; 32 x 4 096: 11.4 against 13.5); bf16 up to 768 streams
    // (32 frames x 512 streams: 0.49 against 0.68 ms; 768: 0.66 against 0.73; 1 024: equal at best -- beyond, the chunked kernels'
    // resident weights win -- except in calls of 2-4 frames)
    const bool wave_wins = prec_ == kBf16 ? (mtb <= 48 || (mtb <= 64 && T <= 4)) : mtb <= 256;
    const bool wave = T > 1 && (dev_wave_mt_ >= 0 ? mtb <= dev_wave_mt_ : wave_wins) && !no_small_ && !debug_taps_ && (only < 0 || only == kClsGru) && wave_fits();
    // One-frame bf16 calls whose m-tiles come in whole quads (from 60 m-tiles on): a GRU layer is ONE launch (kns_gruq.hip) -- input
    // GEMM, recurrent GEMM and gates fused over CU quads.  Same arithmetic as the two-kernel form, bit for bit
    // (tests/test_gpu_parity.py::test_alternative_kernels_give_identical_pcm).
    const bool quad = use_quad_ && !small && T == 1;
    auto gru_quad = [&](const void *a0, int nb0, const void *a1, const void *wih, const float *bih, const void *whh,
                        const float *bhh, int layer, void *hseq, const StageDev *head = nullptr) {
        GruQuadArgs g;
        if (head) {  // the narrow head of the stage before, inside this launch (one-frame calls)
            g.yh = d_hseq_b_;
            g.yw = head->w_head;
            g.yb = head->b_head;
            g.yvalid = head->head_dim;
        }
        g.a0 = a0;
        g.a1 = a1;
        g.wih = wih;
        g.bih = bih;
        g.whh = whh;
        g.bhh = bhh;
        g.hstate_in = d_hstate_[hs_cur_] + (size_t) layer * mtb * kUnitTiles * 256;
        g.hstate_out = d_hstate_[hs_cur_ ^ 1] + (size_t) layer * mtb * kUnitTiles * 256;
        g.hseq = hseq;
        g.nb0 = nb0;
        g.mtiles = mtb;
        g.dbg = d_qdbg_;
        g.dbg_block = qdbg_block_;
        tick(kClsGru);
        if (only < 0 || only == kClsGru) launch_gru_quad(g, stream_);
        tock(kClsGru);
    };

    const bool small_steps = T > 1 && mtb <= steps_mt && prec_ != kBf16 && !no_small_ && !wave;
    last_route_ = small ? kRouteSmall : wave ? kRouteWave : small_steps ? kRouteSmallSteps : quad ? kRouteQuad1 : kRouteChunked;
    // ---- mid-size batches: the layer pipeline over sub-chunks of frames (see the chunk loop below)
    const int pipe_chunk = dev_pipe_chunk_ > 0 ? dev_pipe_chunk_ : (T + 1) / 2;
    const int pipe_mt = dev_pipe_mt_ >= 0 ? dev_pipe_mt_ : 144;
    const int pipe_grid = dev_pipe_grid_ > 0 ? dev_pipe_grid_ : 128;
    const int pipe_streams = dev_pipe_streams_ > 0 && dev_pipe_streams_ <= kPipeStreams ? dev_pipe_streams_ : 2;
    const bool pipelined = !wave && !small && !small_steps && !quad && prec_ == kBf16 && mtb <= pipe_mt && T >= 32 && T >= 2 * pipe_chunk - 1 && !profiling_ &&
                           !debug_taps_ && only < 0 && pipe_ready();
    int nchunks = 1;
    if (pipelined) {
        nchunks = (T + pipe_chunk - 1) / pipe_chunk;
        if (T - (nchunks - 1) * pipe_chunk < pipe_chunk / 2 && nchunks > 2) --nchunks;  // (a short tail joins the last chunk)
    }
    // ... whose analysis and synthesis are cut into the same slices of frames when the spectrum is rebuilt from the PCM and the front-end
    // is folded (no launch between analysis and the first stage): slice c's analysis then runs beside slice c - 1's layers and slice
    // c's synthesis beside slice c + 1's -- 96 of a 1 024-stream call's 1 157 us were the two STFT launches alone at either end
    // (measured, tools/pipe_sweep.py: +1.3 % at 1 024 and 1 536 streams x 64 frames, nothing at 800, -0.3 % at 2 048; at 32 frames per call
    // -1 ... -1.5 % from 1 024 streams on -- the slices' launches then cost what their overlap gives: long calls of up to 112 m-tiles only)
    const bool stft_sliced = pipelined && fold_ && recompute && !an.write_spec && taps_ == 1 && !dev_pipe_whole_stft_ &&
                             (dev_pipe_chunk_ > 0 || (T >= 48 && mtb <= 112));
    auto analysis_slice = [&](int t0c, int Tc, bool first, bool last, hipStream_t st) {
        AnalysisArgs a = an;
        a.pcm = d_pcm + (size_t) t0c * kFrame;
        a.feat = feat_now + (size_t) t0c * feat_frame_bytes_;
        a.T = Tc;
        a.pitch = T;
        a.prev_in_pcm = first ? 0 : 1;
        a.write_hist = last ? 1 : 0;
        int seg = Tc;
        while (seg > 1 && (Bpad_ / 16) * ((Tc + seg - 1) / seg) < 1024) seg = (seg + 1) / 2;
        a.seg = dev_analysis_seg_ > 0 ? (dev_analysis_seg_ < Tc ? dev_analysis_seg_ : Tc) : seg;
        launch_analysis(a, st);
    };
    if (!stft_sliced) {
        tick(kClsAnalysis);
        if (only < 0 || only == kClsAnalysis) launch_analysis(an, stream_);
        tock(kClsAnalysis);
    }
    if (!in_place) hist_cur_ ^= 1;
    // front-end: e = features . W_in + b_in
    // (bf16 with a one-frame front-end: folded into the stage-input GEMMs, which read the features themselves -- no launch, no `e`)
    if (!fold_)
        gemm(kClsGemmHead, nullptr, 0, roll_in_analysis ? d_fhist_ : d_feat_, nbf_, w_in_, b_in_, d_e_, nbh_ * pi_.npb, kHidden, kOutAPlain,
             taps_);
    const void *stage_in = fold_ ? (const void *) feat_now : (const void *) d_e_;  // the e part of every stage's layer-A input
    if (taps_ > 1 && !roll_in_analysis)  // the last taps - 1 frames of [context | call] are the next call's context
        (void) hipMemcpyAsync(fhist_last, (char *) d_feat_ + (size_t) T * feat_frame_bytes_, (size_t) (taps_ - 1) * feat_frame_bytes_,
                              hipMemcpyDeviceToDevice, stream_);
    // One-frame calls through the quad kernel: the narrow head of stage s (271 -> 1, 5, 40) is computed inside stage s + 1's first
    // layer launch instead of in a launch of its own (kns_gruq.hip, kHead): 15 launches per frame step become 12.
    // ... and the mask head in the synthesis launch (bf16, stored spectrum -- what a one-frame call uses; kns_stft.hip, kMaskIn)
    const bool mask_in_synthesis = T == 1 && prec_ == kBf16 && fuse_front_ && !debug_taps_ && !recompute &&
                                   sd_[kStages - 1].head_tiles == kMaskTiles;
    mask_valid_ = !mask_in_synthesis;
    bool head_in_next = false;  // stage s - 1's head has been left to this stage's first layer
    bool head_in_recurrent = false;  // this stage's head was computed by its layer-B recurrent launch
    if (wave) {
        run_wave(T, mtb);
        if (!mask_in_synthesis)
            gemm(kClsGemmHead, nullptr, 0, d_hseq_b_, nbh_, sd_[kStages - 1].w_head, sd_[kStages - 1].b_head, d_mask_,
                 sd_[kStages - 1].head_tiles, kBins, kOutMask);
    }
    // Mid-size batches (round 6): with fewer than ~160 m-tiles the weight-resident recurrent launches occupy as many CUs and take the
    // same ~150 us per 64 frames as with 256 -- at 1 024 streams eight of them are 1.23 of the call's 1.37 ms on a quarter of the chip.
    // The layers of a call depend on each other frame by frame only, so the call is cut into sub-chunks of frames and the (layer, chunk)
    // grid runs as a wavefront over two streams: chunk c's 18 launches in order on stream c mod S, its recurrent launch of layer l
    // behind chunk c - 1's (an event per (chunk, layer)); up to S recurrent launches of different layers then share the chip.  Same
    // kernels on slices of the same buffers: the same bits (chunking is invisible).  Hidden state: chunk c reads ping-pong buffer
    // (hs_cur_ + c) & 1 and writes the other one.
    // Measured (tools/pipe_sweep.py, profiles/r06_pipe_sweep.txt): TWO sub-chunks on two streams is the best form at every size -- 64 frames x
    // 1 024 streams 1.34 -> 1.06 ms, x 2 048 streams 1.68 -> 1.48 ms; three chunks equal it, four and more lose (a launch of 8-16 frames is
    // mostly prologue, and every cross-stream event costs microseconds) -- and it stops paying at 160 m-tiles.
    // synthesis of frames [t0c, t0c + Tc) of the call (the whole call unless the STFT stages are sliced): slice number `ci` reads the
    // overlap-add tail from ping-pong buffer (tail_cur_ + ci) & 1 and leaves it in the other one
    auto synthesis_slice = [&](int t0c, int Tc, int ci, hipStream_t st) {
        SynthesisArgs sy;
        sy.spec = d_spec_;
        sy.mask = (const float *) ((const char *) d_mask_ + (size_t) t0c * mtb * kMaskTiles * 256 * (prec_ == kBf16 ? 2 : 4));
        sy.window = d_window_;
        sy.twiddle = d_twiddle_;
        const int tc = (tail_cur_ + ci) & 1;
        sy.tail_in = d_tail_[tc];
        sy.tail_out = d_tail_[in_place ? tc : tc ^ 1];
        const int seg_env = dev_synth_seg_;
        // two segments per stream tile (512 workgroups at B = 4096) measured best: fewer, longer segments amortise the
        // one replayed frame; a single segment leaves half the chip without a second workgroup to overlap with
        // ... and with few stream tiles the segments shrink (down to one frame) until there are about two workgroups per CU
        int seg_auto = Tc <= 4 ? Tc : ((Tc + 1) / 2 > 4 ? (Tc + 1) / 2 : 4);
        while (seg_auto > 1 && mtb * ((Tc + seg_auto - 1) / seg_auto) < 512) seg_auto = (seg_auto + 1) / 2;  // (few streams: down to one frame + its replay)
        const int seg = seg_env > 0 ? seg_env : seg_auto;
        sy.seg = Tc <= seg ? Tc : seg;
        sy.out = d_out + (size_t) t0c * kFrame;
        sy.pcm = d_pcm + (size_t) t0c * kFrame;
        sy.hist_in = hist_before;
        sy.recompute = recompute;
        sy.mask_fp16 = prec_ == kBf16;
        if (mask_in_synthesis) {
            sy.mask_h = d_hseq_b_;
            sy.mask_w = sd_[kStages - 1].w_head;
            sy.mask_b = sd_[kStages - 1].b_head;
        }
        sy.B = B_;
        sy.Bpad = Bpad_;
        sy.T = Tc;
        sy.pitch = T;
        sy.prev_in_pcm = t0c > 0 ? 1 : 0;
        launch_synthesis(sy, st);
    };
    if (pipelined) (void) hipEventRecord(pipe_fork_, stream_);
    for (int c = 0; c < nchunks && !wave; ++c) {
      if (pipelined) {
          c_t0 = c * pipe_chunk;
          c_T = c == nchunks - 1 ? T - c_t0 : pipe_chunk;
          c_hs = (hs_cur_ + c) & 1;
          c_stream = pipe_stream_[c % pipe_streams];
          c_grid = pipe_grid;
          if (c < pipe_streams) (void) hipStreamWaitEvent(c_stream, pipe_fork_, 0);
          if (stft_sliced) analysis_slice(c_t0, c_T, c == 0, c == nchunks - 1, c_stream);
      }
      auto gru_dep = [&](const void *whh, const float *bhh, int layer, void *hseq, const StageDev *head = nullptr) {
          if (pipelined && c > 0) (void) hipStreamWaitEvent(c_stream, pipe_ev_[(c - 1) % kPipeRing][layer], 0);
          gru(whh, bhh, layer, hseq, head);
          if (pipelined && c + 1 < nchunks) (void) hipEventRecord(pipe_ev_[c % kPipeRing][layer], c_stream);
      };
      for (int s = 0; s < kStages; ++s) {
        const StageDev &d = sd_[s];
        head_in_recurrent = false;
        // (d.ypad: the previous head's few values sit in the padding of the features' last k-block -- no y part of its own)
        const void *yprev = s && !d.ypad ? d_y_[s - 1] : nullptr;
        const int nby = s && !d.ypad ? nby_[s - 1] : 0;
        if (small) {
            gru_small(yprev, nby, stage_in, d.w_ih_a, d.b_ih_a, d.w_hh_a, d.b_hh_a, 2 * s, d_hseq_a_, 0, head_in_next ? &sd_[s - 1] : nullptr);
            gru_small(nullptr, 0, d_hseq_a_, d.w_ih_b, d.b_ih_b, d.w_hh_b, d.b_hh_b, 2 * s + 1, d_hseq_b_);
        } else if (small_steps) {
            for (int t = 0; t < T; ++t) gru_small(yprev, nby, stage_in, d.w_ih_a, d.b_ih_a, d.w_hh_a, d.b_hh_a, 2 * s, d_hseq_a_, t);
            for (int t = 0; t < T; ++t)
                gru_small(nullptr, 0, d_hseq_a_, d.w_ih_b, d.b_ih_b, d.w_hh_b, d.b_hh_b, 2 * s + 1, d_hseq_b_, t);
        } else {
            if (quad && nby <= quad_nb0_max_) {
                gru_quad(yprev, nby, stage_in, d.w_ih_a, d.b_ih_a, d.w_hh_a, d.b_hh_a, 2 * s, d_hseq_a_, head_in_next ? &sd_[s - 1] : nullptr);
            } else {
                gemm(kClsGemmIn, yprev, nby, stage_in, nbh_, d.w_ih_a, d.b_ih_a, d_gi_, kGateTiles, 3 * kHidden, kOutGi);
                gru_dep(d.w_hh_a, d.b_hh_a, 2 * s, d_hseq_a_);
            }
            if (quad) {
                gru_quad(nullptr, 0, d_hseq_a_, d.w_ih_b, d.b_ih_b, d.w_hh_b, d.b_hh_b, 2 * s + 1, d_hseq_b_);
            } else {
                // multi-frame calls: a narrow head whose values go into the features' padding (stages 0 and 1) is computed by layer
                // B's recurrent kernel itself, step by step, while the hidden vectors are in LDS: two head launches and two passes
                // over a hidden sequence (2 x 151 MB at the bench shape) less
                head_in_recurrent = T > 1 && s < kStages - 1 && sd_[s + 1].ypad && d.head_tiles == pi_.npb && fuse_head_ && !debug_taps_ &&
                                    !(dev_variant_ & kDevGruStream);
                gemm(kClsGemmIn, nullptr, 0, d_hseq_a_, nbh_, d.w_ih_b, d.b_ih_b, d_gi_, kGateTiles, 3 * kHidden, kOutGi);
                gru_dep(d.w_hh_b, d.b_hh_b, 2 * s + 1, d_hseq_b_, head_in_recurrent ? &d : nullptr);
            }
        }
        head_in_next = s < kStages - 1 && T == 1 && fuse_head_ && nby_[s] >= 1 && d.head_tiles == pi_.npb * nby_[s] && !debug_taps_ &&
                       (small || (quad && nby_[s] <= quad_nb0_max_));
        if (head_in_next || head_in_recurrent)
            ;
        else if (s < kStages - 1 && sd_[s + 1].ypad)  // into columns 257 ... of the features (k = 1 ... of their block 8)
            gemm(kClsGemmHead, nullptr, 0, d_hseq_b_, nbh_, d.w_head, d.b_head, feat_now, d.head_tiles, d.head_dim, kOutASigmoid, 1,
                 /*pad_nb=*/nbf_, /*pad_blk=*/kBins / pi_.kb, /*pad_kk0=*/kBins % pi_.kb);
        else if (s < kStages - 1)
            gemm(kClsGemmHead, nullptr, 0, d_hseq_b_, nbh_, d.w_head, d.b_head, d_y_[s], d.head_tiles, d.head_dim,
                 kOutASigmoid);
        else if (!mask_in_synthesis)
            gemm(kClsGemmHead, nullptr, 0, d_hseq_b_, nbh_, d.w_head, d.b_head, d_mask_, d.head_tiles, kBins, kOutMask);
    }
      if (stft_sliced) {  // this slice's synthesis, behind the previous slice's (the overlap-add tail goes from one to the next)
          if (c > 0) (void) hipStreamWaitEvent(c_stream, pipe_syn_[(c - 1) % kPipeRing], 0);
          synthesis_slice(c_t0, c_T, c, c_stream);
          if (c + 1 < nchunks) (void) hipEventRecord(pipe_syn_[c % kPipeRing], c_stream);
      }
      if (pipelined && (c >= nchunks - pipe_streams)) {  // the last chunk of each stream: the handle's stream continues behind it
          (void) hipEventRecord(pipe_join_[c % pipe_streams], c_stream);
          (void) hipStreamWaitEvent(stream_, pipe_join_[c % pipe_streams], 0);
      }
    }
    c_t0 = 0, c_T = T, c_hs = hs_cur_, c_stream = stream_, c_grid = 0;

    if (!stft_sliced) {
        tick(kClsSynthesis);
        if (only < 0 || only == kClsSynthesis) synthesis_slice(0, T, 0, stream_);
        tock(kClsSynthesis);
    }
    hs_cur_ = small_steps || wave ? (hs_cur_ + T) & 1 : pipelined ? (hs_cur_ + nchunks) & 1 : hs_cur_ ^ 1;
    if (pipelined) last_route_ = kRoutePipelined;
    if (!in_place) tail_cur_ = stft_sliced ? (tail_cur_ + nchunks) & 1 : tail_cur_ ^ 1;

    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        *err = std::string("HIP launch error: ") + hipGetErrorString(e);
        return false;
    }
    return true;
}

static std::mutex g_capture_mutex;

enum PointerKind { kPtrPageable = 0, kPtrPinned = 1, kPtrDevice = 2 };

static PointerKind pointer_kind(const void *p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void) hipGetLastError();
        return kPtrPageable;
    }
    if (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) return kPtrDevice;
    return attr.type == hipMemoryTypeHost ? kPtrPinned : kPtrPageable;
}

// rows x width bytes between pitched host buffers; large copies are split over a few threads (one core moves ~10 GB/s,
// which is a tenth of what the GPU consumes)
static void host_copy_2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows) {
    auto span = [=](size_t r0, size_t r1) {
        if (dpitch == width && spitch == width) {
            memcpy((char *) dst + r0 * width, (const char *) src + r0 * width, (r1 - r0) * width);
            return;
        }
        for (size_t r = r0; r < r1; ++r) memcpy((char *) dst + r * dpitch, (const char *) src + r * spitch, width);
    };
    const size_t bytes = width * rows;
    unsigned hw = std::thread::hardware_concurrency();
    size_t nthreads = bytes >= ((size_t) 4 << 20) ? bytes >> 21 : 1;  // one thread per 2 MiB ...
    if (nthreads > 8) nthreads = 8;                                  // ... up to 8
    if (hw && nthreads > hw) nthreads = hw;
    if (nthreads > rows) nthreads = rows ? rows : 1;
    if (nthreads <= 1) {
        span(0, rows);
        return;
    }
    std::vector<std::thread> pool;
    for (size_t i = 1; i < nthreads; ++i) pool.emplace_back(span, rows * i / nthreads, rows * (i + 1) / nthreads);
    span(0, rows / nthreads);
    for (std::thread &t : pool) t.join();
}

// Host-pointer call in sub-chunks of host_chunk_ frames: H2D of chunk c+1, the kernels of chunk c and D2H of chunk c-1 run
// on three streams.  Pinned user buffers (hipHostMalloc / hipHostRegister / pv_koala_batch_host_alloc) are read and written
// by the copy engines directly (strided 2D copies); pageable ones go through the engine's pinned staging slots.
// The sub-chunk lengths of a synchronous host call of T frames.  What the caller waits for is
//   copy-in(first chunk) + sum of the chunks' kernel times + copy-out(last chunk),
// and a chunk's kernels cost ~0.17 ms of weight-resident prologues whatever its length (16 launches): short chunks at the two ENDS
// (little exposed copy time), long ones in the middle (few prologues) -- for 64 frames 4 8 12 16 12 8 4 instead of 4 x 16: 57.8 -> 61.6 M
// frames/s with page-locked buffers (profiles/r05_host_sched.txt; a model with the measured copy rate and K(T) had said 61.5).  No chunk
// exceeds host_chunk_ (a staging slot).
std::vector<int> Engine::host_schedule(int T) const {
    std::vector<int> sched;
    if (!dev_host_sched_.empty()) {  // developer override: the given lengths, repeated / truncated to T
        int left = T;
        for (size_t i = 0; left > 0; ++i) {
            const int c = std::min(left, std::max(1, std::min(host_chunk_, dev_host_sched_[i % dev_host_sched_.size()])));
            sched.push_back(c);
            left -= c;
        }
        return sched;
    }
    // a ramp of P/4, P/2, 3P/4 frames at either end (as many of its steps as the call has room for), chunks of at most P between them
    const int P = host_chunk_;
    int ramp[3] = {std::max(1, P / 4), std::max(1, P / 2), std::max(1, (3 * P) / 4)}, steps = 3;
    while (steps > 0 && 2 * (ramp[0] + (steps > 1 ? ramp[1] : 0) + (steps > 2 ? ramp[2] : 0)) + P > T) --steps;
    int ends = 0;
    for (int i = 0; i < steps; ++i) ends += ramp[i];
    const int mid = T - 2 * ends, n = (mid + P - 1) / P;
    for (int i = 0; i < steps; ++i) sched.push_back(ramp[i]);
    for (int i = 0; i < n; ++i) sched.push_back(mid / n + (i < mid % n ? 1 : 0));
    for (int i = steps - 1; i >= 0; --i) sched.push_back(ramp[i]);
    return sched;
}

bool Engine::process_host_pipelined(int T, const int16_t *pcm, int16_t *out, bool pinned, std::string *err) {
    const std::vector<int> sched = host_schedule(T);
    const int n = (int) sched.size();
    std::vector<int> first(n + 1, 0);  // first frame of chunk c
    for (int c = 0; c < n; ++c) first[c + 1] = first[c] + sched[c];
    const size_t row_user = (size_t) T * kFrame * 2;           // pitch of the caller's [B][T * 256] matrices
    const size_t slot = (size_t) B_ * host_chunk_ * kFrame;    // int16 elements per staging slot (the longest chunk fits)
    bool ok = true;
    auto check = [&](hipError_t e) { ok = ok && e == hipSuccess; };
    auto drain_to_user = [&](int c) {  // chunk c's output: staging slot -> caller (pageable path)
        const int s = c & 1, tc = sched[c];
        check(hipEventSynchronize(ev_out_[s]));
        host_copy_2d((char *) out + (size_t) first[c] * kFrame * 2, row_user, h_out_ + s * slot, (size_t) tc * kFrame * 2,
                     (size_t) tc * kFrame * 2, B_);
    };
    for (int c = 0; c < n && ok; ++c) {
        const int s = c & 1, tc = sched[c];
        const size_t width = (size_t) tc * kFrame * 2;
        const char *src = (const char *) pcm + (size_t) first[c] * kFrame * 2;
        // ---- copy-in (slot s was last read by the kernels of chunk c - 2)
        if (c >= 2) check(hipStreamWaitEvent(copy_in_, ev_done_[s], 0));
        if (pinned) {
            check(hipMemcpy2DAsync(d_in_ + s * slot, width, src, row_user, width, B_, hipMemcpyHostToDevice, copy_in_));
        } else {
            if (c >= 2) check(hipEventSynchronize(ev_in_[s]));  // the H2D of chunk c - 2 has left the staging slot
            host_copy_2d(h_in_ + s * slot, width, src, row_user, width, B_);
            check(hipMemcpyAsync(d_in_ + s * slot, h_in_ + s * slot, width * B_, hipMemcpyHostToDevice, copy_in_));
        }
        check(hipEventRecord(ev_in_[s], copy_in_));
        // ---- kernels (d_out_ slot s was last drained by the D2H of chunk c - 2)
        check(hipStreamWaitEvent(stream_, ev_in_[s], 0));
        if (c >= 2) check(hipStreamWaitEvent(stream_, ev_out_[s], 0));
        if (ok && !run_device(tc, d_in_ + s * slot, d_out_ + s * slot, err)) {
            // copies of earlier sub-chunks may still be writing into the caller's buffers: let them finish first
            (void) hipStreamSynchronize(copy_in_);
            (void) hipStreamSynchronize(copy_out_);
            (void) hipStreamSynchronize(stream_);
            return false;
        }
        check(hipEventRecord(ev_done_[s], stream_));
        // ---- copy-out
        check(hipStreamWaitEvent(copy_out_, ev_done_[s], 0));
        if (pinned) {
            check(hipMemcpy2DAsync((char *) out + (size_t) first[c] * kFrame * 2, row_user, d_out_ + s * slot, width, width, B_,
                                   hipMemcpyDeviceToHost, copy_out_));
        } else {
            if (c >= 2) drain_to_user(c - 2);  // frees staging slot s
            check(hipMemcpyAsync(h_out_ + s * slot, d_out_ + s * slot, width * B_, hipMemcpyDeviceToHost, copy_out_));
        }
        check(hipEventRecord(ev_out_[s], copy_out_));
    }
    if (ok && !pinned) {
        if (n >= 2) drain_to_user(n - 2);
        drain_to_user(n - 1);
    }
    if (ok && pinned) {
        check(hipEventSynchronize(ev_out_[(n - 1) & 1]));
        if (n >= 2) check(hipEventSynchronize(ev_out_[(n - 2) & 1]));
    }
    if (ok) check(hipStreamSynchronize(stream_));
    if (!ok) {
        *err = std::string("HIP error: ") + hipGetErrorString(hipGetLastError());
        (void) hipDeviceSynchronize();
        return false;
    }
    return true;
}

void Engine::set_stream(hipStream_t s) {
    hipStream_t next = s ? s : own_stream_;
    if (next == stream_) return;
    (void) hipSetDevice(device_);
    std::string ignored;
    (void) drain_async(&ignored);
    if (hipStreamSynchronize(stream_) != hipSuccess) (void) hipGetLastError();
    stream_ = next;
}

bool Engine::drain_async(std::string *err) {
    bool ok = true;
    for (int i = 0; i < 4; ++i) {
        if (!async_busy_[i]) continue;
        if (hipEventSynchronize(aev_out_[i]) != hipSuccess) ok = false;
        async_busy_[i] = false;
    }
    if (!ok) {
        *err = std::string("HIP error: ") + hipGetErrorString(hipGetLastError());
        (void) hipDeviceSynchronize();
    }
    return ok;
}

bool Engine::async_wait(int max_in_flight, std::string *err) {
    if (max_in_flight <= 0) return drain_async(err);
    for (int j = 3; j > max_in_flight; --j) {  // the calls async_n_ - 3 ... async_n_ - max_in_flight - 1, oldest first
        if ((unsigned) j > async_n_) continue;
        const int ring = (int) ((async_n_ - (unsigned) j) & 3u);
        if (!async_busy_[ring]) continue;
        if (hipEventSynchronize(aev_out_[ring]) != hipSuccess) {
            *err = std::string("HIP error: ") + hipGetErrorString(hipGetLastError());
            return false;
        }
        async_busy_[ring] = false;
    }
    return true;
}

// One asynchronous host call = H2D on copy_in_, the kernels on the handle's stream, D2H on copy_out_, chained by events.  Up to
// THREE calls are in flight: with two, a caller alternating between two buffer pairs cannot issue call n + 2 before call n's copy-out has
// finished, which puts a slot's copy-in, kernels and copy-out in series (measured: 4.03 ms per 4096 x 64 frames = (2.8 + 2.6 + 2.8) / 2);
// with three, the link works in both directions under the kernels (~48 GB/s each way at once, profiles/r05_pcie_probe.txt).  Device
// staging alternates between two slots: the input half of slot s is free once the kernels of call n - 2 have read it, the output half
// once call n - 2's copy-out has -- both waited for on the device, by the stream that needs it.  A synchronous call of that size cannot
// hide its first copy-in and last copy-out and has to cut its kernels into short, less efficient sub-chunks (process_host_pipelined).
bool Engine::process_host_async(int T, const int16_t *pcm, int16_t *out, std::string *err) {
    (void) hipSetDevice(device_);
    if (pointer_kind(pcm) != kPtrPinned || pointer_kind(out) != kPtrPinned) {
        *err = "asynchronous host calls need page-locked `pcm` and `enhanced` (pv_koala_batch_host_alloc, hipHostMalloc or hipHostRegister).";
        return false;
    }
    const size_t bytes = (size_t) B_ * T * kFrame * 2;
    {
        const char *pa = (const char *) pcm, *pb = (const char *) out;
        if (pa != pb && pa < pb + bytes && pb < pa + bytes) {
            *err = "`pcm` and `enhanced` overlap partially.";
            return false;
        }
    }
    if (!async_ready_) {  // set only once EVERY event and both buffers exist: a partial failure is retried by the next call
        bool ok = true;
        for (int i = 0; i < 4 && ok; ++i)
            if (!aev_out_[i]) ok = hipEventCreateWithFlags(&aev_out_[i], hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 2 && ok; ++i) {
            if (!aev_in_[i]) ok = hipEventCreateWithFlags(&aev_in_[i], hipEventDisableTiming) == hipSuccess;
            if (ok && !aev_done_[i]) ok = hipEventCreateWithFlags(&aev_done_[i], hipEventDisableTiming) == hipSuccess;
        }
        if (ok && !d_in2_) d_in2_ = (int16_t *) dalloc((size_t) B_ * Tmax_ * kFrame * 2, false);
        if (ok && !d_out2_) d_out2_ = (int16_t *) dalloc((size_t) B_ * Tmax_ * kFrame * 2, false);
        ok = ok && d_in2_ && d_out2_;
        if (!ok) {
            (void) hipGetLastError();
            *err = "Failed to allocate the second staging slot of asynchronous host calls.";
            return false;
        }
        async_ready_ = true;
    }
    const unsigned n = async_n_;
    const int s = (int) (n & 1u), ring = (int) (n & 3u), ring3 = (int) ((n - 3u) & 3u), ring2 = (int) ((n - 2u) & 3u);
    if (n >= 3 && async_busy_[ring3]) {  // the window: the call three back has completed before this one is accepted
        if (hipEventSynchronize(aev_out_[ring3]) != hipSuccess) {
            *err = std::string("HIP error: ") + hipGetErrorString(hipGetLastError());
            return false;
        }
        async_busy_[ring3] = false;
    }
    int16_t *din = s ? d_in2_ : d_in_, *dout = s ? d_out2_ : d_out_;
    bool ok = true;
    if (n >= 2) ok = hipStreamWaitEvent(copy_in_, aev_done_[s], 0) == hipSuccess;  // the kernels of call n - 2 have read this slot's input
    ok = ok && hipMemcpyAsync(din, pcm, bytes, hipMemcpyHostToDevice, copy_in_) == hipSuccess;
    ok = ok && hipEventRecord(aev_in_[s], copy_in_) == hipSuccess;
    ok = ok && hipStreamWaitEvent(stream_, aev_in_[s], 0) == hipSuccess;
    if (n >= 2) ok = ok && hipStreamWaitEvent(stream_, aev_out_[ring2], 0) == hipSuccess;  // ... and its copy-out this slot's output
    if (!ok) {
        *err = std::string("HIP error: ") + hipGetErrorString(hipGetLastError());
        (void) hipDeviceSynchronize();  // (a copy-in may have been enqueued: nothing of this call stays in flight)
        return false;
    }
    if (!run_device(T, din, dout, err)) {
        (void) hipDeviceSynchronize();
        return false;
    }
    ok = hipEventRecord(aev_done_[s], stream_) == hipSuccess;
    ok = ok && hipStreamWaitEvent(copy_out_, aev_done_[s], 0) == hipSuccess;
    // (as a 2-D copy of B_ rows: the runtime hands those to the copy engines; a plain hipMemcpyAsync into page-locked memory went
    // through its copy KERNEL, which took compute units from the engine's own kernels -- profiles/r05_host_async_trace.txt)
    ok = ok && hipMemcpy2DAsync(out, (size_t) T * kFrame * 2, dout, (size_t) T * kFrame * 2, (size_t) T * kFrame * 2, B_, hipMemcpyDeviceToHost,
                                copy_out_) == hipSuccess;
    ok = ok && hipEventRecord(aev_out_[ring], copy_out_) == hipSuccess;
    if (!ok) {
        *err = std::string("HIP error: ") + hipGetErrorString(hipGetLastError());
        (void) hipDeviceSynchronize();
        return false;
    }
    async_busy_[ring] = true;
    ++async_n_;
    return true;
}

bool Engine::process(int T, const int16_t *pcm, int16_t *out, std::string *err, bool host_pointers) {
    (void) hipSetDevice(device_);
    if (async_n_ && !drain_async(err)) return false;
    const size_t bytes = (size_t) B_ * T * kFrame * 2;
    // (the single-stream ABI takes host buffers by contract: no driver query per frame on the latency path)
    const PointerKind kin = host_pointers ? kPtrPageable : pointer_kind(pcm), kout = host_pointers ? kPtrPageable : pointer_kind(out);
    if ((kin == kPtrDevice) != (kout == kPtrDevice)) {
        *err = "`pcm` and `enhanced` must both be host or both be device memory.";
        return false;
    }
    if (kin == kPtrDevice) {
        const size_t n = (size_t) B_ * T * kFrame;
        const bool overlap = pcm < out + n && out < pcm + n;
        return run_device(T, pcm, out, err, !overlap);
    }
    if (T > host_chunk_ && bytes >= host_pipeline_min_bytes_) {
        const char *pa = (const char *) pcm, *pb = (const char *) out;
        // partial overlap: chunk c's copy-out would land on input chunks not yet read.  EXACT aliasing (enhanced == pcm) is fine:
        // chunk c's output columns are chunk c's own input columns, which have been staged by then
        if (pa != pb && pa < pb + bytes && pb < pa + bytes) {
            *err = "`pcm` and `enhanced` overlap: host-memory calls of this size are pipelined in sub-chunks and need "
                   "disjoint buffers.";
            return false;
        }
        // On a caller's stream the kernels of the sub-chunks still run on the handle's OWN stream, fenced behind whatever the caller
        // has enqueued (the call is synchronous, so everything the caller enqueues later is behind it anyway): the handle's three
        // streams were created together and sit on three different hardware queues, while a foreign stream may share its queue with
        // one of the copy streams -- measured: 40 instead of 62 M frames/s for page-locked buffers (profiles/r06_bench.json notes)
        hipStream_t user = stream_;
        if (user != own_stream_ && !profiling_) {
            if (!host_fork_ && hipEventCreateWithFlags(&host_fork_, hipEventDisableTiming) != hipSuccess) {
                (void) hipGetLastError();
                host_fork_ = nullptr;
            }
            if (host_fork_ && hipEventRecord(host_fork_, user) == hipSuccess && hipStreamWaitEvent(own_stream_, host_fork_, 0) == hipSuccess)
                stream_ = own_stream_;
        }
        const bool done = process_host_pipelined(T, pcm, out, kin == kPtrPinned && kout == kPtrPinned, err);
        stream_ = user;
        return done;
    }
    memcpy(h_in_, pcm, bytes);
    if (T == 1 && use_graph_ && stream_ == own_stream_ && !profiling_) {
        // frame-by-frame streaming: (copy-in,) the kernels of one frame (and copy-out) replayed as one hipGraph
        // A captured frame has the state buffers it reads and writes baked in: one graph per combination of the three
        // ping-pong indices.  (A single-frame call leaves history and tail in place and flips the hidden state only, but
        // multi-frame calls in between flip the other two as well.)
        const int hs = hs_cur_;
        const int parity = hs | (hist_cur_ << 1) | (tail_cur_ << 2);
        if (!frame_graph_[parity]) {
            hipGraph_t graph = nullptr;
            // (relaxed mode and one capture at a time in the process: other threads -- other handles being created, the
            // host application -- may touch the legacy stream meanwhile, which would invalidate a stricter capture)
            std::lock_guard<std::mutex> capture_lock(g_capture_mutex);
            bool ok = hipStreamBeginCapture(stream_, hipStreamCaptureModeRelaxed) == hipSuccess;
            if (ok) {
                // small batches: the analysis kernel reads the frame from, and the synthesis kernel writes it to, the
                // pinned host staging buffers directly (device-visible memory): two copy nodes of ~4 us each less per frame
                const bool zero_copy = bytes <= 64 * 1024 && !no_zero_copy_;
                ok = zero_copy || hipMemcpyAsync(d_in_, h_in_, bytes, hipMemcpyHostToDevice, stream_) == hipSuccess;
                ok = ok && run_device(1, zero_copy ? h_in_ : d_in_, zero_copy ? h_out_ : d_out_, err);
                ok = ok && (zero_copy || hipMemcpyAsync(h_out_, d_out_, bytes, hipMemcpyDeviceToHost, stream_) == hipSuccess);
                // zero-copy frames end with the completion word (the output is already in host memory when that node runs)
                frame_graph_signals_[parity] = ok && zero_copy && spin_wait_ && h_frame_word_ && d_frame_count_;
                if (frame_graph_signals_[parity]) launch_frame_done(d_frame_count_, h_frame_word_, stream_);
                ok = (hipStreamEndCapture(stream_, &graph) == hipSuccess) && ok;
                ok = ok && hipGraphInstantiate(&frame_graph_[parity], graph, nullptr, nullptr, 0) == hipSuccess;
                if (graph) (void) hipGraphDestroy(graph);
            }
            hs_cur_ = hs;  // the capture only recorded the launches; run_device's bookkeeping is replayed below
            if (!ok) {
                // a failed capture must not leave the handle half-switched: nothing was executed, the ping-pong index is
                // back where it was, and this and every later frame take the plain copy / launch / copy path below
                (void) hipGetLastError();
                frame_graph_[parity] = nullptr;
                use_graph_ = false;
            }
        }
        if (use_graph_) {
            if (hipGraphLaunch(frame_graph_[parity], stream_) != hipSuccess) goto fail;
            hs_cur_ = hs ^ 1;
            if (frame_graph_signals_[parity]) {
                // Poll the frame's completion word (a frame takes 50-100 us: longer than the runtime's own active-wait window,
                // after which hipStreamSynchronize sleeps on an interrupt and wakes up whenever the host scheduler gets to it).
                // The first ~250 us -- two to five frame times -- in a tight loop, then yielding the core between polls (a host that
                // runs more handles than cores must not have its pollers starve each other); a frame that has not reported
                // after 20 ms is handed to hipStreamSynchronize, which also surfaces errors.
                const unsigned want = ++frame_seq_;
                const auto t0 = std::chrono::steady_clock::now();
                bool seen = false, yielding = false;
                for (unsigned spins = 0;; ++spins) {
                    if (__atomic_load_n(h_frame_word_, __ATOMIC_ACQUIRE) == want) {
                        seen = true;
                        break;
                    }
                    if (yielding) {
                        std::this_thread::yield();
                    } else {
#if defined(__x86_64__) || defined(__i386__)
                        __builtin_ia32_pause();
#elif defined(__aarch64__)
                        asm volatile("yield" ::: "memory");
#endif
                    }
                    if (yielding || (spins & 0xff) == 0xff) {
                        const auto dt = std::chrono::steady_clock::now() - t0;
                        if (dt > std::chrono::milliseconds(20)) break;
                        yielding = dt > std::chrono::microseconds(250);
                    }
                }
                if (!seen) {
                    if (hipStreamSynchronize(stream_) != hipSuccess) goto fail;
                    frame_seq_ = __atomic_load_n(h_frame_word_, __ATOMIC_ACQUIRE);  // (resynchronise the expectation)
                } else if ((want & 1023u) == 0) {
                    // (now and then: lets the runtime retire the launches it has been tracking; returns at once, the stream is idle)
                    if (hipStreamSynchronize(stream_) != hipSuccess) goto fail;
                }
            } else if (hipStreamSynchronize(stream_) != hipSuccess) {
                goto fail;
            }
            memcpy(out, h_out_, bytes);
            return true;
        }
    }
    if (hipMemcpyAsync(d_in_, h_in_, bytes, hipMemcpyHostToDevice, stream_) != hipSuccess) goto fail;
    if (!run_device(T, d_in_, d_out_, err)) return false;
    if (hipMemcpyAsync(h_out_, d_out_, bytes, hipMemcpyDeviceToHost, stream_) != hipSuccess) goto fail;
    if (hipStreamSynchronize(stream_) != hipSuccess) goto fail;
    memcpy(out, h_out_, bytes);
    return true;
fail:
    *err = std::string("HIP error: ") + hipGetErrorString(hipGetLastError());
    return false;
}

// ------------------------------------------------------------------------------------------------ debug taps

static float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t) (h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t u;
    if (e == 0) {  // zero / subnormal: value = m * 2^-24
        float f = (float) m * (1.0f / 16777216.0f);
        memcpy(&u, &f, 4);
        u |= sign;
    } else if (e == 31) {
        u = sign | 0x7f800000u | (m << 13);
    } else {
        u = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static float bf16_to_float(uint16_t h) {
    uint32_t u = (uint32_t) h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int64_t Engine::debug_read(int what, float *out, int64_t capacity, std::string *err) {
    (void) hipSetDevice(device_);
    if (hipStreamSynchronize(stream_) != hipSuccess) {
        *err = std::string("HIP error: ") + hipGetErrorString(hipGetLastError());
        return -1;
    }
    const int T = last_T_, mtb = Bpad_ / 16;
    auto fetch = [&](const void *d, size_t bytes) {
        std::vector<uint8_t> h(bytes);
        (void) hipMemcpyAsync(h.data(), d, bytes, hipMemcpyDeviceToHost, stream_);
        (void) hipStreamSynchronize(stream_);
        return h;
    };
    // logical element (row r of m-tile, k) of an A-packed buffer with nb blocks per m-tile
    auto a_at = [&](const std::vector<uint8_t> &h, size_t mtile, int nb, int r, int k) -> float {
        const size_t blk = mtile * nb + k / pi_.kb;
        const int off = pack_off(prec_, r, k % pi_.kb);
        if (prec_ == kBf16) {
            uint16_t v;
            memcpy(&v, h.data() + blk * 1024 + (size_t) off * 2, 2);
            return bf16_to_float(v);
        }
        float v;
        memcpy(&v, h.data() + blk * 1024 + (size_t) off * 4, 4);
        return v;
    };
    int64_t n = 0;
    if ((what == 0 && !feat_valid_) || (what == 2 && !mask_valid_)) {
        *err = "the last call kept this intermediate on chip (one-frame calls fuse the front-end / the mask head into the STFT "
               "launches): use the developer build with its debug-taps switch";
        return -1;
    }
    if (what == 4 && fold_) {
        *err = "this configuration forms no embedding (bf16, one-frame front-end: folded into the stage-input GEMMs)";
        return -1;
    }
    if (what == 0 || what == 4) {  // features / embedding
        const int width = what == 0 ? kBins : kHidden, nb = what == 0 ? nbf_ : nbh_;
        n = (int64_t) T * B_ * width;
        if (n > capacity) return -2;
        auto h = fetch(what == 0 ? (const char *) d_feat_ + (size_t) (taps_ - 1) * feat_frame_bytes_ : (const char *) d_e_,
                       (size_t) T * mtb * nb * 1024);
        for (int t = 0; t < T; ++t)
            for (int b = 0; b < B_; ++b)
                for (int k = 0; k < width; ++k)
                    out[((size_t) t * B_ + b) * width + k] = a_at(h, (size_t) t * mtb + b / 16, nb, b % 16, k);
    } else if (what == 1) {  // spectrum
        if (!spec_valid_) {  // multi-frame calls rebuild the spectrum from the PCM and store none
            *err = "the last call stored no spectrum (multi-frame calls recompute it): use the developer build with its "
                   "debug-taps switch";
            return -1;
        }
        n = (int64_t) T * B_ * kBins * 2;
        if (n > capacity) return -2;
        auto h = fetch(d_spec_, (size_t) T * Bpad_ * 256 * 8);
        const float *s = (const float *) h.data();
        for (int t = 0; t < T; ++t)
            for (int b = 0; b < B_; ++b) {
                const float *src = s + ((size_t) t * Bpad_ + b) * 512;
                float *dst = out + ((size_t) t * B_ + b) * kBins * 2;
                for (int k = 0; k < 256; ++k) {  // un-permute the kernels' lane order (kns_kernels.h)
                    const int c = k & 15, k2 = k >> 4, slot = ((k2 >> 1) * 16 + c) * 2 + (k2 & 1);
                    dst[2 * k] = src[2 * slot];
                    dst[2 * k + 1] = src[2 * slot + 1];
                }
                dst[512] = dst[1];  // Nyquist travels in the imaginary slot of bin 0
                dst[513] = 0.0f;
                dst[1] = 0.0f;
            }
    } else if (what == 2) {  // mask
        n = (int64_t) T * B_ * kBins;
        if (n > capacity) return -2;
        auto h = fetch(d_mask_, (size_t) T * mtb * kMaskTiles * 1024);
        const float *s = (const float *) h.data();
        const uint16_t *sh = (const uint16_t *) h.data();  // bf16 configuration: fp16 C fragments
        for (int t = 0; t < T; ++t)
            for (int b = 0; b < B_; ++b)
                for (int k = 0; k < kBins; ++k) {
                    const size_t idx = (((size_t) t * mtb + b / 16) * kMaskTiles + k / 16) * 256 + cpack_off(b % 16, k % 16);
                    out[((size_t) t * B_ + b) * kBins + k] = prec_ == kBf16 ? half_to_float(sh[idx]) : s[idx];
                }
    } else if (what == 3) {  // hidden state
        n = (int64_t) kGruLayers * B_ * kHidden;
        if (n > capacity) return -2;
        auto h = fetch(d_hstate_[hs_cur_], (size_t) kGruLayers * mtb * kUnitTiles * 1024);
        const float *s = (const float *) h.data();
        for (int l = 0; l < kGruLayers; ++l)
            for (int b = 0; b < B_; ++b)
                for (int k = 0; k < kHidden; ++k)
                    out[((size_t) l * B_ + b) * kHidden + k] =
                        s[(((size_t) l * mtb + b / 16) * kUnitTiles + k / 16) * 256 + cpack_off(b % 16, k % 16)];
#ifdef KNS_DEV
    } else if (what == 6) {  // developer build: the route of the last call (enum Route), what rode inside other launches
        if (capacity < 4) return -2;
        out[0] = (float) last_route_;
        out[1] = feat_valid_ ? 0.0f : 1.0f;   // front-end inside the analysis launch / features rolled into the history
        out[2] = mask_valid_ ? 0.0f : 1.0f;   // mask head inside the synthesis launch
        out[3] = spec_valid_ ? 1.0f : 0.0f;   // spectrum stored (not recomputed)
        return 4;
#endif
    } else if (what == 5) {  // developer build: stamps of the last fused layer launch, [8 waves][4 T][8] ticks since the first one
        if (!d_qdbg_) {
            *err = "no stamps: developer build with KOALA_AMD_QUAD_DBG=<workgroup>";
            return -1;
        }
        n = (int64_t) 8 * 4 * T * 8;
        if (n > capacity) return -2;
        auto h = fetch(d_qdbg_, (size_t) n * 8);
        const unsigned long long *st = (const unsigned long long *) h.data();
        unsigned long long t0 = ~0ull;
        for (int64_t i = 0; i < n; ++i)
            if (i % 8 < 6 && st[i] && st[i] < t0) t0 = st[i];
        // slots 0..5: ticks since the first stamp; slot 6: a 32-bit mask as two 16-bit halves is too wide for a float -- split
        // below; slot 7: a small count
        for (int64_t i = 0; i < n; ++i)
            out[i] = i % 8 == 7 ? (float) (st[i] & 0xffff) + 65536.0f * (float) (st[i - 1] >> 16 & 0xffff)  // count + mask's high half
                     : i % 8 == 6 ? (float) (st[i] & 0xffff)
                                  : (st[i] ? (float) (st[i] - t0) : -1.0f);
    } else {
        *err = "unknown debug tap";
        return -1;
    }
    return n;
}

}  // namespace kns
