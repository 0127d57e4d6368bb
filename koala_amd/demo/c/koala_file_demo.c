/*
 * koala_file_demo.c -- the drop-in boundary exercised from plain C.
 *
 * A C program that binds libpv_koala.so at run time exactly as a C host of the reference does (dlopen + dlsym of the
 * pv_koala.h entry points; the reference's own C demo, demo/c/koala_demo_file.c:262-527, is the model for WHAT is
 * bound: init, process, delay_sample, frame_length, version, error stack, delete) and runs a 16 kHz mono 16-bit WAV
 * through pv_koala_process frame by frame with the delay compensated.  With --streams N (N > 1) the same file is fed
 * to N lock-stepped streams through the batch extension (pv_koala_batch.h) in chunks of --frames frames.
 *
 *   gcc -O2 -o koala_file_demo koala_file_demo.c -ldl
 *   ./koala_file_demo -l libpv_koala.so -m model.kns -i noisy.wav -o clean.wav [-d gpu:0] [--streams 64 --frames 8]
 *
 * Written for this repository (own WAV reader/writer, own control flow).
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int32_t status_t;  /* pv_status_t: 0 = success */

typedef struct {
    void *lib;
    const char *(*status_to_string)(status_t);
    int32_t (*sample_rate)(void);
    int32_t (*frame_length)(void);
    const char *(*version)(void);
    status_t (*init)(const char *, const char *, const char *, void **);
    void (*del)(void *);
    status_t (*process)(void *, const int16_t *, int16_t *);
    status_t (*delay_sample)(const void *, int32_t *);
    status_t (*get_error_stack)(char ***, int32_t *);
    void (*free_error_stack)(char **);
    status_t (*batch_init)(const char *, const char *, const char *, int32_t, int32_t, int32_t, void **);
    status_t (*batch_process_chunk)(void *, int32_t, const int16_t *, int16_t *);
    void (*batch_delete)(void *);
} api_t;

static void *must_sym(void *lib, const char *name) {
    void *p = dlsym(lib, name);
    if (!p) {
        fprintf(stderr, "symbol `%s` is missing: %s\n", name, dlerror());
        exit(2);
    }
    return p;
}

static void bind(api_t *a, const char *path) {
    a->lib = dlopen(path, RTLD_NOW);
    if (!a->lib) {
        fprintf(stderr, "cannot load `%s`: %s\n", path, dlerror());
        exit(2);
    }
    *(void **) &a->status_to_string = must_sym(a->lib, "pv_status_to_string");
    *(void **) &a->sample_rate = must_sym(a->lib, "pv_sample_rate");
    *(void **) &a->frame_length = must_sym(a->lib, "pv_koala_frame_length");
    *(void **) &a->version = must_sym(a->lib, "pv_koala_version");
    *(void **) &a->init = must_sym(a->lib, "pv_koala_init");
    *(void **) &a->del = must_sym(a->lib, "pv_koala_delete");
    *(void **) &a->process = must_sym(a->lib, "pv_koala_process");
    *(void **) &a->delay_sample = must_sym(a->lib, "pv_koala_delay_sample");
    *(void **) &a->get_error_stack = must_sym(a->lib, "pv_get_error_stack");
    *(void **) &a->free_error_stack = must_sym(a->lib, "pv_free_error_stack");
    *(void **) &a->batch_init = must_sym(a->lib, "pv_koala_batch_init");
    *(void **) &a->batch_process_chunk = must_sym(a->lib, "pv_koala_batch_process_chunk");
    *(void **) &a->batch_delete = must_sym(a->lib, "pv_koala_batch_delete");
}

static void die_with_stack(const api_t *a, const char *what, status_t st) {
    fprintf(stderr, "%s failed with `%s`\n", what, a->status_to_string(st));
    char **msgs = NULL;
    int32_t n = 0;
    if (a->get_error_stack(&msgs, &n) == 0) {
        for (int32_t i = 0; i < n; ++i) fprintf(stderr, "  [%d] %s\n", i, msgs[i]);
        if (n) a->free_error_stack(msgs);
    }
    exit(1);
}

/* ---- minimal RIFF/WAVE I/O: PCM, 16 bit, mono */
static uint32_t rd32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t) p[3] << 24); }

static int16_t *read_wav(const char *path, int32_t want_rate, long *num_samples) {
    FILE *f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "cannot open `%s`\n", path);
        exit(1);
    }
    unsigned char h[12];
    if (fread(h, 1, 12, f) != 12 || memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) {
        fprintf(stderr, "`%s` is not a RIFF/WAVE file\n", path);
        exit(1);
    }
    int have_fmt = 0;
    int16_t *pcm = NULL;
    for (;;) {
        unsigned char c[8];
        if (fread(c, 1, 8, f) != 8) break;
        const uint32_t size = rd32(c + 4);
        if (!memcmp(c, "fmt ", 4)) {
            unsigned char fmt[16];
            if (size < 16 || fread(fmt, 1, 16, f) != 16) break;
            const int tag = fmt[0] | (fmt[1] << 8), channels = fmt[2] | (fmt[3] << 8), bits = fmt[14] | (fmt[15] << 8);
            if (tag != 1 || channels != 1 || bits != 16 || (int32_t) rd32(fmt + 4) != want_rate) {
                fprintf(stderr, "`%s` must be %d Hz, single-channel, 16-bit PCM\n", path, want_rate);
                exit(1);
            }
            have_fmt = 1;
            fseek(f, (long) (size - 16 + (size & 1)), SEEK_CUR);
        } else if (!memcmp(c, "data", 4) && have_fmt) {
            pcm = (int16_t *) malloc(size ? size : 2);
            *num_samples = (long) (fread(pcm, 1, size, f) / 2);
            break;
        } else {
            fseek(f, (long) (size + (size & 1)), SEEK_CUR);
        }
    }
    fclose(f);
    if (!pcm) {
        fprintf(stderr, "`%s` has no PCM data chunk\n", path);
        exit(1);
    }
    return pcm;
}

static void write_wav(const char *path, const int16_t *pcm, long n, int32_t rate) {
    FILE *f = fopen(path, "wb");
    if (!f) {
        fprintf(stderr, "cannot create `%s`\n", path);
        exit(1);
    }
    const uint32_t bytes = (uint32_t) n * 2, riff = 36 + bytes, byte_rate = (uint32_t) rate * 2;
    const unsigned char hdr[44] = {'R', 'I', 'F', 'F', riff, riff >> 8, riff >> 16, riff >> 24, 'W', 'A', 'V', 'E', 'f', 'm', 't', ' ',
                                   16, 0, 0, 0, 1, 0, 1, 0, rate, rate >> 8, rate >> 16, rate >> 24,
                                   byte_rate, byte_rate >> 8, byte_rate >> 16, byte_rate >> 24, 2, 0, 16, 0,
                                   'd', 'a', 't', 'a', bytes, bytes >> 8, bytes >> 16, bytes >> 24};
    fwrite(hdr, 1, 44, f);
    fwrite(pcm, 2, (size_t) n, f);
    fclose(f);
}

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}

int main(int argc, char **argv) {
    const char *lib = NULL, *model = NULL, *in = NULL, *out = NULL, *device = "best", *key = "koala-amd";
    int streams = 1, frames = 8;
    for (int i = 1; i < argc; ++i) {
        const char *v = i + 1 < argc ? argv[i + 1] : NULL;
        if (!strcmp(argv[i], "-l") && v) lib = v, ++i;
        else if (!strcmp(argv[i], "-m") && v) model = v, ++i;
        else if (!strcmp(argv[i], "-i") && v) in = v, ++i;
        else if (!strcmp(argv[i], "-o") && v) out = v, ++i;
        else if (!strcmp(argv[i], "-d") && v) device = v, ++i;
        else if (!strcmp(argv[i], "-a") && v) key = v, ++i;
        else if (!strcmp(argv[i], "--streams") && v) streams = atoi(v), ++i;
        else if (!strcmp(argv[i], "--frames") && v) frames = atoi(v), ++i;
        else {
            fprintf(stderr, "usage: %s -l LIBRARY -m MODEL -i IN.wav -o OUT.wav [-d DEVICE] [-a KEY] [--streams N --frames T]\n", argv[0]);
            return 2;
        }
    }
    if (!lib || !model || !in || !out || streams < 1 || frames < 1) {
        fprintf(stderr, "missing -l / -m / -i / -o\n");
        return 2;
    }
    api_t a;
    bind(&a, lib);
    const int32_t n = a.frame_length(), rate = a.sample_rate();
    printf("Koala version %s, %d Hz, %d samples per frame\n", a.version(), rate, n);
    long len = 0;
    int16_t *pcm = read_wav(in, rate, &len);

    int32_t delay = 0;
    long total = 0;  /* frames to push: the input plus `delay` samples of flush */
    int16_t *enh = NULL;
    double t0, busy;
    if (streams == 1) {
        void *k = NULL;
        status_t st = a.init(key, model, device, &k);
        if (st) die_with_stack(&a, "pv_koala_init", st);
        if ((st = a.delay_sample(k, &delay))) die_with_stack(&a, "pv_koala_delay_sample", st);
        total = (len + delay + n - 1) / n;
        enh = (int16_t *) calloc((size_t) total * n, 2);
        int16_t *frame = (int16_t *) calloc((size_t) n, 2);
        t0 = now_s();
        for (long f = 0; f < total; ++f) {
            memset(frame, 0, (size_t) n * 2);
            const long start = f * n, avail = len - start;
            if (avail > 0) memcpy(frame, pcm + start, (size_t) (avail < n ? avail : n) * 2);
            if ((st = a.process(k, frame, enh + start))) die_with_stack(&a, "pv_koala_process", st);
        }
        busy = now_s() - t0;
        free(frame);
        a.del(k);
    } else {
        void *k = NULL;
        status_t st = a.batch_init(key, model, device, streams, frames, 0 /* fp32 */, &k);
        if (st) die_with_stack(&a, "pv_koala_batch_init", st);
        delay = 256;
        total = (len + delay + n - 1) / n;
        total = (total + frames - 1) / frames * frames;
        enh = (int16_t *) calloc((size_t) total * n, 2);
        const size_t chunk = (size_t) frames * n;
        int16_t *bin = (int16_t *) calloc((size_t) streams * chunk, 2), *bout = (int16_t *) calloc((size_t) streams * chunk, 2);
        t0 = now_s();
        for (long f = 0; f < total; f += frames) {
            memset(bin, 0, (size_t) streams * chunk * 2);
            const long start = f * n, avail = len - start;
            if (avail > 0)
                for (int s = 0; s < streams; ++s) memcpy(bin + s * chunk, pcm + start, (size_t) (avail < (long) chunk ? avail : (long) chunk) * 2);
            if ((st = a.batch_process_chunk(k, frames, bin, bout))) die_with_stack(&a, "pv_koala_batch_process_chunk", st);
            memcpy(enh + start, bout + (size_t) (streams - 1) * chunk, chunk * 2);  /* the last stream's copy */
        }
        busy = now_s() - t0;
        free(bin);
        free(bout);
        a.batch_delete(k);
    }
    write_wav(out, enh + delay, len, rate);  /* delay compensated: output sample i belongs to input sample i */
    printf("%ld samples x %d stream(s); real time factor: %.5f\n", len, streams, busy / ((double) len / rate) / streams);
    free(enh);
    free(pcm);
    dlclose(a.lib);
    return 0;
}
