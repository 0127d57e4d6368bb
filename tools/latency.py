"""BASELINE configs[4]: single stream, one 256-sample frame per call (pv_koala_process through the Python surface and
through raw ctypes), p50 / p99 per-frame latency on the GPU next to the CPU oracle's per-frame time."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import koala_amd  # noqa: E402
from koala_amd import params  # noqa: E402
from oracle import oracle  # noqa: E402


def wav(name):
    import wave
    with wave.open(os.path.join(ROOT, 'tests', 'golden', name)) as w:
        return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).copy()


def main():
    koala_amd.build_native()
    model = koala_amd.default_model_path()  # the packaged adaptive-floor gate
    pcm = wav('test.wav')
    n = len(pcm) // 256
    frames = [np.ascontiguousarray(pcm[i * 256:(i + 1) * 256]) for i in range(n)]
    res = {}
    for prec in ('fp32', 'bf16'):
        os.environ['KOALA_AMD_PRECISION'] = prec
        k = koala_amd.create('key', model_path=model, device='gpu:0', library_path=os.environ.get('LATENCY_LIB') or None)
        lib = k._library
        out = (C.c_short * 256)()
        for f in frames[:50]:
            lib.pv_koala_process(k._handle, f.ctypes.data_as(C.POINTER(C.c_short)), out)
        lat = []
        for rep in range(6):
            k.reset()
            for f in frames:
                t0 = time.perf_counter()
                lib.pv_koala_process(k._handle, f.ctypes.data_as(C.POINTER(C.c_short)), out)
                lat.append(time.perf_counter() - t0)
        lat = np.array(lat) * 1e6
        if os.environ.get('LATENCY_DETAIL'):  # where the slow frames are: percentiles, and their positions within a pass over the file
            slow = np.nonzero(lat > 1.1 * np.percentile(lat, 50))[0]
            print(prec, 'percentiles 50/90/95/99/99.9: %s' % ' '.join('%.1f' % np.percentile(lat, q) for q in (50, 90, 95, 99, 99.9)),
                  '| %d of %d frames above 1.1 x p50; positions mod %d: %s' % (len(slow), len(lat), n, sorted(set((slow % n).tolist()))[:40]))
        # the reference's perf loop (binding/python/test_koala_perf.py:42-58): all frames of test.wav through process()
        t0 = time.perf_counter()
        for f in frames:
            k.process(f)
        loop = time.perf_counter() - t0
        k.delete()
        res[prec] = {'p50_us': float(np.percentile(lat, 50)), 'p99_us': float(np.percentile(lat, 99)),
                     'mean_us': float(lat.mean()), 'frames_per_s': 1e6 / float(lat.mean()),
                     'rtf': float(lat.mean()) / 16000.0, 'python_process_loop_s_365_frames': loop}
    o = oracle.Oracle(model, 1)
    t0 = time.perf_counter()
    o.process(pcm[:n * 256], num_threads=1)
    res['cpu_oracle_1_thread'] = {'mean_us': (time.perf_counter() - t0) / n * 1e6}
    for kk, v in res.items():
        print(kk, {a: round(b, 4) for a, b in v.items()})


if __name__ == '__main__':
    main()
