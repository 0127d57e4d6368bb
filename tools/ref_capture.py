"""
For a holder of a Picovoice AccessKey (needs network for the licence check): records what the REFERENCE engine does
on its own fixtures, so that sample-level parity with `pv_koala_process` stops being "unpinned" (SURVEY.md 8c / 8f-3).

    python tools/ref_capture.py --access-key KEY --reference /path/to/koala/checkout --out tests/golden/reference_capture.npz

It drives the reference's shipped library through plain ctypes (nothing of the reference's Python is imported),
one stream, device cpu:1, over resources/audio_samples/test.wav, noise.wav and their sample-wise sum, and stores
delay_sample plus every output frame.  tests/ can then compare koala_amd (run with an imported .pv model once
tools exist for that) or, today, check the energy envelope of the capture against the same envelope of KNS-v1.
Cannot run in the build container: no key, no network.
"""
import argparse
import ctypes as C
import os
import wave

import numpy as np


def wav(path):
    with wave.open(path) as w:
        return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--access-key', required=True)
    ap.add_argument('--reference', required=True, help='root of a Picovoice/koala checkout')
    ap.add_argument('--device', default='cpu:1')
    ap.add_argument('--out', required=True)
    a = ap.parse_args()
    lib = C.CDLL(os.path.join(a.reference, 'lib/linux/x86_64/libpv_koala.so'))
    lib.pv_koala_init.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
    lib.pv_koala_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pv_koala_delay_sample.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.pv_koala_delete.argtypes = [C.c_void_p]
    lib.pv_koala_version.restype = C.c_char_p
    model = os.path.join(a.reference, 'lib/common/koala_params.pv').encode()
    test = wav(os.path.join(a.reference, 'resources/audio_samples/test.wav'))
    noise = wav(os.path.join(a.reference, 'resources/audio_samples/noise.wav'))
    inputs = {'test': test, 'noise': noise, 'mixed': (test.astype(np.int32) + noise).astype(np.int16)}
    res = {}
    for name, pcm in inputs.items():
        h = C.c_void_p()
        st = lib.pv_koala_init(a.access_key.encode(), model, a.device.encode(), C.byref(h))
        if st != 0:
            raise SystemExit('pv_koala_init failed with status %d' % st)
        d = C.c_int32()
        lib.pv_koala_delay_sample(h, C.byref(d))
        n = len(pcm) // 256
        out = np.zeros(n * 256, np.int16)
        for i in range(n):
            frame = np.ascontiguousarray(pcm[i * 256:(i + 1) * 256])
            st = lib.pv_koala_process(h, frame.ctypes.data, out[i * 256:].ctypes.data)
            if st != 0:
                raise SystemExit('pv_koala_process failed with status %d' % st)
        lib.pv_koala_delete(h)
        res['in_' + name] = pcm[:n * 256]
        res['out_' + name] = out
        res['delay_sample'] = np.int32(d.value)
    res['version'] = np.bytes_(lib.pv_koala_version())
    res['device'] = np.bytes_(a.device.encode())
    np.savez_compressed(a.out, **res)
    print('wrote', a.out, 'delay_sample', int(res['delay_sample']))


if __name__ == '__main__':
    main()
