"""
Writes tests/golden/kns_v1_golden.npz: input frames and the oracle's int16 output for them, for the seeded-random and
the two gate parameter sets (fixture-calibrated `gate`, adaptive-floor `adaptive`) in both precision modes.  The vectors pin the KNS-v1 spec itself (oracle regressions, compiler
or libm drift on another host) and give the GPU parity tests a committed target that does not depend on running the
oracle.  Inputs: 48 frames of resources/audio_samples/test.wav starting at the first speech onset, the same span of
noise.wav, and their sum.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_wav, model_file  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    test, noise = load_wav('test.wav'), load_wav('noise.wav')
    a, n = 34 * 256, 48 * 256
    x = np.stack([test[a:a + n], noise[a:a + n], (test[a:a + n].astype(np.int32) + noise[a:a + n]).astype(np.int16)])
    out = {'pcm': x}
    for kind in ('random', 'gate', 'adaptive'):
        model = model_file(kind)
        for prec, name in ((oracle.PREC_FP32, 'fp32'), (oracle.PREC_BF16, 'bf16')):
            out['%s_%s' % (kind, name)] = oracle.Oracle(model, 3, prec).process(x)
    path = os.path.join(ROOT, 'tests', 'golden', 'kns_v1_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
