"""How far apart do two valid implementations of the tolerance-specified bf16 configuration land on ONE model?  (CPU only.)
The oracle's bf16 mode is run twice over the soak workload -- plain, and as a "second implementation" whose transcendental results
and some GEMM outputs differ in their last bit (oracle/kns_oracle.h, kns_oracle_set_jitter) -- and the PCM distance is reported.
That distance is what the GPU-vs-oracle comparison of the same model shows (GPU: hardware 2^x / 1/x / log2 and an MFMA that sums
eight products before it rounds, profiles/r05_mfma_probe.txt): steep units amplify a flipped bf16 rounding, smooth ones do not.
    python tools/model_sensitivity.py [model.kns | kind ...]   (kinds: random adaptive gate)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_wav, model_file  # noqa: E402
from koala_amd.workload import synth_streams  # noqa: E402
from oracle import oracle  # noqa: E402


def distance(model, x, seeds=(11, 23)):
    """max |plain - jittered| in LSB and the share of samples within 1 LSB, over `seeds` second implementations"""
    oracle.set_jitter(0)
    ref = oracle.Oracle(model, x.shape[0], oracle.PREC_BF16).process(x)
    worst, within1, n = 0, 0, 0
    for seed in seeds:
        oracle.set_jitter(seed)
        y = oracle.Oracle(model, x.shape[0], oracle.PREC_BF16).process(x)
        d = np.abs(y.astype(np.int64) - ref.astype(np.int64))
        worst = max(worst, int(d.max()))
        within1 += int((d <= 1).sum())
        n += d.size
    oracle.set_jitter(0)
    return worst, 100.0 * within1 / n


def workload(streams=256, frames=120):
    """the soak's synthetic streams plus the reference WAVs (speech, noise, mixed) looped"""
    x = synth_streams(streams, frames, seed=5000)
    t, nz = load_wav('test.wav'), load_wav('noise.wav')
    n = frames * 256
    for i, w in enumerate((t, nz, (t.astype(int) + nz).astype(np.int16))):
        x[i] = np.resize(w[:len(w) // 256 * 256], n)
    return x


def main():
    args = sys.argv[1:] or ['random', 'adaptive']
    x = workload()
    for a in args:
        model = a if os.path.exists(a) else model_file(a)
        worst, w1 = distance(model, x)
        print('%-40s bf16, %d streams x %d frames: worst |plain - jittered| = %3d LSB, %.4f %% within 1 LSB' % (os.path.basename(model), x.shape[0], x.shape[1] // 256, worst, w1), flush=True)


if __name__ == '__main__':
    main()
