"""
Writes koala_amd/data/noise_prior.npy: mean log-power per STFT bin of tests/golden/noise.wav (the reference's
resources/audio_samples/noise.wav) under KNS-v1's own analysis (sqrt-Hann 512/256, samples / 32768).
It is the stationary-noise prior that params.make_gate() turns into the per-bin gate threshold.
"""
import os
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    with wave.open(os.path.join(ROOT, 'tests', 'golden', 'noise.wav')) as w:
        x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float64) / 32768.0
    n = len(x) // 256 * 256
    win = np.sin(np.pi * np.arange(512) / 512)
    frames = np.stack([x[i:i + 512] * win for i in range(0, n - 512 + 1, 256)])
    logp = np.log(np.abs(np.fft.rfft(frames, axis=1)) ** 2 + 1e-10)
    out = os.path.join(ROOT, 'koala_amd', 'data', 'noise_prior.npy')
    np.save(out, logp.mean(0).astype(np.float32))
    print('wrote', out, logp.mean(0)[:4])


if __name__ == '__main__':
    main()
