"""Synthetic 16 kHz mono streams for benchmarks and tests: speech-like AR(2)-coloured Gaussian noise
with 4 Hz amplitude modulation plus white noise.  Stream s depends only on (seed, s), so a shard can generate its
own rows."""

import numpy as np


def synth_streams(num_streams: int, num_frames: int, seed: int = 1234, first_stream: int = 0) -> np.ndarray:
    from scipy.signal import lfilter

    n = num_frames * 256
    t = np.arange(n) / 16000.0
    out = np.empty((num_streams, n), np.int16)
    for i in range(num_streams):
        rng = np.random.default_rng(seed + first_stream + i)
        x = lfilter([1.0], [1.0, -1.8, 0.9], rng.standard_normal(n))  # resonance near 500 Hz
        x /= np.std(x) + 1e-9
        env = 0.55 + 0.45 * np.sin(2 * np.pi * 4.0 * t + rng.uniform(0, 6.28))
        sig = 2000.0 * x * env + 600.0 * rng.standard_normal(n)
        out[i] = np.clip(np.rint(sig), -32768, 32767).astype(np.int16)
    return out


__all__ = ['synth_streams']
