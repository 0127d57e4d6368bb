"""BASELINE configs[1] (256 streams, fp32 mask network) through the batch ABI with device pointers: frames/s at T frames per
call (developer tool).   python tools/config1_time.py [T ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402,F401  (first: one HIP runtime in the process)

import koala_amd  # noqa: E402
from conftest import model_file, synth_streams  # noqa: E402


def main():
    model = model_file('random', 1234)
    B = int(os.environ.get('CONFIG1_STREAMS', '256'))
    lib = os.environ.get('CONFIG1_LIB') or None
    for T in [int(a) for a in sys.argv[1:]] or [32, 1]:
        x = torch.from_numpy(synth_streams(B, T, seed=1)).cuda()
        y = torch.zeros_like(x)
        kb = koala_amd.create_batch('key', B, T, 'fp32', model_path=model, library_path=lib) if lib else \
            koala_amd.create_batch('key', B, T, 'fp32', model_path=model)
        kb.set_stream(torch.cuda.current_stream().cuda_stream)
        for _ in range(10):
            kb.process_device(T, x.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        n = 100
        t0 = time.perf_counter()
        for _ in range(n):
            kb.process_device(T, x.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        kb.set_stream(0)
        kb.delete()
        print('fp32 B=%d T=%d: %.3f ms per call, %.2f M frames/s' % (B, T, dt * 1e3, B * T / dt / 1e6))


if __name__ == '__main__':
    main()
