"""Developer tool: frames/s of the engine over batch sizes and frames per call, both precisions, with the route each call took
(device-resident PCM, developer library for the route query).   python tools/batch_sweep.py > gpurun_out/batch_sweep.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402,F401

import koala_amd  # noqa: E402
from koala_amd._util import developer_library_path  # noqa: E402
from conftest import model_file, synth_streams  # noqa: E402

ROUTES = {0: 'chunked', 1: 'layer kernel', 2: 'layer kernel, frame by frame', 3: 'quad kernel', 4: 'wavefront'}


def main():
    model = model_file('random', 1234)
    print('# frames/s over batch size and frames per call (random-weight KNS-v1 model, device-resident PCM, one MI355X); route = the kernel '
          'family the GRU layers took (kns_engine.cpp, dispatch table)')
    for prec in ('bf16', 'fp32'):
        for T in (1, 8, 32, 64):
            for B in (1, 16, 64, 256, 512, 1024, 2048, 3072, 4096, 8192):
                if prec == 'fp32' and B > 4096:
                    continue
                base = synth_streams(min(B, 64), T, seed=5)
                x = torch.from_numpy(np.tile(base, ((B + 63) // 64, 1))[:B].copy()).cuda()
                y = torch.zeros_like(x)
                kb = koala_amd.create_batch('key', B, T, prec, model_path=model, library_path=developer_library_path())
                kb.set_stream(torch.cuda.current_stream().cuda_stream)
                for _ in range(5):
                    kb.process_device(T, x.data_ptr(), y.data_ptr())
                torch.cuda.synchronize()
                reps = 200 if B * T <= 4096 else 40 if B * T <= 65536 else 12
                t0 = time.perf_counter()
                for _ in range(reps):
                    kb.process_device(T, x.data_ptr(), y.data_ptr())
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / reps
                route = int(kb.debug_read('route', T)[0])
                kb.set_stream(0)
                kb.delete()
                print('%s  %5d streams x %2d frames: %8.3f ms per call  %8.3f M frames/s  (%s)' % (prec, B, T, dt * 1e3, B * T / dt / 1e6, ROUTES.get(route, route)),
                      flush=True)


if __name__ == '__main__':
    main()
