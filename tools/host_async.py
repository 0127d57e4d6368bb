"""Developer tool: host-pointer throughput of the bench shape with page-locked buffers -- synchronous calls against
pv_koala_batch_process_chunk_async with K buffer pairs in rotation (K = 2, 3), and that the asynchronous path returns the
synchronous path's samples."""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import koala_amd  # noqa: E402
from koala_amd import params  # noqa: E402
from koala_amd.workload import synth_streams  # noqa: E402

model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
B, T = int(os.environ.get('HP_B', 4096)), int(os.environ.get('HP_T', 64))
x = np.tile(synth_streams(64, T, seed=1), (B // 64, 1))
kb = koala_amd.create_batch('k', B, T, 'bf16', model_path=model)
pairs = [(kb.alloc_host(T), kb.alloc_host(T)) for _ in range(3)]
for a, _ in pairs:
    a[:] = x
want = [kb.process(x) for _ in range(3)]
kb.reset()
for n in range(3):
    kb.process_async(*pairs[n])
kb.synchronize()
print('asynchronous = synchronous, bit for bit:', all(np.array_equal(pairs[n][1], want[n]) for n in range(3)))
for K in (2, 3):
    for n in range(6):
        kb.process_async(*pairs[n % K])
    kb.synchronize()
    t0 = time.perf_counter()
    N = 30
    for n in range(N):
        kb.process_async(*pairs[n % K])
    kb.synchronize()
    dt = (time.perf_counter() - t0) / N
    print('asynchronous, %d buffer pairs in rotation: %.3f ms/call  %.2f Mframes/s' % (K, dt * 1e3, B * T / dt / 1e6))
t0 = time.perf_counter()
for n in range(8):
    kb.process_into(pairs[0][0], pairs[0][1])
dt = (time.perf_counter() - t0) / 8
print('synchronous, page-locked: %.3f ms/call  %.2f Mframes/s' % (dt * 1e3, B * T / dt / 1e6))
