"""Developer tool: PCIe-inclusive throughput of host-pointer calls (pageable vs page-locked buffers, with and without
sub-chunk pipelining) next to the device-pointer rate of the same workload."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import koala_amd  # noqa: E402
from koala_amd import params  # noqa: E402
from koala_amd.workload import synth_streams  # noqa: E402

model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
B = int(os.environ.get('HP_B', 4096))
T = int(os.environ.get('HP_T', 32))
x = np.tile(synth_streams(64, T, seed=1), (B // 64, 1))
for chunk in (None, '0'):
    if chunk is None:
        os.environ.pop('KOALA_AMD_HOST_CHUNK', None)
    else:
        os.environ['KOALA_AMD_HOST_CHUNK'] = chunk
    kb = koala_amd.create_batch('k', B, T, 'bf16', model_path=model, library_path=koala_amd.developer_library_path())
    pin_in, pin_out = kb.alloc_host(T), kb.alloc_host(T)
    pin_in[:] = x
    out = np.empty_like(x)
    dx = torch.from_numpy(x).cuda()
    dy = torch.empty_like(dx)
    cases = {'pageable': lambda: kb.process_into(x, out), 'page-locked': lambda: kb.process_into(pin_in, pin_out),
             'device': lambda: (kb.process_device(T, dx.data_ptr(), dy.data_ptr()), kb.synchronize())}
    for name, fn in cases.items():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.5:
            fn()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        dt = (time.perf_counter() - t0) / n
        print('sub-chunks %-9s %-12s %.3f ms/call  %.2f Mframes/s' % ('off' if chunk == '0' else 'on', name, dt * 1e3, B * T / dt / 1e6),
              flush=True)
    kb.delete()
