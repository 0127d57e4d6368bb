"""Developer tool: synchronous host calls at the bench shape under different sub-chunk schedules (KOALA_AMD_HOST_SCHED, developer library):
page-locked and pageable buffers, ms per call and M frames/s -- the A/B behind Engine::host_schedule (profiles/r05_host_sched.txt)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import koala_amd
from koala_amd import params
from koala_amd.workload import synth_streams
model = params.ensure_params('build/random_1234.kns', 'random', 1234)
B, T = 4096, 64
x = np.tile(synth_streams(64, T, seed=1), (B // 64, 1))
for sched in ('16', None, '4,8,12,16,12,8,4', '8,12,12,12,12,8', '6,10,16,16,10,6', '3,5,8,12,12,12,7,5'):
    if sched is None: os.environ.pop('KOALA_AMD_HOST_SCHED', None)
    else: os.environ['KOALA_AMD_HOST_SCHED'] = sched
    kb = koala_amd.create_batch('k', B, T, 'bf16', model_path=model, library_path=koala_amd.developer_library_path())
    pi, po = kb.alloc_host(T), kb.alloc_host(T); pi[:] = x
    out = np.empty_like(x)
    res = []
    for name, fn in (('page-locked', lambda: kb.process_into(pi, po)), ('pageable', lambda: kb.process_into(x, out))):
        for _ in range(3): fn()
        t0 = time.perf_counter(); n = 12
        for _ in range(n): fn()
        dt = (time.perf_counter() - t0) / n
        res.append('%s %.3f ms %.2f M' % (name, dt * 1e3, B * T / dt / 1e6))
    print('schedule %-22s %s' % (sched or 'default (6 8 12.. 8 6)', ' | '.join(res)), flush=True)
    kb.delete()
