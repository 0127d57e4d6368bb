"""Developer sweep: per-kernel-class time vs frames per call (separates per-launch prologue from per-step cost)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import koala_amd  # noqa: E402
from koala_amd import params  # noqa: E402
from koala_amd.workload import synth_streams  # noqa: E402

model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
B = int(os.environ.get('SWEEP_B', 4096))
for T in [int(t) for t in os.environ.get('SWEEP_T', '1,4,16,32,64').split(',')]:
    base = synth_streams(64, T, seed=1)
    x = torch.from_numpy(np.tile(base, (B // 64, 1))).cuda()
    y = torch.empty_like(x)
    kb = koala_amd.create_batch('k', B, T, os.environ.get('SWEEP_PREC', 'bf16'), model_path=model,
                               library_path=os.environ.get('SWEEP_LIB'))
    kb.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        kb.process_device(T, x.data_ptr(), y.data_ptr())
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        kb.process_device(T, x.data_ptr(), y.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    kb.profile_enable(True)
    for _ in range(3):
        kb.process_device(T, x.data_ptr(), y.data_ptr())
    p = kb.profile_read()
    print('T=%d  %.3f ms/call  %.2f Mframes/s | ' % (T, dt * 1e3, B * T / dt / 1e6) +
          '  '.join('%s %.1f us' % (k, v['ms'] / max(1, v['launches']) * 1e3) for k, v in p.items()), flush=True)
    kb.delete()
