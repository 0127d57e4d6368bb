"""Developer tool (through gpurun): the bound on cross-call weight residency for one-frame calls -- the frame step
of 4 096 / 8 192 streams with the product's one-step quad kernel and with a TIMING variant whose weights are never fetched
(-DQ1_T_NOFETCH, garbage results: as if W_ih / W_hh were already on the CU), per kernel class.
    tools/variant_lib.sh q1nofetch "-DQ1_T_NOFETCH" kns_gruq;  python tools/t1_bound.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import koala_amd  # noqa: E402
from koala_amd import params  # noqa: E402
from koala_amd.workload import synth_streams  # noqa: E402

model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
s = torch.cuda.Stream()
torch.cuda.set_stream(s)
for B in (4096, 8192):
    x = torch.from_numpy(np.tile(synth_streams(64, 1, 1), (B // 64, 1)).copy()).cuda()
    y = torch.empty_like(x)
    for name, lib in (('product kernels', koala_amd.developer_library_path()), ('weights never fetched (timing only)', os.path.join(ROOT, 'build/ab/libq1nofetch.so'))):
        kb = koala_amd.create_batch('k', B, 1, 'bf16', model_path=model, library_path=lib)
        kb.set_stream(s.cuda_stream)
        for _ in range(100):
            kb.process_device(1, x.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1000):
            kb.process_device(1, x.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 1000
        kb.profile_enable(True)
        for _ in range(100):
            kb.process_device(1, x.data_ptr(), y.data_ptr())
        pr = kb.profile_read()
        kb.delete()
        print('B=%d %-38s %6.1f us per frame step = %6.2f M frames/s | per class, us per step (HIP events): %s' % (
            B, name, dt * 1e6, B / dt / 1e6, '  '.join('%s %.1f (%d launches)' % (k, v['ms'] / 100 * 1e3, v['launches'] // 100) for k, v in pr.items() if v['launches'])))
