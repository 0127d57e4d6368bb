"""Developer tool (through gpurun): what a pv_koala_init handle costs -- creation time and device bytes of the 1st ... 64th handle on one
model (round 6: the packed weights are one shared, ref-counted device image per (model, device, precision)).
    python tools/handles.py [fp32|bf16] > profiles/r06_handles.txt"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import koala_amd  # noqa: E402
from koala_amd import params  # noqa: E402

torch.cuda.init()
model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
frame = np.zeros(256, np.int16)
for prec in sys.argv[1:] or ['fp32', 'bf16']:
    os.environ['KOALA_AMD_PRECISION'] = prec
    for cache in (True, False):
        lib = koala_amd.default_library_path() if cache else koala_amd.developer_library_path()
        if not cache:
            os.environ['KOALA_AMD_NO_WEIGHT_CACHE'] = '1'
        hs, ms, mib = [], [], []
        for i in range(64):
            torch.cuda.synchronize()
            f0 = torch.cuda.mem_get_info()[0]
            t0 = time.perf_counter()
            h = koala_amd.create('k', model_path=model, device='gpu:0', library_path=lib)
            ms.append((time.perf_counter() - t0) * 1e3)
            torch.cuda.synchronize()
            mib.append((f0 - torch.cuda.mem_get_info()[0]) / 2.0 ** 20)
            hs.append(h)
        t0 = time.perf_counter()
        for _ in range(20):
            for h in hs:
                h.process(frame)
        per_frame = (time.perf_counter() - t0) / (20 * 64) * 1e6
        for h in hs:
            h.delete()
        os.environ.pop('KOALA_AMD_NO_WEIGHT_CACHE', None)
        print('%s, %s: pv_koala_init ms  1st %.1f  2nd %.1f  64th %.1f  (mean 2..64 %.1f) | device MiB  1st %.1f  2nd %.1f  64th %.1f  '
              '(sum of 64: %.0f) | %.1f us per frame round-robin over the 64 handles from one thread'
              % (prec, 'shared weight image' if cache else 'one image per handle (rounds 1-5; developer switch)', ms[0], ms[1], ms[63],
                 float(np.mean(ms[1:])), mib[0], mib[1], mib[63], float(np.sum(mib)), per_frame))
