"""Developer tool (GPU box): the fused quad GRU-layer kernel (kns_gruq.hip) against the two-kernel form, bit for bit, on a few
shapes, and their per-class times at the bench shape.   python tools/quad_check.py [quick]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (first: see tests/conftest.py)
import koala_amd  # noqa: E402
from koala_amd import params  # noqa: E402
from koala_amd.workload import synth_streams  # noqa: E402

DEV = os.environ.get("QUAD_LIB") or koala_amd.developer_library_path()


def make(B, T, quad, model):
    os.environ.pop('KOALA_AMD_QUAD', None)
    os.environ.pop('KOALA_AMD_NO_QUAD', None)
    os.environ['KOALA_AMD_QUAD' if quad else 'KOALA_AMD_NO_QUAD'] = '1'
    kb = koala_amd.create_batch('k', B, T, 'bf16', model_path=model, library_path=DEV)
    os.environ.pop('KOALA_AMD_QUAD', None)
    os.environ.pop('KOALA_AMD_NO_QUAD', None)
    return kb


def main():
    koala_amd.build_native()
    model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
    shapes = [(64, 2, 2), (64, 7, 2), (320, 5, 2), (1024, 16, 1), (4096, 8, 2), (4096, 1, 3)]
    if len(sys.argv) > 1 and sys.argv[1] == 'quick':
        shapes = shapes[:2]
    ok = True
    for B, T, calls in shapes:
        x = synth_streams(B, T * calls, seed=B + T)
        outs = []
        for quad in (False, True):
            kb = make(B, T, quad, model)
            try:
                y = [kb.process(np.ascontiguousarray(x[:, c * T * 256:(c + 1) * T * 256])) for c in range(calls)]
                outs.append(np.concatenate(y, axis=1))
            except Exception as e:
                print('B=%d T=%d quad=%s FAILED: %s' % (B, T, quad, e))
                outs.append(None)
            kb.delete()
        if outs[0] is None or outs[1] is None:
            ok = False
            continue
        d = np.abs(outs[0].astype(int) - outs[1].astype(int))
        print('B=%d T=%d calls=%d: max |two-kernel - fused| = %d LSB, differing samples %d of %d' % (B, T, calls, d.max(), (d > 0).sum(), d.size))
        ok = ok and d.max() == 0
    # timing at the bench shape
    B, T = 4096, 64
    x = synth_streams(64, T, seed=1)
    dx = torch.from_numpy(np.ascontiguousarray(np.tile(x, (B // 64, 1)))).cuda()
    dy = torch.empty_like(dx)
    for quad in (False, True):
        kb = make(B, T, quad, model)
        kb.set_stream(torch.cuda.current_stream().cuda_stream)
        for _ in range(30):
            kb.process_device(T, dx.data_ptr(), dy.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 100
        for _ in range(n):
            kb.process_device(T, dx.data_ptr(), dy.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        kb.profile_enable(True)
        for _ in range(10):
            kb.process_device(T, dx.data_ptr(), dy.data_ptr())
        prof = kb.profile_read()
        kb.profile_enable(False)
        try:
            kb.synchronize()
            err = ''
        except Exception as e:
            err = ' ERROR: %s' % e
        print('quad=%s: %.3f ms/step = %.1f M frames/s | ' % (quad, dt * 1e3, B * T / dt / 1e6) +
              '  '.join('%s %.1f us x%d' % (k, v['ms'] / max(v['launches'], 1) * 1e3, v['launches'] // 10) for k, v in prof.items()) + err)
        kb.delete()
    print('IDENTICAL' if ok else 'MISMATCH')


if __name__ == '__main__':
    main()
