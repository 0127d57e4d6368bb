"""Long random call sequences against the oracle (developer tool, GPU box): chunk lengths 1 .. Tmax, host and device pointers, in-place
calls, masked and full resets, three models (random, five-frame front-end, the default adaptive gate), both precisions.  The suite's
soak tests run a few dozen calls per case; this runs hundreds.   python tools/soak.py [calls-scale]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import koala_amd  # noqa: E402
from conftest import model_file  # noqa: E402
from koala_amd.workload import synth_streams  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    rng = np.random.default_rng(7)
    cases = [('random', 'bf16', 4096, 8, 60), ('random', 'fp32', 1000, 8, 40), ('random', 'bf16', 300, 16, 80), ('random5', 'bf16', 1030, 6, 60),
             ('random5', 'fp32', 77, 9, 60), ('adaptive', 'bf16', 2048, 4, 40), ('adaptive', 'fp32', 256, 32, 12), ('random', 'bf16', 17, 5, 200),
             ('random', 'fp32', 16, 3, 200), ('random', 'fp32', 200, 12, 60), ('random5', 'bf16', 100, 9, 60), ('random', 'bf16', 500, 40, 30),
             # mid-size batches: calls of 32 frames and more take the layer pipeline over sub-chunks (long ones with sliced STFT stages), shorter
             # ones the plain route -- state handed back and forth between the two, with resets
             ('random', 'bf16', 1200, 64, 24), ('adaptive', 'bf16', 1024, 56, 16), ('random5', 'bf16', 2000, 40, 12)]
    only = os.environ.get('SOAK_ONLY')  # e.g. adaptive:bf16
    if only:
        cases = [c for c in cases if '%s:%s' % (c[0], c[1]) == only]
    bad = 0
    for kind, precision, B, Tmax, calls in cases:
        calls = max(4, int(calls * scale))
        model = os.environ.get('SOAK_MODEL') or model_file(kind)  # (SOAK_MODEL: a candidate parameter file in place of the case's)
        kb = koala_amd.create_batch('key', B, Tmax, precision, model_path=model)
        ref = oracle.Oracle(model, B, oracle.PREC_BF16 if precision == 'bf16' else oracle.PREC_FP32)
        worst, within1, n = 0, 0, 0
        for call in range(calls):
            T = int(rng.integers(1, Tmax + 1))
            x = synth_streams(B, T, seed=5000 + call)
            r = rng.random()
            if r < 0.15:
                m = (rng.random(B) < 0.4).astype(np.uint8)
                kb.reset(m)
                ref.reset(m)
            elif r < 0.2:
                kb.reset()
                ref.reset()
            mode = rng.random()
            if mode < 0.3:  # device pointers, separate buffers
                dx = torch.from_numpy(x).cuda()
                dy = torch.zeros_like(dx)
                torch.cuda.synchronize()
                kb.process_device(T, dx.data_ptr(), dy.data_ptr())
                kb.synchronize()
                y = dy.cpu().numpy()
            elif mode < 0.45:  # device pointers, in place
                dx = torch.from_numpy(x).cuda()
                torch.cuda.synchronize()
                kb.process_device(T, dx.data_ptr(), dx.data_ptr())
                kb.synchronize()
                y = dx.cpu().numpy()
            elif mode < 0.6:  # host pointers, in place
                y = x.copy()
                kb.process_into(y, y)
            else:
                y = kb.process(x)
            d = np.abs(y.astype(np.int64) - ref.process(x).astype(np.int64))
            worst = max(worst, int(d.max()))
            within1 += int((d <= 1).sum())
            n += d.size
        tol = 5 if precision == 'bf16' else 0
        ok = worst <= tol and (precision != 'bf16' or within1 / n >= 0.999)
        bad += not ok
        print('%-8s %s B=%-5d Tmax=%-3d %4d calls: worst |gpu - oracle| = %d LSB, %.4f %% within 1 LSB  %s'
              % (kind, precision, B, Tmax, calls, worst, 100.0 * within1 / n, 'ok' if ok else 'FAIL'), flush=True)
        kb.delete()
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
