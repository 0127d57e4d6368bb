"""Stage-by-stage comparison of the HIP engine with the CPU oracle (developer tool; run on a GPU box)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import koala_amd  # noqa: E402
from koala_amd import params  # noqa: E402
from oracle import oracle  # noqa: E402


def synth(B, T, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, T * 256)) * 3000.0
    x += 6000.0 * np.sin(2 * np.pi * 440.0 / 16000.0 * np.arange(T * 256))[None, :] * rng.uniform(0.2, 1.0, (B, 1))
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def main():
    os.environ['KOALA_AMD_DEBUG_TAPS'] = '1'  # multi-frame calls store no spectrum otherwise (developer library only)
    koala_amd.build_native()
    model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
    print('devices', koala_amd.available_devices())
    for prec, oprec in (('fp32', oracle.PREC_FP32), ('bf16', oracle.PREC_BF16)):
        for (B, T, calls) in ((19, 3, 2), (1, 1, 3), (40, 8, 1)):
            x = synth(B, T * calls, seed=B)
            kb = koala_amd.create_batch('key', B, T, prec, model_path=model, library_path=koala_amd.developer_library_path())
            orc = [oracle.Oracle(model, 1, oprec) for _ in range(B)]
            worst = {}
            for c in range(calls):
                xc = np.ascontiguousarray(x[:, c * T * 256:(c + 1) * T * 256])
                y = kb.process(xc)
                names = ('features', 'spectrum', 'mask') + (('embed',) if prec == 'fp32' else ())  # (bf16 folds the front-end away: no embedding)
                taps = {k: kb.debug_read(k, T) for k in names}
                hid = kb.debug_read('hidden', T)
                for b in range(B):
                    for t in range(T):
                        o, tp = orc[b].process_tap(xc[b, t * 256:(t + 1) * 256])
                        for k in names:
                            d = float(np.max(np.abs(taps[k][t, b] - tp[k])))
                            worst[k] = max(worst.get(k, 0.0), d)
                        d = int(np.max(np.abs(o.astype(int) - y[b, t * 256:(t + 1) * 256].astype(int))))
                        worst['pcm_lsb'] = max(worst.get('pcm_lsb', 0), d)
                    d = float(np.max(np.abs(hid[:, b] - tp['hidden'])))
                    worst['hidden'] = max(worst.get('hidden', 0.0), d)
            print(prec, 'B=%d T=%d calls=%d' % (B, T, calls), {k: (round(v, 8) if isinstance(v, float) else v) for k, v in worst.items()})
            kb.delete()
    # throughput smoke
    for prec in ('fp32', 'bf16'):
        B, T = 4096, 8
        kb = koala_amd.create_batch('key', B, T, prec, model_path=model, library_path=koala_amd.developer_library_path())
        x = synth(B, T, 7)
        kb.process(x)
        t0 = time.time()
        n = 3
        for _ in range(n):
            kb.process(x)
        dt = (time.time() - t0) / n
        kb.profile_enable(True)
        kb.process(x)
        print(prec, 'B=4096 T=8 host-path frames/s', B * T / dt, kb.profile_read())
        kb.delete()


if __name__ == '__main__':
    main()
