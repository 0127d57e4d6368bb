"""fp32 engine vs oracle, bit for bit: counts of differing values per tap and of differing PCM samples (developer tool).
usage: python tools/exact_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['KOALA_AMD_DEBUG_TAPS'] = '1'
import koala_amd  # noqa: E402
from conftest import load_wav, model_file, synth_streams  # noqa: E402
from oracle import oracle  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def main():
    model = model_file('random', 1234)
    dev = koala_amd.developer_library_path()
    for B, T, calls in ((1, 1, 6), (19, 3, 3), (40, 8, 2)):
        x = synth_streams(B, T * calls, seed=100 + B)
        kb = koala_amd.create_batch('key', B, T, 'fp32', model_path=model, library_path=dev)
        streams = [oracle.Oracle(model) for _ in range(B)]
        nd = {k: 0 for k in ('spectrum', 'features', 'embed', 'mask', 'hidden', 'pcm')}
        mx = dict(nd)
        for c in range(calls):
            xc = np.ascontiguousarray(x[:, c * T * 256:(c + 1) * T * 256])
            y = kb.process(xc)
            taps = {k: kb.debug_read(k, T) for k in ('spectrum', 'features', 'embed', 'mask')}
            hidden = kb.debug_read('hidden', T)
            for b in range(B):
                for t in range(T):
                    ref, tp = streams[b].process_tap(xc[b, t * 256:(t + 1) * 256])
                    for k in taps:
                        g, w = taps[k][t, b], tp[k]
                        ne = (bits(g) != bits(w)) & ~((g == 0) & (w == 0))
                        nd[k] += int(ne.sum())
                        mx[k] = max(mx[k], float(np.abs(g - w).max()))
                    d = np.abs(ref.astype(int) - y[b, t * 256:(t + 1) * 256])
                    nd['pcm'] += int((d != 0).sum())
                    mx['pcm'] = max(mx['pcm'], int(d.max()))
                nd['hidden'] += int((bits(hidden[:, b]) != bits(tp['hidden'])).sum())
        kb.delete()
        print('B=%d T=%d calls=%d differing values:' % (B, T, calls), nd, 'max abs:', mx)
    # the reference WAVs through the single-stream ABI, and a long run (drift)
    test, noise = load_wav('test.wav'), load_wav('noise.wav')
    for name, pcm in (('test', test), ('noise', noise), ('mixed', (test.astype(int) + noise).astype(np.int16))):
        n = len(pcm) // 256 * 256
        for kind in ('adaptive', 'random'):
            m = model_file(kind)
            k = koala_amd.create('key', model_path=m, device='gpu:0')
            out = np.concatenate([np.array(k.process(pcm[i:i + 256]), np.int16) for i in range(0, n, 256)])
            k.delete()
            ref = oracle.Oracle(m, 1).process(pcm[None, :n])[0]
            d = np.abs(out.astype(int) - ref)
            print('%s / %s: %d frames, differing samples %d, max %d' % (name, kind, n // 256, int((d != 0).sum()), int(d.max())))
    imp = os.path.join(ROOT, 'build', 'imported_pv_default.kns')
    if os.path.exists(imp):
        n = len(test) // 256 * 256
        kb = koala_amd.create_batch('key', 1, 73, 'fp32', model_path=imp)
        out = np.concatenate([kb.process(np.ascontiguousarray(test[None, i:i + 73 * 256])) for i in range(0, n, 73 * 256)], axis=1)[0]
        kb.delete()
        ref = oracle.Oracle(imp, 1).process(test[None, :n])[0]
        d = np.abs(out.astype(int) - ref)
        print('imported reference model, test.wav: differing samples %d, max %d' % (int((d != 0).sum()), int(d.max())))


if __name__ == '__main__':
    main()
