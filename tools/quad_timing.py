"""Developer tool (GPU box): where one workgroup of the fused quad GRU-layer kernel spends a block (s_memtime stamps, developer
library, KOALA_AMD_QUAD_DBG=<workgroup>).   python tools/quad_timing.py [workgroup] [T]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import koala_amd  # noqa: E402
from koala_amd import params  # noqa: E402
from koala_amd.workload import synth_streams  # noqa: E402
from ctypes import c_int64  # noqa: E402


def main():
    wg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    B = 4096
    os.environ['KOALA_AMD_QUAD_DBG'] = str(wg)
    koala_amd.build_native()
    model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
    kb = koala_amd.create_batch('k', B, T, 'bf16', model_path=model, library_path=koala_amd.developer_library_path())
    x = synth_streams(64, T, seed=1)
    dx = torch.from_numpy(np.ascontiguousarray(np.tile(x, (B // 64, 1)))).cuda()
    dy = torch.empty_like(dx)
    kb.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(20):
        kb.process_device(T, dx.data_ptr(), dy.data_ptr())
    torch.cuda.synchronize()
    out = np.empty(8 * 4 * T * 8, np.float32)
    n = kb._lib.pv_koala_batch_debug_read(kb._handle, 5, out.ctypes.data, c_int64(out.size))
    assert n == out.size, n
    st = out.reshape(8, 4 * T, 8)
    lo, hi = 4 * 8, 4 * (T - 4)  # steady state: steps 8 .. T - 4
    total = st[:, :, :7][st[:, :, :7] >= 0].max()
    print('workgroup %d, T = %d: last stamp at %.0f ticks; steady-state blocks %d..%d' % (wg, T, total, lo, hi))
    step = np.diff(st[4, lo:hi:4, 0]).mean()
    print('ticks per step (h wave 0, block start to block start 4 blocks later): %.0f' % step)
    for w in range(4, 8):
        s = st[w, lo:hi]
        nxt = np.concatenate([s[1:, 0], s[-1:, 0]])
        d = [s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 5] - s[:, 3], nxt - s[:, 5]]
        print('h wave %d: wait(+deferred tile write) %.0f  MFMA %.0f  gi+gates %.0f  pack+publish %.0f  loop %.0f' % (
            (w - 4,) + tuple(v.mean() for v in d)))
        for m in range(4):
            print('     m=%d: wait %.0f  mfma %.0f  gates %.0f  publish %.0f' % (
                m, d[0][m::4].mean(), d[1][m::4].mean(), d[2][m::4].mean(), d[3][m::4].mean()))
    for w in range(0, 4):
        s = st[w, lo:hi]
        if w < 3:
            filed = s[:, 4] > 0
            print('x wave %d: wait %.0f  gather issue + mfma %.0f  epilogue+stage %.0f  check+file %.0f | block-to-block %.0f | '
                  're-polls per block %.2f' % (w, (s[:, 1] - s[:, 0]).mean(), (s[:, 2] - s[:, 1]).mean(), (s[:, 3] - s[:, 2]).mean(),
                                              (s[:, 4] - s[:, 3])[filed].mean(), np.diff(s[:, 1]).mean(), (s[:, 7].astype(np.int64) & 0xffff).mean()))
        else:
            print('x wave 3: mfma %.0f  epilogue+stage(+tile 16 projection) %.0f | block-to-block %.0f | tile-16 gates %.0f' % (
                (s[:, 2] - s[:, 1]).mean(), (s[:, 3] - s[:, 2]).mean(), np.diff(s[:, 1]).mean(),
                np.mean([v for v in (s[:, 5] - s[:, 4]) if v > 0])))
    groups = [('XW', 0, 4), ('GI', 4, 8), ('GC', 8, 12), ('HL', 12, 16), ('HG', 16, 20), ('HM', 20, 24), ('GH16', 24, 27),
              ('G16C', 27, 28), ('H16', 28, 29)]
    for w in (0, 4, 7):
        masks = st[w, lo:hi, 6].astype(np.int64) + ((st[w, lo:hi, 7].astype(np.int64) >> 16) << 16)
        txt = []
        for nm, a, b_ in groups:
            for i in range(a, b_):
                frac = float(((masks >> i) & 1).mean())
                if frac > 0.02:
                    txt.append('%s[%d] %.0f%%' % (nm, i - a, 100 * frac))
        print('wave %d: counters behind at the first look of a block\'s wait: %s' % (w, ', '.join(txt) or 'none'))
    # lead of the x waves over the h waves: block index of x wave 0 when h wave 0 starts block b
    hx = st[4, lo:hi, 1]
    xs = st[0, :, 3]
    lead = [np.searchsorted(xs[xs >= 0], t) - (lo + i) for i, t in enumerate(hx)]
    print('x wave 0 lead over h wave 0 (blocks): mean %.2f min %d max %d' % (np.mean(lead), min(lead), max(lead)))
    kb.delete()


if __name__ == '__main__':
    main()
