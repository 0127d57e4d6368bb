"""Developer tool (GPU box): where one workgroup of the fused quad GRU-layer kernel spends a block (s_memtime stamps, developer
library, KOALA_AMD_QUAD_DBG=<workgroup>).   python tools/quad_timing.py [workgroup] [T]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
os.environ['KOALA_AMD_QUAD'] = '1'
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
    for w in range(8):
        s = st[w, lo:hi]
        nxt = np.concatenate([s[1:, 0], s[-1:, 0]])
        d = [s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], nxt - s[:, 3]]
        if w < 4:
            print('x wave %d: stage(+tile-16 gates) %.0f  projection %.0f  file %.0f  request+barrier %.0f | re-polls per phase %.2f' % (
                (w,) + tuple(v[:-1].mean() for v in d) + ((s[:, 7].astype(np.int64) & 0xffff).mean(),)))
        else:
            print('h wave %d: image write+MFMA %.0f  gi+gates %.0f  pack+publish %.0f  barrier %.0f' % (
                (w - 4,) + tuple(v[:-1].mean() for v in d)))
    kb.delete()


if __name__ == '__main__':
    main()
