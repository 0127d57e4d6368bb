"""Developer tool (GPU box): where a workgroup of the one-step quad kernel (gru_quad1_kernel, last GRU layer of a one-frame call)
spends its time: s_memtime stamps of all 8 waves (developer library, KOALA_AMD_QUAD_DBG=<workgroup>).  python tools/t1_stamps.py [wg]"""
import os
import sys
from ctypes import c_int64

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import koala_amd  # noqa: E402
from koala_amd import params  # noqa: E402
from koala_amd.workload import synth_streams  # noqa: E402


def main():
    wg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    os.environ['KOALA_AMD_QUAD_DBG'] = str(wg)
    model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
    B = 4096
    kb = koala_amd.create_batch('k', B, 1, 'bf16', model_path=model, library_path=koala_amd.developer_library_path())
    dx = torch.from_numpy(np.ascontiguousarray(np.tile(synth_streams(64, 1, seed=1), (B // 64, 1)))).cuda()
    dy = torch.empty_like(dx)
    kb.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(50):
        kb.process_device(1, dx.data_ptr(), dy.data_ptr())
    torch.cuda.synchronize()
    out = np.empty(8 * 4 * 8, np.float32)
    n = kb._lib.pv_koala_batch_debug_read(kb._handle, 5, out.ctypes.data, c_int64(out.size))
    assert n == out.size, n
    st = out.reshape(8, 4, 8)
    names = ['start', 'loads requested', 'staging in LDS', 'barrier A passed', 'MFMAs done', 'barrier B passed']
    print('workgroup %d, ticks (100 MHz? see below) since the first stamp; x waves 0-3, h waves 4-7' % wg)
    for i, nm in enumerate(names):
        print('%-18s ' % nm + ' '.join('%7.0f' % st[w, 0, i] for w in range(8)))
    print('%-18s ' % 'gates + stores out' + ' '.join('%7.0f' % st[w, 1, 0] for w in range(8)))
    kb.delete()


if __name__ == '__main__':
    main()
