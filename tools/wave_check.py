"""Developer tool: the wavefront route of multi-frame calls (kns_engine.cpp, kRouteWave) against the layer-by-layer routes --
same PCM bit for bit, and the time per call.   python tools/wave_check.py [streams ...]      (developer library)
WAVE_MT: m-tile limit handed to the wavefront arm (default 4096 = always); WAVE_T: frames per call (default 32)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402,F401

import koala_amd  # noqa: E402
from koala_amd._util import developer_library_path  # noqa: E402
from conftest import model_file, synth_streams  # noqa: E402


def run(B, T, prec, wave_mt, calls=3, reps=50):
    os.environ['KOALA_AMD_WAVE_MT'] = str(wave_mt)
    model = model_file('random', 1234)
    x = [torch.from_numpy(synth_streams(B, T, seed=10 + c)).cuda() for c in range(calls)]
    y = [torch.zeros_like(x[0]) for _ in range(calls)]
    kb = koala_amd.create_batch('key', B, T, prec, model_path=model, library_path=os.environ.get('WAVE_LIB') or developer_library_path())
    kb.set_stream(torch.cuda.current_stream().cuda_stream)
    for c in range(calls):
        kb.process_device(T, x[c].data_ptr(), y[c].data_ptr())
    torch.cuda.synchronize()
    out = np.concatenate([v.cpu().numpy() for v in y], axis=1)
    scratch = torch.zeros_like(x[0])
    for _ in range(5):
        kb.process_device(T, x[0].data_ptr(), scratch.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        kb.process_device(T, x[0].data_ptr(), scratch.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    kb.set_stream(0)
    kb.delete()
    return out, dt


def main():
    T = int(os.environ.get('WAVE_T', '32'))
    wave_mt = int(os.environ.get('WAVE_MT', '4096'))
    for B in [int(a) for a in sys.argv[1:]] or [256]:
        for prec in ('fp32', 'bf16'):
            ref, t_ref = run(B, T, prec, 0)
            got, t_got = run(B, T, prec, wave_mt)
            d = np.abs(ref.astype(np.int32) - got.astype(np.int32))
            print('%s B=%d T=%d: layer by layer %.3f ms (%.2f M frames/s) | wavefront %.3f ms (%.2f M frames/s) | max |diff| %d, '
                  'differing samples %d of %d' % (prec, B, T, t_ref * 1e3, B * T / t_ref / 1e6, t_got * 1e3, B * T / t_got / 1e6,
                                                 d.max(), int((d != 0).sum()), d.size), flush=True)


if __name__ == '__main__':
    main()
