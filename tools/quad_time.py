"""Developer tool (GPU box): ms/step and per-class times of the bench shape for a list of libraries (fused quad path on).
   python tools/quad_time.py lib1.so lib2.so ..."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
os.environ['KOALA_AMD_QUAD'] = '1'
import koala_amd
from koala_amd import params
from koala_amd.workload import synth_streams
model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
B, T = 4096, int(os.environ.get('QT_T', '64'))
x = synth_streams(64, T, seed=1)
dx = torch.from_numpy(np.ascontiguousarray(np.tile(x, (B // 64, 1)))).cuda()
dy = torch.empty_like(dx)
for lib in sys.argv[1:]:
    kb = koala_amd.create_batch('k', B, T, 'bf16', model_path=model, library_path=os.path.abspath(lib))
    kb.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        for _ in range(20):
            kb.process_device(T, dx.data_ptr(), dy.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(60):
            kb.process_device(T, dx.data_ptr(), dy.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 60
        kb.profile_enable(True)
        for _ in range(10):
            kb.process_device(T, dx.data_ptr(), dy.data_ptr())
        prof = kb.profile_read()
        kb.profile_enable(False)
        err = ''
        try:
            kb.synchronize()
        except Exception as e:
            err = ' ERROR %s' % e
        print('%-28s %.3f ms/step | gru %.1f us x%d%s' % (os.path.basename(lib), dt * 1e3, prof['gru_recurrent']['ms'] / max(1, prof['gru_recurrent']['launches']) * 1e3, prof['gru_recurrent']['launches'] // 10, err))
    except Exception as e:
        print(os.path.basename(lib), 'FAILED', e)
    kb.delete()
