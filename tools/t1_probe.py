import sys, time, os
sys.path.insert(0,'/root/repo')
import numpy as np, torch
import koala_amd
from koala_amd import params
from koala_amd.workload import synth_streams
model = params.ensure_params('/root/repo/build/random_1234.kns', 'random', 1234)
for B in (4096, 2048, 512):
    x = torch.from_numpy(np.tile(synth_streams(64, 1, 1), (B // 64, 1))).cuda()
    y = torch.empty_like(x)
    for prec in ('fp32',):
        kb = koala_amd.create_batch('k', B, 1, prec, model_path=model, library_path=koala_amd.developer_library_path())
        kb.set_stream(torch.cuda.current_stream().cuda_stream)
        for _ in range(50): kb.process_device(1, x.data_ptr(), y.data_ptr())
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(500): kb.process_device(1, x.data_ptr(), y.data_ptr())
        torch.cuda.synchronize(); dt=time.perf_counter()-t0
        print('SMALL_MT=%s B=%d %s: %.1f us per frame step, %.2f M frames/s' % (os.environ.get('KOALA_AMD_SMALL_MT','16'), B, prec, dt/500*1e6, B*500/dt/1e6))
        kb.delete()
