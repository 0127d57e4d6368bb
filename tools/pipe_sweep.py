"""Developer tool (through gpurun): mid-size batches with and without the layer pipeline over sub-chunks of frames (kns_engine.cpp run_device,
kRoutePipelined): ms per call and frames/s, bf16, device-resident PCM.   python tools/pipe_sweep.py > gpurun_out/pipe_sweep.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import koala_amd  # noqa: E402
from koala_amd import params  # noqa: E402
from koala_amd.workload import synth_streams  # noqa: E402

model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
arms = [('one launch per layer (rounds 1-5)', {'KOALA_AMD_PIPE_MT': '0'}), ('pipelined, 16 frames per sub-chunk', {'KOALA_AMD_PIPE_MT': '4096'}),
        ('pipelined, 8', {'KOALA_AMD_PIPE_MT': '4096', 'KOALA_AMD_PIPE_CHUNK': '8'}), ('pipelined, 32', {'KOALA_AMD_PIPE_MT': '4096', 'KOALA_AMD_PIPE_CHUNK': '32'}),
        ('pipelined, whole-call STFT launches', {'KOALA_AMD_PIPE_MT': '144', 'KOALA_AMD_PIPE_WHOLE_STFT': '1'}), ('product default', {})]
if os.environ.get('SWEEP_GRID'):  # chunk x grid arms
    arms = [('one launch per layer', {'KOALA_AMD_PIPE_MT': '0'})] + [
        ('chunk %s grid %s streams %s' % (c, g, n), {'KOALA_AMD_PIPE_MT': '4096', 'KOALA_AMD_PIPE_CHUNK': c, 'KOALA_AMD_PIPE_GRID': g, 'KOALA_AMD_PIPE_STREAMS': n})
        for c in os.environ.get('SWEEP_CHUNK', '16,22,32').split(',') for g in os.environ['SWEEP_GRID'].split(',')
        for n in os.environ.get('SWEEP_STREAMS', '3').split(',')]
for T in [int(v) for v in os.environ.get('SWEEP_T', '64,32').split(',')]:
    for B in [int(v) for v in os.environ.get('SWEEP_B', '800,1024,1536,2048,2560,3072,4096').split(',')]:
        x = torch.from_numpy(np.tile(synth_streams(64, T, seed=5), ((B + 63) // 64, 1))[:B].copy()).cuda()
        y = torch.zeros_like(x)
        row = []
        for name, env in arms:
            for k in ('KOALA_AMD_PIPE_MT', 'KOALA_AMD_PIPE_CHUNK', 'KOALA_AMD_PIPE_GRID', 'KOALA_AMD_PIPE_STREAMS', 'KOALA_AMD_PIPE_WHOLE_STFT'):
                os.environ.pop(k, None)
            os.environ.update(env)
            kb = koala_amd.create_batch('key', B, T, 'bf16', model_path=model, library_path=koala_amd.developer_library_path())
            kb.set_stream(st.cuda_stream)
            for _ in range(6):
                kb.process_device(T, x.data_ptr(), y.data_ptr())
            torch.cuda.synchronize()
            reps = 40
            t0 = time.perf_counter()
            for _ in range(reps):
                kb.process_device(T, x.data_ptr(), y.data_ptr())
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            route = int(kb.debug_read('route', T)[0])
            kb.set_stream(0)
            kb.delete()
            row.append('%s: %.3f ms %.1f M%s' % (name, dt * 1e3, B * T / dt / 1e6, '' if route == 5 or 'rounds' in name else ' [route %d]' % route))
        print('%5d streams x %2d frames | ' % (B, T) + ' | '.join(row), flush=True)
        if os.environ.get('SWEEP_GRID'):
            print()
