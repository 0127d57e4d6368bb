"""Developer tool: in-kernel s_memtime stamps of one recurrent step (needs a -DKNS_TIMING build)."""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, 'build', 'libpv_koala_timing%s.so' % os.environ.get('TIMING_TAG', ''))
os.makedirs(os.path.dirname(lib), exist_ok=True)
src = [os.path.join(ROOT, 'koala_amd', 'csrc', f) for f in ('kns_stft.hip', 'kns_gemm.hip', 'kns_gru.hip', 'kns_gruq.hip', 'kns_engine.cpp', 'pv_api.cpp')]
if not (os.environ.get('TIMING_NOBUILD') and os.path.exists(lib)):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden',
                       '-ffp-contract=off', '-DKNS_TIMING'] + os.environ.get('TIMING_FLAGS', '').split() + ['-x', 'hip'] + src + ['-shared', '-o', lib])
import koala_amd
from koala_amd import params
from koala_amd.workload import synth_streams
model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
B, T = 4096, int(os.environ.get('TIMING_T', 32))
x = torch.from_numpy(np.tile(synth_streams(64, T, 1), (B // 64, 1))).cuda()
y = torch.empty_like(x)
kb = koala_amd.create_batch('k', B, T, 'bf16', model_path=model, library_path=lib)
for _ in range(3):
    kb.process_device(T, x.data_ptr(), y.data_ptr())
kb.synchronize()
l = C.CDLL(lib)
buf = (C.c_ulonglong * 128)()
l.pv_koala_debug_timing(buf)
t = np.array(buf[:128], dtype=np.int64).reshape(8, 16)
n = int(os.environ.get('TIMING_STAMPS', 9))
base = t[:, 0].min()
print('per-wave stamps of workgroup 0, step 5 (s_memtime ticks from the earliest stamp 0); rows = waves')
for w in range(8):
    print('wave %d:' % w, ' '.join('%6d' % (v - base) for v in t[w, :n]), ' | deltas:', ' '.join('%5d' % d for d in np.diff(t[w, :n])))
print('steady state: %.0f ticks per step (steps 8..24, wave 0)' % ((t[0, 10] - t[0, 9]) / 16.0))
kb.profile_enable(True)
for _ in range(3):
    kb.process_device(T, x.data_ptr(), y.data_ptr())
pr = kb.profile_read()
us = pr['gru_recurrent']['ms'] / pr['gru_recurrent']['launches'] * 1e3
print('recurrent kernel %.1f us per launch = %.3f us per step -> %.2f GHz' % (us, us / T, (t[0, 10] - t[0, 9]) / 16.0 / (us / T) / 1e3))
