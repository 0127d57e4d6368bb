"""Developer tool: s_memtime stamps (shader-clock cycles) inside gru_wave_kernel -- the workgroup in slot 1 of XCD 3 (layer 3), second
m-tile of its group, in a mid-call launch (needs a -DKNS_TIMING -DKNS_DEV build, made here).
  python tools/wave_timing.py [streams]       WAVE_PREC=fp32|bf16, KOALA_AMD_WAVE_GROUP=<m-tiles per workgroup>"""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
lib = os.path.join(ROOT, 'build', 'libpv_koala_wtiming.so')
os.makedirs(os.path.dirname(lib), exist_ok=True)
src = [os.path.join(ROOT, 'koala_amd', 'csrc', f) for f in ('kns_stft.hip', 'kns_gemm.hip', 'kns_gru.hip', 'kns_gruq.hip', 'kns_engine.cpp', 'pv_api.cpp')]
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden',
                       '-ffp-contract=off', '-Xarch_host', '-mfma', '-Xarch_host', '-mavx2', '-DKNS_TIMING', '-DKNS_DEV', '-x', 'hip'] + src + ['-shared', '-o', lib])
import koala_amd
from conftest import model_file, synth_streams
os.environ.setdefault('KOALA_AMD_WAVE_MT', '4096')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = 32
prec = os.environ.get('WAVE_PREC', 'fp32')
x = torch.from_numpy(synth_streams(B, T, seed=3)).cuda()
y = torch.empty_like(x)
kb = koala_amd.create_batch('k', B, T, prec, model_path=model_file('random', 1234), library_path=lib)
for _ in range(3):
    kb.process_device(T, x.data_ptr(), y.data_ptr())
kb.synchronize()
l = C.CDLL(lib)
buf = (C.c_ulonglong * 128)()
l.pv_koala_debug_timing(buf)
t = np.array(buf[:128], dtype=np.int64).reshape(8, 16)
base = t[:4, 0].min()
print('%s %d streams, group %s: stamps of roles 0-2 (MFMA waves: top of the m-tile, after B1, MFMAs done, after B2) and 3 '
      '(top, after B1, next requests issued + gates of the m-tile before done, after B2, exchange taken + requested blocks landed); cycles from the earliest' % (prec, B, os.environ.get('KOALA_AMD_WAVE_GROUP', 'auto')))
for w in range(4):
    n = 4 if w < 3 else 5
    print('role %d:' % w, ' '.join('%6d' % (v - base) for v in t[w, :n]), ' | deltas:', ' '.join('%5d' % d for d in np.diff(t[w, :n])))
