"""
The C-ABI shim (koala_amd/csrc/pv_api.cpp) under AddressSanitizer + UndefinedBehaviorSanitizer: every entry
point of include/pv_koala.h, include/picovoice.h and include/pv_koala_batch.h driven through its argument checks, its thread-local
error stack (reference contract: include/picovoice.h:73-86, binding/python/_koala.py:299-312) and its ownership rules by a C++ host
(tests/abi_sanitizer/driver.cpp), with the engine replaced by a host-only double (engine_stub.cpp): no GPU, no HIP runtime.
"""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT, model_file

SRC = os.path.join(ROOT, 'tests', 'abi_sanitizer')


def test_c_abi_shim_under_asan_and_ubsan(tmp_path):
    gxx = shutil.which('g++')
    if not gxx or not os.path.isdir('/opt/rocm/include/hip'):
        pytest.skip('needs g++ and the HIP headers')
    exe = str(tmp_path / 'abi_driver')
    cmd = [gxx, '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-fno-omit-frame-pointer',
           '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I' + os.path.join(ROOT, 'include'),
           '-I' + os.path.join(ROOT, 'koala_amd', 'csrc'), '-Wno-deprecated-declarations', '-Wno-unused-result',
           os.path.join(ROOT, 'koala_amd', 'csrc', 'pv_api.cpp'), os.path.join(SRC, 'engine_stub.cpp'), os.path.join(SRC, 'driver.cpp'),
           '-o', exe, '-lpthread']
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and 'sanitizer' in build.stderr.lower() and 'cannot find' in build.stderr.lower():
        pytest.skip('sanitizer runtimes not installed: ' + build.stderr[-300:])
    assert build.returncode == 0, build.stderr[-4000:]
    garbage = tmp_path / 'garbage.bin'
    garbage.write_bytes(b'this is not a model file at all')
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    for k in ('STUB_GPUS', 'STUB_OOM', 'STUB_FAIL_PROCESS', 'STUB_THROW', 'LD_PRELOAD'):
        env.pop(k, None)
    run = subprocess.run([exe, model_file('random', 1234), str(garbage)], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-6000:])
