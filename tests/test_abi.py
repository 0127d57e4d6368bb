"""
CPU tests of the drop-in boundary: libpv_koala.so loads without a GPU, exports every symbol include/*.h declares,
reproduces the reference's keyless behaviour (tests/golden/abi_fixtures.json, captured from the shipped library)
and fails loudly -- never silently -- when no MI355X is reachable.
"""
import ctypes as C
import json
import os
import re
import threading

import numpy as np
import pytest

import koala_amd
from conftest import GOLDEN, ROOT


@pytest.fixture(scope='module')
def fx():
    with open(os.path.join(GOLDEN, 'abi_fixtures.json')) as f:
        return json.load(f)


@pytest.fixture(scope='module')
def lib(native_library):
    l = C.CDLL(native_library)
    l.pv_koala_version.restype = C.c_char_p
    l.pv_status_to_string.restype = C.c_char_p
    l.pv_get_sdk.restype = C.c_char_p
    l.pv_set_sdk.argtypes = [C.c_char_p]
    l.pv_get_error_stack.argtypes = [C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.c_int32)]
    l.pv_free_error_stack.argtypes = [C.POINTER(C.c_char_p)]
    l.pv_koala_init.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
    l.pv_koala_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    l.pv_koala_delay_sample.argtypes = [C.c_void_p, C.c_void_p]
    l.pv_koala_reset.argtypes = [C.c_void_p]
    l.pv_koala_delete.argtypes = [C.c_void_p]
    return l


def stack(lib):
    ref = C.POINTER(C.c_char_p)()
    depth = C.c_int32()
    st = lib.pv_get_error_stack(C.byref(ref), C.byref(depth))
    raw = [ref[i].decode() for i in range(depth.value)]
    if ref:
        lib.pv_free_error_stack(ref)
    return st, raw


def strip(msgs):
    return [re.sub(r'^[0-9a-f]{7} [0-9A-F]{8}: ', '', m) for m in msgs]


def has_gpu():
    return len(koala_amd.available_devices()) > 0


def test_exports_every_declared_symbol(native_library):
    declared = set()
    for h in ('picovoice.h', 'pv_koala.h', 'pv_koala_batch.h'):
        text = open(os.path.join(ROOT, 'include', h)).read()
        declared |= set(re.findall(r'PV_API\s+[\w\s\*]+?\b(pv_\w+)\s*\(', text))
    assert len(declared) >= 30
    l = C.CDLL(native_library)
    for name in sorted(declared):
        assert hasattr(l, name), name
    # the 18 dynamic symbols of the reference library
    for name in ('pv_koala_init pv_koala_delete pv_koala_process pv_koala_reset pv_koala_delay_sample '
                 'pv_koala_frame_length pv_koala_version pv_koala_list_hardware_devices '
                 'pv_koala_free_hardware_devices pv_sample_rate pv_status_to_string pv_get_error_stack '
                 'pv_free_error_stack pv_set_sdk pv_get_sdk pv_free pv_log_enable pv_log_disable').split():
        assert hasattr(l, name), name


def test_dynamic_symbol_table_is_the_abi_and_nothing_else(native_library):
    """The linker version script (koala_amd/csrc/libpv_koala.map) keeps kernel stubs, __hip_cuid_* and libstdc++ template
    instances out of the dynamic symbol table: what a dlsym caller can see is the reference ABI + the batch extension."""
    import subprocess
    for lib in (native_library, koala_amd.developer_library_path()):
        out = subprocess.run(['nm', '-D', '--defined-only', lib], capture_output=True, text=True, check=True).stdout
        names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
        assert len(names) >= 30
        stray = [n for n in names if not n.startswith('pv_')]
        assert not stray, stray
    # ... and it is a superset of what the reference's own library exports (captured list)
    ref = set('pv_free pv_free_error_stack pv_get_error_stack pv_get_sdk pv_koala_delay_sample pv_koala_delete '
              'pv_koala_frame_length pv_koala_free_hardware_devices pv_koala_init pv_koala_list_hardware_devices pv_koala_process '
              'pv_koala_reset pv_koala_version pv_log_disable pv_log_enable pv_sample_rate pv_set_sdk pv_status_to_string'.split())
    assert ref <= set(names)


def test_product_library_reads_no_developer_switch(native_library):
    """KOALA_AMD_ONLY_CLASS & co. exist in the -DKNS_DEV build only: their names do not even occur in the product binary."""
    blob = open(native_library, 'rb').read()
    dev = open(koala_amd.developer_library_path(), 'rb').read()
    for name in (b'KOALA_AMD_ONLY_CLASS', b'KOALA_AMD_GRU_STREAM', b'KOALA_AMD_GEMM_GENERIC', b'KOALA_AMD_SMALL_MT',
                 b'KOALA_AMD_DEBUG_TAPS', b'KOALA_AMD_HOST_CHUNK', b'KOALA_AMD_HOST_SCHED', b'KOALA_AMD_NO_SPIN_WAIT'):
        assert name not in blob, name
        assert name in dev, name
    assert b'KOALA_AMD_PRECISION' in blob  # the documented run-time options of the single-stream ABI: precision ...
    assert b'KOALA_AMD_WAIT' in blob       # ... and how a one-frame call waits for its frame (poll / block)


def test_constants(lib, fx):
    assert lib.pv_koala_version().decode() == fx['version']
    assert lib.pv_koala_frame_length() == fx['frame_length']
    assert lib.pv_sample_rate() == fx['sample_rate']
    assert [lib.pv_status_to_string(i).decode() for i in range(12)] == fx['status_strings']
    assert lib.pv_status_to_string(12) is None


def test_sdk_tag(lib, fx):
    lib.pv_set_sdk(fx['default_sdk'].encode())
    assert lib.pv_get_sdk().decode() == fx['default_sdk']
    lib.pv_set_sdk(b'python')
    assert lib.pv_get_sdk().decode() == fx['sdk_after_set_python']


def test_error_stack_protocol(lib, fx, gate_model):
    st, raw = stack(lib)
    assert st == fx['empty_stack']['status'] and len(raw) == fx['empty_stack']['depth']
    h = C.c_void_p()
    cases = fx['cases']
    calls = {
        'init_null_access_key': lambda: lib.pv_koala_init(None, gate_model.encode(), b'best', C.byref(h)),
        'init_null_model_path': lambda: lib.pv_koala_init(b'key', None, b'best', C.byref(h)),
        'init_null_object': lambda: lib.pv_koala_init(b'key', gate_model.encode(), b'best', None),
        'init_bad_device': lambda: lib.pv_koala_init(b'key', gate_model.encode(), b'foo', C.byref(h)),
        'process_null_object': lambda: lib.pv_koala_process(None, None, None),
        'delay_sample_null_object': lambda: lib.pv_koala_delay_sample(None, None),
        'reset_null_object': lambda: lib.pv_koala_reset(None),
    }
    for name, call in calls.items():
        assert call() == cases[name]['status'], name
        st, raw = stack(lib)
        assert strip(raw) == cases[name]['messages'], name
        for m in raw:
            assert re.match(r'^[0-9a-f]{7} [0-9A-F]{8}: ', m)  # same "<build id> <code>: text" shape
        assert stack(lib)[0] == fx['empty_stack']['status']  # drained by the first read
    # missing model: same status and first message; the reference's second line is an opaque token
    assert lib.pv_koala_init(b'key', b'/nonexistent.pv', b'best', C.byref(h)) == cases['init_missing_model']['status']
    st, raw = stack(lib)
    assert strip(raw)[0] == cases['init_missing_model']['messages'][0] and 0 < len(raw) < 8
    lib.pv_koala_delete(None)  # no-op, as in the reference


def test_device_null_is_an_argument_error_not_a_crash(lib, gate_model):
    h = C.c_void_p()
    assert lib.pv_koala_init(b'key', gate_model.encode(), None, C.byref(h)) == 3
    assert strip(stack(lib)[1]) == ['Argument `device` is NULL.']


def test_error_stack_is_thread_local(lib):
    assert lib.pv_koala_process(None, None, None) == 3
    seen = {}

    def other():
        seen['other'] = stack(lib)

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen['other'][0] == 6 and seen['other'][1] == []
    assert len(stack(lib)[1]) == 1


def test_device_grammar(lib, gate_model):
    h = C.c_void_p()
    for bad in (b'', b'GPU', b'gpu:', b'gpu:x', b'gpu:-1', b'cpu:', b'invalid:9', b'best:0'):
        assert lib.pv_koala_init(b'key', gate_model.encode(), bad, C.byref(h)) == 3, bad
        assert 'is not a valid device string' in stack(lib)[1][0]
    # a well-formed CPU request is refused, loudly: there is no CPU path to fall back to
    for cpu in (b'cpu', b'cpu:4'):
        assert lib.pv_koala_init(b'key', gate_model.encode(), cpu, C.byref(h)) == 7
        assert 'no CPU backend' in stack(lib)[1][0]


def test_no_gpu_means_runtime_error_never_a_fallback(lib, gate_model):
    if has_gpu():
        pytest.skip('a GPU is visible')
    h = C.c_void_p()
    for dev in (b'best', b'gpu', b'gpu:0'):
        assert lib.pv_koala_init(b'key', gate_model.encode(), dev, C.byref(h)) == 7
        assert strip(stack(lib)[1])[0] == 'Failed to communicate with device.'
    assert koala_amd.available_devices() == []
    with pytest.raises(koala_amd.KoalaRuntimeError) as e:
        koala_amd.create('key', model_path=gate_model)
    assert len(e.value.message_stack) > 0
    with pytest.raises(koala_amd.KoalaRuntimeError):
        koala_amd.create_batch('key', 4, model_path=gate_model)


def test_python_surface_argument_checks(gate_model, native_library):
    # exception types and texts of reference binding/python/_koala.py:142-152
    with pytest.raises(koala_amd.KoalaInvalidArgumentError, match='`access_key` should be a non-empty string.'):
        koala_amd.create('')
    with pytest.raises(koala_amd.KoalaIOError, match='Could not find model file at `/nope`.'):
        koala_amd.create('k', model_path='/nope')
    with pytest.raises(koala_amd.KoalaIOError, match="Could not find Koala's dynamic library at `/nope.so`."):
        koala_amd.create('k', model_path=gate_model, library_path='/nope.so')
    with pytest.raises(koala_amd.KoalaInvalidArgumentError, match='`device` should be a non-empty string.'):
        koala_amd.Koala('k', gate_model, '', native_library)
    with pytest.raises(koala_amd.KoalaInvalidArgumentError):
        koala_amd.create('k', model_path=gate_model, device='invalid:9')  # Android testInitFailWithInvalidDevice


def test_koala_error_formatting():
    e = koala_amd.KoalaError('Initialization failed', ['a', 'b'])
    assert str(e) == 'Initialization failed:\n  [0] a\n  [1] b'
    assert str(koala_amd.KoalaError('plain')) == 'plain'
    assert e.message == 'Initialization failed' and list(e.message_stack) == ['a', 'b']


def test_model_file_checks(lib, tmp_path):
    bad = tmp_path / 'bad.kns'
    bad.write_bytes(b'KNS0\0\0\0\0' + bytes(64))
    h = C.c_void_p()
    assert lib.pv_koala_init(b'key', str(bad).encode(), b'best', C.byref(h)) == 2
    assert 'not a Koala (KNS1) model file' in stack(lib)[1][0]


_HIDDEN_GPU_SCRIPT = r'''
import json, sys
sys.path.insert(0, %(root)r)
import koala_amd
res = {}
def attempt(name, **kw):
    args = dict(access_key='reference-binding-check')
    args.update(kw)
    try:
        koala_amd.create(**args).delete()
        res[name] = {'exception': None}
    except Exception as e:
        res[name] = {'exception': type(e).__name__, 'str': str(e), 'message_stack': list(getattr(e, 'message_stack', []) or [])}
attempt('init_no_gpu_best')
attempt('init_no_gpu_gpu0', device='gpu:0')
attempt('init_bad_device', device='foo')
attempt('init_cpu_device', device='cpu:1')
attempt('init_missing_model_python_side', model_path='/nope.kns')
attempt('init_empty_key_python_side', access_key='')
attempt('init_missing_library_python_side', library_path='/nope.so')
res['list_hardware_devices'] = list(koala_amd.available_devices())
print('CAPTURE', json.dumps(res))
'''


def test_own_binding_shows_a_caller_what_the_reference_binding_shows():
    """tests/golden/reference_binding_capture.json is what the REFERENCE's unmodified binding/python/_koala.py reports when it
    is pointed at this library on a box without a GPU (tools/check_reference_binding.py, build container only).  koala_amd's
    own binding must give a caller the same exception types, str() texts and message stacks.  Runs with the GPUs hidden, so
    it checks the same thing here and on the GPU box."""
    import subprocess
    import sys
    with open(os.path.join(GOLDEN, 'reference_binding_capture.json')) as f:
        cap = json.load(f)
    assert all(cap['symbols_resolved'].values())
    env = dict(os.environ, HIP_VISIBLE_DEVICES='-1', ROCR_VISIBLE_DEVICES='-1')
    out = subprocess.run([sys.executable, '-c', _HIDDEN_GPU_SCRIPT % {'root': ROOT}], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('CAPTURE')][-1][8:])
    for name in ('init_no_gpu_best', 'init_no_gpu_gpu0', 'init_bad_device', 'init_cpu_device',
                 'init_missing_model_python_side', 'init_empty_key_python_side', 'init_missing_library_python_side'):
        assert got[name]['exception'] == cap[name]['exception'], name
        assert got[name]['str'] == cap[name]['str'], name
        assert got[name]['message_stack'] == cap[name]['message_stack'], name
    assert got['list_hardware_devices'] == cap['list_hardware_devices'] == []


def test_reference_pv_model_gets_a_dedicated_error(lib, tmp_path):
    """A reference `.pv` parameter file (magic `koala3.0.0`) is recognised and refused by name, not with
    the generic read error."""
    fake = tmp_path / 'koala_params.pv'
    fake.write_bytes(b'koala3.0.0' + bytes(5) + bytes(4096))
    h = C.c_void_p()
    st = lib.pv_koala_init(b'key', str(fake).encode(), b'best', C.byref(h))
    assert st == 2  # PV_STATUS_IO_ERROR
    _, msgs = stack(lib)
    assert 'reference Koala `.pv` model' in msgs[0] and 'not supported' in msgs[0]


def test_listed_device_strings_are_accepted_back(lib, random_model):
    """pv_koala_list_hardware_devices prints `gpu:N - <name>`; create(device=<that string>) must parse (it may then fail for
    lack of a GPU, but never as `not a valid device string`)."""
    h = C.c_void_p()
    st = lib.pv_koala_init(b'key', random_model.encode(), b'gpu:0 - AMD Instinct MI355X', C.byref(h))
    _, msgs = stack(lib)
    assert not any('not a valid device string' in m for m in msgs)
    if st == 0:
        lib.pv_koala_delete(h)
    st = lib.pv_koala_init(b'key', random_model.encode(), b' - x', C.byref(h))
    assert st == 3 and 'not a valid device string' in stack(lib)[1][0]
