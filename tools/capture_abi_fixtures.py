"""
Captures the keyless behaviour of the reference engine into tests/golden/abi_fixtures.json.

Runs ONLY in the build container (it loads /root/reference/lib/linux/x86_64/libpv_koala.so with ctypes and imports
nothing from the reference's Python sources); the JSON it writes is data: constants, status strings and the
status codes / message texts of argument failures.  pv_koala_init cannot succeed without an AccessKey, so no PCM
can be captured: sample-level parity with the reference stays unpinned.
"""
import ctypes as C
import json
import os
import re
import sys

REF = '/root/reference/lib/linux/x86_64/libpv_koala.so'
MODEL = '/root/reference/lib/common/koala_params.pv'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    lib = C.CDLL(REF)
    lib.pv_koala_version.restype = C.c_char_p
    lib.pv_status_to_string.restype = C.c_char_p
    lib.pv_get_sdk.restype = C.c_char_p
    lib.pv_get_error_stack.argtypes = [C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.c_int32)]
    lib.pv_free_error_stack.argtypes = [C.POINTER(C.c_char_p)]
    lib.pv_koala_init.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
    lib.pv_koala_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pv_koala_delay_sample.argtypes = [C.c_void_p, C.c_void_p]
    lib.pv_koala_reset.argtypes = [C.c_void_p]
    lib.pv_koala_delete.argtypes = [C.c_void_p]

    def stack():
        ref = C.POINTER(C.c_char_p)()
        depth = C.c_int32()
        st = lib.pv_get_error_stack(C.byref(ref), C.byref(depth))
        msgs = [ref[i].decode() for i in range(depth.value)]
        if ref:
            lib.pv_free_error_stack(ref)
        # strip the build id and location code: "<7 hex> <8 hex>: text"
        return st, [re.sub(r'^[0-9a-f]{7} [0-9A-F]{8}: ', '', m) for m in msgs], msgs

    fx = {
        'source': 'Picovoice Koala libpv_koala.so (lib/linux/x86_64), loaded keyless with ctypes',
        'version': lib.pv_koala_version().decode(),
        'frame_length': lib.pv_koala_frame_length(),
        'sample_rate': lib.pv_sample_rate(),
        'status_strings': [lib.pv_status_to_string(i).decode() for i in range(12)],
        'default_sdk': lib.pv_get_sdk().decode(),
    }
    st, texts, raw = stack()
    fx['empty_stack'] = {'status': st, 'depth': len(texts)}
    fx['message_format_example'] = None
    cases = {}
    h = C.c_void_p()

    def init_case(name, key, model, device, obj=True):
        st = lib.pv_koala_init(key, model, device, C.byref(h) if obj else None)
        s2, texts, raw = stack()
        cases[name] = {'status': st, 'messages': texts}
        if raw and fx['message_format_example'] is None:
            fx['message_format_example'] = raw[0]

    init_case('init_null_access_key', None, MODEL.encode(), b'cpu')
    init_case('init_null_model_path', b'invalid', None, b'cpu')
    init_case('init_null_object', b'invalid', MODEL.encode(), b'cpu', obj=False)
    init_case('init_missing_model', b'invalid', b'/nonexistent.pv', b'cpu')
    init_case('init_bad_device', b'invalid', MODEL.encode(), b'foo')
    st = lib.pv_koala_process(None, None, None)
    cases['process_null_object'] = {'status': st, 'messages': stack()[1]}
    st = lib.pv_koala_delay_sample(None, None)
    cases['delay_sample_null_object'] = {'status': st, 'messages': stack()[1]}
    st = lib.pv_koala_reset(None)
    cases['reset_null_object'] = {'status': st, 'messages': stack()[1]}
    lib.pv_koala_delete(None)
    cases['delete_null_object'] = {'status': 0, 'messages': []}
    fx['cases'] = cases
    lib.pv_set_sdk.argtypes = [C.c_char_p]
    lib.pv_set_sdk(b'python')
    fx['sdk_after_set_python'] = lib.pv_get_sdk().decode()
    out = os.path.join(ROOT, 'tests', 'golden', 'abi_fixtures.json')
    with open(out, 'w') as f:
        json.dump(fx, f, indent=1, sort_keys=True)
    print(json.dumps(fx, indent=1, sort_keys=True))


if __name__ == '__main__':
    sys.exit(main())
