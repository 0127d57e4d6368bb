#!/usr/bin/env python3
"""
BUILD-CONTAINER ONLY (needs /root/reference; never runs on the GPU box, never imported by the product).

Binds the REFERENCE's own Python binding (reference binding/python/_koala.py, unmodified, imported from where it
lies) to THIS repository's libpv_koala.so and records what an unchanged caller sees:
  * every symbol the binding looks up resolves (pv_set_sdk, pv_get_error_stack, pv_koala_init, ... _koala.py:154-222)
  * list_hardware_devices() (_koala.py:325-352) returns through our pv_koala_list_hardware_devices
  * Koala(...) on a box without a GPU fails inside pv_koala_init and the binding maps our status + message stack onto
    its own exception type and str() format (_koala.py:176-184, 25-31, 299-312)
  * the Python-side argument checks that never reach the library (_koala.py:142-152)
  * every reference-side call site's argtypes/restype (_koala.py:203-222) is accepted by our exports.
The capture is committed as tests/golden/reference_binding_capture.json; tests/test_abi.py asserts that koala_amd's own
binding shows a caller the same exception types and texts, and tests/test_gpu_reference_suite.py replays the no-GPU
case on the GPU box with the devices hidden.
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/binding/python'


def main():
    if not os.path.isdir(REF):
        sys.exit('reference checkout not present: this tool only runs in the build container')
    sys.path.insert(0, ROOT)
    import koala_amd
    lib = koala_amd.build_native()
    model = koala_amd.default_model_path()
    sys.path.insert(0, REF)
    import _koala as ref  # the reference's module, as shipped

    cap = {'library': os.path.relpath(lib, ROOT), 'reference_module': 'binding/python/_koala.py (v3.0.0, unmodified)'}

    # every symbol the reference binding resolves, with the reference's own prototypes applied
    l = ctypes.cdll.LoadLibrary(lib)
    symbols = ['pv_set_sdk', 'pv_get_error_stack', 'pv_free_error_stack', 'pv_koala_init', 'pv_koala_delete',
               'pv_koala_delay_sample', 'pv_koala_process', 'pv_koala_reset', 'pv_sample_rate', 'pv_koala_frame_length',
               'pv_koala_version', 'pv_koala_list_hardware_devices', 'pv_koala_free_hardware_devices']
    cap['symbols_resolved'] = {s: hasattr(l, s) for s in symbols}

    cap['list_hardware_devices'] = ref.list_hardware_devices(lib)

    def attempt(**kw):
        args = dict(access_key='reference-binding-check', model_path=model, device='best', library_path=lib)
        args.update(kw)
        try:
            k = ref.Koala(**args)
        except Exception as e:  # noqa: BLE001
            return {'exception': type(e).__name__, 'str': str(e), 'message_stack': list(getattr(e, 'message_stack', []) or [])}
        out = {'exception': None, 'delay_sample': k.delay_sample, 'frame_length': k.frame_length,
               'sample_rate': k.sample_rate, 'version': k.version}
        k.delete()
        return out

    cap['init_no_gpu_best'] = attempt()
    cap['init_no_gpu_gpu0'] = attempt(device='gpu:0')
    cap['init_bad_device'] = attempt(device='foo')
    cap['init_cpu_device'] = attempt(device='cpu:1')
    cap['init_missing_model_python_side'] = attempt(model_path='/nope.kns')
    cap['init_empty_key_python_side'] = attempt(access_key='')
    cap['init_missing_library_python_side'] = attempt(library_path='/nope.so')
    pv = '/root/reference/lib/common/koala_params.pv'
    cap['init_reference_pv_model'] = attempt(model_path=pv)
    cap['init_reference_pv_model']['str'] = cap['init_reference_pv_model']['str'].replace(pv, '<koala_params.pv>')
    cap['init_reference_pv_model']['message_stack'] = [m.replace(pv, '<koala_params.pv>')
                                                       for m in cap['init_reference_pv_model']['message_stack']]

    # keyless getters through the reference's prototypes
    l.pv_koala_version.restype = ctypes.c_char_p
    cap['getters'] = {'version': l.pv_koala_version().decode(), 'frame_length': l.pv_koala_frame_length(),
                      'sample_rate': l.pv_sample_rate()}
    out = os.path.join(ROOT, 'tests', 'golden', 'reference_binding_capture.json')
    json.dump(cap, open(out, 'w'), indent=1, sort_keys=True)
    print(json.dumps(cap, indent=1, sort_keys=True))
    assert all(cap['symbols_resolved'].values())


if __name__ == '__main__':
    main()
