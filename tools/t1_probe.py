"""One frame per call (T = 1) at several stream counts, both precisions: microseconds per frame step and frames/s with device
buffers and HIP-graph replay, the product library's own dispatch (low-latency layer kernel, one-step quad kernel, chunked
kernels -- kns_engine.cpp run_device()).  `KOALA_AMD_NO_QUAD=1 python tools/t1_probe.py` (developer library) is the A/B arm."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import koala_amd  # noqa: E402
from koala_amd import params  # noqa: E402
from koala_amd.workload import synth_streams  # noqa: E402


def main():
    model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
    dev = any(k.startswith('KOALA_AMD_') and k != 'KOALA_AMD_PRECISION' for k in os.environ)
    lib = koala_amd.developer_library_path() if dev else None
    if os.environ.get('T1_LIB'):  # a variant library (tools/variant_lib.sh)
        lib = os.path.abspath(os.environ['T1_LIB'])
    sizes = [int(v) for v in os.environ.get('T1_SIZES', '512,1024,1776,1792,2048,3072,4096,8192').split(',')]
    for B in sizes:
        x = torch.from_numpy(np.tile(synth_streams(64, 1, 1), (B // 64 + 1, 1))[:B].copy()).cuda()
        y = torch.empty_like(x)
        for prec in os.environ.get('T1_PRECISIONS', 'bf16,fp32').split(','):
            kb = koala_amd.create_batch('k', B, 1, prec, model_path=model, library_path=lib)
            kb.set_stream(torch.cuda.current_stream().cuda_stream)
            for _ in range(50):
                kb.process_device(1, x.data_ptr(), y.data_ptr())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(500):
                kb.process_device(1, x.data_ptr(), y.data_ptr())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print('B=%5d %s: %6.1f us per frame step, %6.2f M frames/s' % (B, prec, dt / 500 * 1e6, B * 500 / dt / 1e6), flush=True)
            kb.delete()


if __name__ == '__main__':
    main()
