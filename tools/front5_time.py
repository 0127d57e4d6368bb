"""The five-frame front-end (KNS-v1.1, `gemm_front5_kernel`): ms per call of the whole engine and of the launch class the front-end
is in, per library.  python tools/front5_time.py [lib.so ...]   (default: the developer build; build/ab/lib*.so: tools/variant_lib.sh)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import koala_amd  # noqa: E402
from koala_amd import params  # noqa: E402
from koala_amd.workload import synth_streams  # noqa: E402


def main():
    libs = sys.argv[1:] or [koala_amd.developer_library_path()]
    B, T = int(os.environ.get('F5_B', 4096)), int(os.environ.get('F5_T', 64))
    model = params.ensure_params(os.path.join(ROOT, 'build', 'random5_1234.kns'), 'random5', 1234)
    x = torch.from_numpy(np.tile(synth_streams(64, T, 1), (B // 64, 1))).cuda()
    y = torch.empty_like(x)
    for lib in libs:
        kb = koala_amd.create_batch('k', B, T, 'bf16', model_path=model, library_path=os.path.abspath(lib))
        kb.set_stream(torch.cuda.current_stream().cuda_stream)
        for _ in range(20):
            kb.process_device(T, x.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        kb.profile_enable(True)
        for _ in range(20):
            kb.process_device(T, x.data_ptr(), y.data_ptr())
        p = kb.profile_read()
        kb.delete()
        heads = p['gemm_head']
        print('%s: gemm_head class %.1f us per call over %d launches (four stage heads + the front-end)' % (
            os.path.basename(lib), heads['ms'] * 1e3 / 20, heads['launches'] // 20), flush=True)


if __name__ == '__main__':
    main()
