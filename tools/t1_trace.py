"""One-frame calls at 4 096 streams (bf16) for `rocprofv3 --kernel-trace`: 300 frame steps; run as
   cd /tmp && rocprofv3 --kernel-trace --stats -d <dir> -o t1 -- python <repo>/tools/t1_trace.py
and summarise with tools/t1_gaps.py <db> (kernel durations and the idle time between consecutive kernels of a frame step)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import koala_amd  # noqa: E402
from koala_amd import params  # noqa: E402
from koala_amd.workload import synth_streams  # noqa: E402

B = int(os.environ.get('T1_B', 4096))
model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
x = torch.from_numpy(np.tile(synth_streams(64, 1, 1), ((B + 63) // 64, 1))[:B]).cuda()
y = torch.empty_like(x)
kb = koala_amd.create_batch('k', B, 1, os.environ.get('T1_PREC', 'bf16'), model_path=model)
kb.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(300):
    kb.process_device(1, x.data_ptr(), y.data_ptr())
torch.cuda.synchronize()
kb.delete()
