"""Developer tool (through gpurun): do the recurrent launches of a mid-size batch really overlap?  Runs N calls of B streams x 64 frames under
`rocprofv3 --kernel-trace --output-format csv` (twice: product route, and KOALA_AMD_PIPE_MT=0), then for the steady-state window prints, per
kernel class, the SUM of the launch durations against the UNION of their intervals (sum / union = how many ran side by side on average).
    python tools/pipe_trace.py run B          (the traced workload; developer library)
    python tools/pipe_trace.py report DIR     (post-processing)
    tools/pipe_trace.py all B                 (both, through rocprofv3; writes gpurun_out/pipe_trace_B.txt)"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(B):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import koala_amd
    from koala_amd import params
    from koala_amd.workload import synth_streams
    model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    x = torch.from_numpy(np.tile(synth_streams(64, 64, seed=5), ((B + 63) // 64, 1))[:B].copy()).cuda()
    y = torch.zeros_like(x)
    kb = koala_amd.create_batch('key', B, 64, 'bf16', model_path=model, library_path=koala_amd.developer_library_path())
    kb.set_stream(st.cuda_stream)
    for _ in range(24):
        kb.process_device(64, x.data_ptr(), y.data_ptr())
    torch.cuda.synchronize()
    kb.set_stream(0)
    kb.delete()


def union(iv):
    busy, cs, ce = 0, None, None
    for s, e in sorted(iv):
        if ce is None or s > ce:
            if ce is not None:
                busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return busy + (ce - cs if ce is not None else 0)


def report(root):
    rows = []
    for f in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
        rows += list(csv.DictReader(open(f)))
    k = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows if 'kns::' in r['Kernel_Name']))
    syn = [r for r in k if 'synthesis_kernel' in r[2]]
    lo, hi = syn[7][1], syn[-1][1]  # steady state: behind the 8th call's synthesis up to the last one's
    calls = len(syn) - 8
    k = [r for r in k if r[0] >= lo and r[1] <= hi]
    print('%d calls in the window, %.3f ms per call' % (calls, (hi - lo) / 1e6 / calls))
    for name in ('gru_resident8', 'gemm_ws2', 'gemm_', 'analysis', 'synthesis', 'kns::'):
        iv = [(s, e) for s, e, n in k if name in n]
        if iv:
            sm, un = sum(e - s for s, e in iv), union(iv)
            print('%-14s %4d launches per call, mean %6.1f us | sum %7.3f ms per call, union %7.3f ms per call: %.2f side by side'
                  % (name if name != 'kns::' else 'all kernels', len(iv) // calls, sm / len(iv) / 1e3, sm / 1e6 / calls, un / 1e6 / calls, sm / un))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(int(sys.argv[2]))
    elif sys.argv[1] == 'report':
        report(sys.argv[2])
    else:
        B = sys.argv[2]
        out = os.path.join(ROOT, 'gpurun_out', 'pipe_trace_%s' % B)
        text = ''
        for arm, env in (('product route (two sub-chunks on two streams)', {}), ('one launch per layer (KOALA_AMD_PIPE_MT=0)', {'KOALA_AMD_PIPE_MT': '0'})):
            d = out + ('_off' if env else '_on')
            subprocess.run(['rocprofv3', '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 't', '--', sys.executable, os.path.abspath(__file__), 'run', B],
                           env=dict(os.environ, TMPDIR='/tmp', **env), cwd='/tmp', stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), 'report', d], capture_output=True, text=True).stdout
            text += '== %s streams x 64 frames, bf16: %s\n%s' % (B, arm, r)
            subprocess.run(['rm', '-rf', d])
        open(out + '.txt', 'w').write(text)
        print(text)
