"""Developer tool: what overlaps in the asynchronous host path.  Post-processes a `rocprofv3 --kernel-trace --memory-copy-trace
--output-format csv` run of tools/host_async.py: busy time (union of intervals) of the kernels, of the host-to-device and of the
device-to-host copies inside the window in which all three are active, against that window's length.
    cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/async_trace -o t -- python tools/host_async.py
    python tools/async_trace.py gpurun_out/async_trace"""
import csv
import glob
import os
import sys


def union(iv):
    iv = sorted(iv)
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    return busy


def load(pattern, root):
    rows = []
    for f in glob.glob(os.path.join(root, '**', pattern), recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def main(root):
    k = load('*kernel_trace.csv', root)
    m = load('*memory_copy_trace.csv', root)
    iv = lambda r: (int(r['Start_Timestamp']), int(r['End_Timestamp']))  # noqa: E731
    long_ = lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 500000  # noqa: E731  (the calls' 134 MB copies: > 0.5 ms)
    kern = [iv(r) for r in k if 'kns::' in r.get('Kernel_Name', '')]
    h2d = sorted(iv(r) for r in m if long_(r) and 'HOST_TO_DEVICE' in r['Direction'].upper())
    # the asynchronous calls' copy-outs into page-locked host memory are done by the runtime's copy KERNEL (no SDMA record);
    # the synchronous path's strided 2-D copies appear as DEVICE_TO_DEVICE records on the GPU agent
    d2h = sorted([iv(r) for r in k if 'copyBuffer' in r.get('Kernel_Name', '') and long_(r)] +
                 [iv(r) for r in m if long_(r) and 'HOST_TO_DEVICE' not in r['Direction'].upper()])
    print('%d engine kernels, %d large host-to-device copies, %d large copy-outs' % (len(kern), len(h2d), len(d2h)))
    if not (kern and h2d and d2h):
        return
    # steady state of the LAST asynchronous run of tools/host_async.py (30 calls, three buffer pairs): its copy-ins are the last 30
    # whole-call host-to-device copies (> 2 ms each; the synchronous path that follows copies in sub-chunks of 16 frames)
    whole = [c for c in h2d if c[1] - c[0] > 2000000][-30:]
    lo, hi = whole[3][0], whole[-1][1]  # (skip the pipeline's fill)
    clip = lambda x: [(max(s, lo), min(e, hi)) for s, e in x if e > lo and s < hi]  # noqa: E731
    span = hi - lo
    total = 0
    for name, x in (('engine kernels', kern), ('host-to-device', h2d), ('copy-out', d2h)):
        b = union(clip(x))
        total += b
        print('%-16s busy %7.2f ms of %7.2f ms = %5.1f %%' % (name, b / 1e6, span / 1e6, 100.0 * b / span))
    print('calls in the window: %d -> %.3f ms per call; sum of the three busy times = %.2f x the window' % (len(whole) - 3, span / 1e6 / (len(whole) - 3), total / span))


if __name__ == '__main__':
    main(sys.argv[1])
