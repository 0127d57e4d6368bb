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
    kern = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in k if 'kns::' in r.get('Kernel_Name', '')]
    big = [r for r in m if int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 500000]  # the calls' 134 MB copies (> 0.5 ms)
    h2d = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in big if 'HOST_TO_DEVICE' in r.get('Direction', '').upper()]
    d2h = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in big if 'DEVICE_TO_HOST' in r.get('Direction', '').upper()]
    print('%d engine kernels, %d large host-to-device and %d large device-to-host copies' % (len(kern), len(h2d), len(d2h)))
    if not (kern and h2d and d2h):
        return
    # the steady-state window of the LAST asynchronous run: from the 4th-last H2D's start to the 4th-last D2H's end would cut the drain;
    # simpler and conservative: the span in which copies of both directions exist
    lo = max(min(s for s, _ in h2d), min(s for s, _ in d2h))
    hi = min(max(e for _, e in h2d), max(e for _, e in d2h))
    clip = lambda iv: [(max(s, lo), min(e, hi)) for s, e in iv if e > lo and s < hi]  # noqa: E731
    span = hi - lo
    for name, iv in (('kernels', kern), ('host-to-device', h2d), ('device-to-host', d2h)):
        print('%-16s busy %7.2f ms of %7.2f ms = %5.1f %%' % (name, union(clip(iv)) / 1e6, span / 1e6, 100.0 * union(clip(iv)) / span))
    both = union(clip(h2d)) + union(clip(d2h)) + union(clip(kern))
    print('sum of the three busy times = %.2f x the window: that much runs concurrently' % (both / span))


if __name__ == '__main__':
    main(sys.argv[1])
