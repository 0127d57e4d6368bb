"""Summary of a rocprofv3 kernel trace (rocpd .db) of tools/t1_trace.py: per kernel name the mean duration and the mean idle
time before it (start minus the previous kernel's end), over the last 200 frame steps."""
import collections
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = 'kernels' if 'kernels' in tables else [t for t in tables if 'kernel_dispatch' in t][0]
    cols = [r[1] for r in c.execute('pragma table_info(%s)' % view)]
    name = 'name' if 'name' in cols else 'kernel_name'
    rows = c.execute('select %s, start, end from %s order by start' % (name, view)).fetchall()
    rows = [r for r in rows if r[0].startswith('kns::') or 'kns::' in r[0]]
    rows = rows[len(rows) // 3:]
    dur = collections.defaultdict(list)
    gap = collections.defaultdict(list)
    for prev, cur in zip(rows, rows[1:]):
        dur[cur[0]].append(cur[2] - cur[1])
        gap[cur[0]].append(cur[1] - prev[2])
    tot_d = tot_g = 0.0
    n_steps = sum(1 for r in rows if 'analysis' in r[0])
    print('%-64s %8s %10s %10s' % ('kernel', 'per step', 'exec us', 'idle before us'))
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        d, g = sum(dur[k]) / len(dur[k]) / 1e3, sum(gap[k]) / len(gap[k]) / 1e3
        per = len(dur[k]) / max(1, n_steps)
        tot_d += d * per
        tot_g += g * per
        print('%-64s %8.1f %10.2f %10.2f' % (k[:64], per, d, g))
    print('per frame step: %.1f us executing + %.1f us idle between kernels' % (tot_d, tot_g))


if __name__ == '__main__':
    main(sys.argv[1])
