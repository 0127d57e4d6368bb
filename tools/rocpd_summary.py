"""Turns a rocprofv3 rocpd database (…_results.db) into the text summary kept under profiles/ (kernel stats and,
when present, PMC counter sums per kernel)."""
import sqlite3
import sys


def main(path, out=None):
    c = sqlite3.connect(path)
    lines = []
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels "
                     "order by total_duration desc").fetchall()
    lines.append('%-72s %8s %14s %12s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
    for name, calls, total, avg, pct in rows:
        lines.append('%-72s %8d %14d %12.0f %7.2f' % (name[:72], calls, total, avg, pct))
    try:
        pmc = c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                        "group by kernel_name, counter_name").fetchall()
    except Exception:
        pmc = []
    if pmc:
        lines.append('')
        lines.append('%-60s %-24s %18s %8s' % ('kernel', 'counter', 'sum over dispatches', 'dispatches'))
        for name, ctr, val, n in pmc:
            lines.append('%-60s %-24s %18.1f %8d' % (name[:60], ctr, val, n))
    text = '\n'.join(lines) + '\n'
    if out:
        open(out, 'w').write(text)
    print(text)


if __name__ == '__main__':
    main(*sys.argv[1:3])
