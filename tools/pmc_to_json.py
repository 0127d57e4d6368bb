"""Condenses the per-pass text summaries written by tools/pmc_run.sh into one JSON: per kernel, per-dispatch averages of
every counter, plus HBM traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB -- FETCH_SIZE is doubled because this rocprofv3
reports exactly half of a wide coalesced streaming read on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is
taken as reported."""
import collections
import json
import re
import sys

CLASS = {'analysis_kernel': 'analysis', 'gemm_ws2_kernel<0, 4>': 'gemm_input', 'gru_resident8_kernel': 'gru_recurrent',
         'synthesis_kernel': 'synthesis', 'gemm_wsr_kernel<2, 2>': 'gemm_head'}


def main(pmc_dir, out, commit=None):
    d = collections.defaultdict(dict)
    for i in range(1, 9):
        try:
            txt = open('%s/pass%d.txt' % (pmc_dir, i)).read()
        except IOError:
            continue
        if 'sum over dispatches' not in txt:
            continue
        for l in txt.split('sum over dispatches')[1].split('\n')[1:]:
            m = re.match(r'^(.{60}) (\S+)\s+([\d.]+)\s+(\d+)$', l)
            if m:
                d[m.group(1).strip()][m.group(2)] = float(m.group(3)) / int(m.group(4))
    res = {}
    for k, v in d.items():
        entry = {'counters_per_dispatch': {c: round(x, 1) for c, x in sorted(v.items())}}
        if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            entry['hbm_read_bytes'] = int(2 * v['FETCH_SIZE'] * 1024)
            entry['hbm_write_bytes'] = int(v['WRITE_SIZE'] * 1024)
            entry['hbm_bytes'] = entry['hbm_read_bytes'] + entry['hbm_write_bytes']
        for frag, cls in CLASS.items():
            if frag in k:
                entry['class'] = cls
        res[k] = entry
    # the workload the counters belong to (from the bench line of the first pass): bench.py only quotes them for the same one
    try:
        line = [l for l in open('%s/pass1.json' % pmc_dir).read().splitlines() if l.startswith('{')][-1]
        cfg = json.loads(line)['config']
        res['_workload'] = {'streams_per_gpu': cfg['streams_per_gpu'], 'frames_per_call': cfg['frames_per_call'],
                            'dtype': json.loads(line)['dtype']}
    except Exception:
        pass
    if commit:
        res['_commit'] = commit  # source revision the counters were collected on (bench.py reports it as traffic_commit)
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
    for k, v in res.items():
        if isinstance(v, dict) and 'class' in v:
            print(v['class'], v.get('hbm_bytes'))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
