"""Scans gfx950 assembly (hipcc -S --cuda-device-only) for the store-data hazard met in round 1: a 12/16-byte vector
store whose address carries an SGPR part (`buffer_store ... sN offen`, `global_store ... s[a:b]`) reads its upper data
dwords late, and hipcc 7.2 may schedule a VALU write of those registers directly behind it.
Usage: python tools/isa_scan.py file.s [...]   -- prints every suspicious pair, exit status 1 if there is any."""
import re
import sys


def scan(text):
    lines = [l.strip() for l in text.splitlines() if l.strip() and not l.strip().startswith((';', '.'))]
    found = []
    for i, l in enumerate(lines):
        m = re.match(r'(global_store_dwordx[34]|buffer_store_dwordx[34])\s+(.*)', l)
        if not m:
            continue
        if m.group(1).startswith('global'):
            mm = re.match(r'v(?:\[\d+:\d+\]|\d+), v\[(\d+):(\d+)\], (s\[\d+:\d+\]|off)', m.group(2))
            if not mm or mm.group(3) == 'off':
                continue
        else:
            mm = re.match(r'v\[(\d+):(\d+)\], v\d+, s\[\d+:\d+\], (\S+)', m.group(2))
            if not mm or not mm.group(3).startswith('s'):
                continue
        lo, hi = int(mm.group(1)), int(mm.group(2))
        for nx in lines[i + 1:i + 3]:
            if nx.startswith(('s_nop', 's_waitcnt')):
                break
            d = re.match(r'v_\S+\s+v(?:\[(\d+):(\d+)\]|(\d+))', nx)
            if d:
                a, b = int(d.group(1) or d.group(3)), int(d.group(2) or d.group(3))
                if a <= hi and b >= lo + 1:  # any dword after the first
                    found.append((l, nx))
    return found


if __name__ == '__main__':
    bad = 0
    for path in sys.argv[1:]:
        for st, nx in scan(open(path).read()):
            print('%s: %s  ->  %s' % (path, st, nx))
            bad += 1
    sys.exit(1 if bad else 0)
