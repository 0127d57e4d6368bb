"""Instruction mix of a kernel's main loop, read off its gfx950 assembly (developer tool; the numbers DESIGN.md quotes for the
recurrent step -- MFMAs, transcendentals, moves, LDS reads, waits per wave-step -- come from here, so that a claimed instruction
diet can be checked against the compiler's output and against SQ_INSTS_VALU of the PMC runs).
usage: python tools/isa_count.py [--flags "<extra hipcc flags>"] koala_amd/csrc/kns_gru.hip gru_resident8_kernel"""
import collections
import os
import re
import subprocess
import sys
import tempfile

HIPCC = '/opt/rocm/bin/hipcc'
CLASSES = [('mfma', r'v_mfma'), ('transcendental', r'v_(exp|rcp|log|rsq|sqrt|sin|cos)_'), ('v_mov', r'v_mov_|v_accvgpr'),
           ('packed f32', r'v_pk_(fma|add|mul)_f32'), ('other VALU', r'v_'), ('LDS read', r'ds_read'), ('LDS write', r'ds_write'),
           ('vector memory', r'(buffer|global)_(load|store|atomic)'), ('s_waitcnt', r's_waitcnt'), ('s_nop', r's_nop'),
           ('barrier', r's_barrier'), ('other scalar', r's_')]


def loop_of(text, kernel):
    m = re.search(r'^(\S*%s\S*):' % re.escape(kernel), text, re.M)
    if not m:
        raise SystemExit('no kernel matching %r' % kernel)
    body = text[m.end():text.index('s_endpgm', m.end())].split('\n')
    heads = [i for i, l in enumerate(body) if 'Loop Header: Depth=1' in l]
    if not heads:
        return m.group(1), body
    # the LONGEST depth-1 loop: label ... last branch back to it
    best = None
    for i0 in heads:
        lab = body[i0].split(':')[0].strip()
        back = [i for i, l in enumerate(body) if i > i0 and re.search(r's_(c?branch)\S*\s+%s\b' % re.escape(lab), l)]
        if back and (best is None or back[-1] - i0 > best[1] - best[0]):
            best = (i0, back[-1])
    return m.group(1), body[best[0]:best[1] + 1] if best else body


def count(lines):
    c, ops = collections.Counter(), collections.Counter()
    for l in lines:
        l = l.strip()
        if not l or l.startswith((';', '.')) or l.endswith(':'):
            continue
        op = l.split()[0]
        ops[op] += 1
        for name, pat in CLASSES:
            if re.match(pat, op):
                c[name] += 1
                break
        else:
            c['other'] += 1
    return c, ops


def main():
    args = sys.argv[1:]
    flags = []
    if args and args[0] == '--flags':
        flags, args = args[1].split(), args[2:]
    src, kernel = args
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off'] + flags +
                              ['-S', '--cuda-device-only', '-x', 'hip', src, '-o', out], stderr=subprocess.DEVNULL)
        text = open(out).read()
    name, lines = loop_of(text, kernel)
    c, ops = count(lines)
    print('%s: main loop, %d instructions (both arms of wave-uniform branches are counted)' % (name, sum(c.values())))
    for k, _ in CLASSES + [('other', '')]:
        if c[k]:
            print('  %-16s %4d' % (k, c[k]))
    print('  most frequent:', ', '.join('%s %d' % kv for kv in ops.most_common(14)))
    m = re.search(re.escape(name) + r'\.num_vgpr, (\d+)', text)
    s = re.search(re.escape(name) + r'\.private_seg_size, (\d+)', text)
    print('  VGPRs %s, scratch %s B' % (m.group(1) if m else '?', s.group(1) if s else '?'))


if __name__ == '__main__':
    main()
