"""Build-time guard (no GPU needed: hipcc cross-compiles): the gfx950 assembly of the kernels must not contain the
store-data hazard that corrupted mask values in round 1 (a 16-byte store with an SGPR address part directly followed by
a VALU write of its upper data registers; tools/isa_scan.py), and must not spill."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not available')
@pytest.mark.parametrize('src', ['kns_stft.hip', 'kns_gemm.hip', 'kns_gru.hip', 'kns_gruq.hip'])
def test_no_store_data_hazard_and_no_spills(src, tmp_path):
    import isa_scan
    out = tmp_path / (src + '.s')
    # the per-file flags of koala_amd/Makefile (FLAGS_<stem> = ...)
    mk = open(os.path.join(ROOT, 'koala_amd', 'Makefile')).read()
    m = re.search(r'^FLAGS_%s\s*=\s*(.*)$' % src.split('.')[0], mk, re.M)
    extra = m.group(1).split() if m else []
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off'] + extra +
                          ['-S', '--cuda-device-only', '-x', 'hip', os.path.join(ROOT, 'koala_amd', 'csrc', src), '-o', str(out)],
                          stderr=subprocess.DEVNULL)
    text = out.read_text()
    assert isa_scan.scan(text) == []
    known = ()
    spills = {}
    found = re.findall(r'\.set (\S+)\.has_indirect_call, \d+\n[^\n]*\n; Kernel info:\n(?:;[^\n]*\n)*?; ScratchSize: (\d+)', text)
    assert found, 'no kernel resource summaries in the assembly'
    for name, size in found:
        # (up to 16 bytes are tolerated: hipcc parks a prologue value needed again only in the epilogue, outside every loop)
        if int(size) > 16 and not any(k in name for k in known):
            spills[name] = int(size)
    assert not spills, 'kernels spill registers to scratch: %r' % spills
