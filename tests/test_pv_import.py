"""Structural import of the reference model container; needs the reference checkout (skipped on the GPU box)."""
import os

import numpy as np
import pytest

from koala_amd import params, pv_import

PV = '/root/reference/lib/common/koala_params.pv'
pytestmark = pytest.mark.skipif(not os.path.exists(PV), reason='reference checkout not present')


def test_block_directory_matches_the_kns_topology():
    m = pv_import.read_pv(PV)
    assert m.version == '3.0.0' and len(m.blocks) == 20
    shapes = [(b.rows, b.cols) for b in m.blocks]
    assert shapes[0] == (271, 813) and shapes[5] == (272, 813) and shapes[10] == (276, 813) and shapes[15] == (311, 813)
    assert [shapes[i][1] for i in (4, 9, 14, 19)] == [1, 5, 40, 257]
    # offsets observed in SURVEY.md Appendix B
    assert m.blocks[0].offset == 349573 and m.blocks[4].offset == 1234149 and m.blocks[19].offset == 3937829
    # int8 payload statistics: symmetric, full range
    w = m.blocks[1].weights.astype(np.float64)
    assert abs(w.mean()) < 2 and 40 < w.std() < 80 and w.min() <= -120 and w.max() >= 120
    assert -5400 < m.table_mean.min() and m.table_mean.max() < -2900 and 800 < m.table_scale.min()


def test_structural_import_writes_a_loadable_kns1(tmp_path):
    t = pv_import.to_kns1(pv_import.read_pv(PV))
    p = str(tmp_path / 'imported.kns')
    params.write_params(p, t)
    from oracle import oracle
    o = oracle.Oracle(p, 1)
    out = o.process(np.zeros(256 * 4, np.int16))
    assert out.shape == (1024,)
