"""Reader + hypothesis-driven import of the reference model container; needs the reference checkout (skipped on the GPU box)."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from koala_amd import params, pv_import

PV = '/root/reference/lib/common/koala_params.pv'
pytestmark = pytest.mark.skipif(not os.path.exists(PV), reason='reference checkout not present')


def test_every_byte_of_the_file_is_accounted_for():
    m = pv_import.read_pv(PV)
    assert m.version == '3.0.0' and len(m.blocks) == 20 and m.unread_bytes == 0
    shapes = [(b.rows, b.cols) for b in m.blocks]
    assert shapes[0] == (271, 813) and shapes[5] == (272, 813) and shapes[10] == (276, 813) and shapes[15] == (311, 813)
    assert [shapes[i][1] for i in (4, 9, 14, 19)] == [1, 5, 40, 257]
    # offsets observed in SURVEY.md Appendix B
    assert m.blocks[0].offset == 349573 and m.blocks[4].offset == 1234149 and m.blocks[19].offset == 3937829
    # the front-end is a linear layer over five stacked feature frames
    assert m.front_header == (2, 2, 4, 5 * 257, 271) and m.front.weights.shape == (1285, 271)
    assert m.stage_tail == [3056, 1013, 1379, 1713]
    # int8 payload statistics: symmetric, full range; the per-column vectors behind the matrices are small signed values
    w = m.blocks[1].weights.astype(np.float64)
    assert abs(w.mean()) < 2 and 40 < w.std() < 80 and w.min() <= -120 and w.max() >= 120
    for b in m.blocks[:4]:
        assert abs(b.trailer.astype(np.float64).mean()) < 8 and b.trailer.astype(np.float64).std() < 70
    assert -5400 < m.table_mean.min() and m.table_mean.max() < -2900 and 800 < m.table_scale.min()


def test_import_maps_all_records_and_writes_a_loadable_kns1(tmp_path):
    m = pv_import.read_pv(PV)
    t = pv_import.to_kns1(m)
    # nothing is left at its zero initialisation: front-end, 16 GRU matrices with biases, 4 heads with biases
    for name, _ in params.tensor_order():
        assert np.any(t[name] != 0), name
    # hypothesis fields do what they say
    h = pv_import.Hypothesis(gate_order='zrn', y_first=False, weight_shift=6)
    t2 = pv_import.to_kns1(m, h)
    blk = m.blocks[5]  # stage 2, W_ih_a [272, 813]: file rows [e ; y]
    assert np.allclose(t2['s1.w_ih_a'][0, :271], blk.weights[271, 271:542] / 64.0)      # y row first, r columns = file group 1
    assert np.allclose(t2['s1.w_ih_a'][1, 271:542], blk.weights[0, 0:271] / 64.0)       # z columns = file group 0
    p = str(tmp_path / 'imported.kns')
    params.write_params(p, t)
    from oracle import oracle
    out = oracle.Oracle(p, 1).process(np.zeros(256 * 4, np.int16))
    assert out.shape == (1024,)
    # the full import (all five front-end taps) as a KNS-v1.1 file under build/ (git-ignored; it travels to the GPU box with the
    # working tree), where tests/test_gpu_front_taps.py runs it on the GPU against the oracle
    full = os.path.join(ROOT, 'build', 'imported_pv_default.kns')
    os.makedirs(os.path.dirname(full), exist_ok=True)
    params.write_params(full, pv_import.to_kns1(m, pv_import.Hypothesis(front_tap=5)))
    assert params.read_params(full)['w_in'].shape == (5 * 257, 271)


def test_hypothesis_search_is_recorded_as_measured(test_pcm, noise_pcm, tmp_path):
    """profiles/r02_pv_import_search.json (tools/pv_hypotheses.py): no reading of the bytes tried so far behaves like the
    reference on its own fixtures.  The test re-measures the default hypothesis so the record cannot drift from the code."""
    rec = json.load(open(os.path.join(ROOT, 'profiles', 'r02_pv_import_search.json')))
    assert rec['hypotheses_tried'] >= 1000
    assert rec['passing'] == sum(1 for e in rec['best'] if max(e['metrics'][k] for k in ('speech', 'noise', 'mixed')) < 0.02)
    print('imported-model envelope search: tried %d, passing %d, best score %.3f, median %.3f'
          % (rec['hypotheses_tried'], rec['passing'], rec['score_quantiles']['0.0'], rec['score_quantiles']['0.5']))
    from oracle import oracle
    n = len(test_pcm) // 256 * 256
    x = np.stack([test_pcm[:n], noise_pcm[:n], (test_pcm[:n].astype(int) + noise_pcm[:n]).astype(np.int16)])
    p = str(tmp_path / 'default.kns')
    params.write_params(p, pv_import.to_kns1(pv_import.read_pv(PV)))
    y = oracle.Oracle(p, 3).process(x)

    def rms(a):
        return np.sqrt(np.mean((a.astype(np.float64) / 32768.0) ** 2, axis=-1))
    fo, ref = rms(y.reshape(3, -1, 256)), rms(x[0].reshape(-1, 256))
    got = {'speech': np.abs(fo[0][1:] - ref[:-1]).max(), 'noise': fo[1].max(), 'mixed': np.abs(fo[2][1:] - ref[:-1]).max()}
    for k, v in got.items():
        assert abs(v - rec['default_hypothesis']['metrics'][k]) < 2e-3, (k, v)
