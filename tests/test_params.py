import numpy as np
import pytest

from koala_amd import params


def test_roundtrip(tmp_path):
    t = params.make_random(7)
    p = str(tmp_path / 'm.kns')
    params.write_params(p, t)
    back = params.read_params(p)
    assert set(back) == set(t)
    for k in t:
        assert np.array_equal(back[k], t[k])


def test_topology_matches_reference_model_blocks():
    # lib/common/koala_params.pv block headers: [271|272|276|311, 813], heads 1/5/40/257
    shapes = dict(params.tensor_order())
    assert [shapes['s%d.w_ih_a' % s][0] for s in range(4)] == [271, 272, 276, 311]
    assert all(shapes['s%d.w_hh_a' % s] == (271, 813) for s in range(4))
    assert [shapes['s%d.w_head' % s][1] for s in range(4)] == [1, 5, 40, 257]
    macs = sum(int(np.prod(s)) for n, s in params.tensor_order() if len(s) == 2)
    assert macs == 3714326


def test_seeded_random_is_reproducible():
    a, b = params.make_random(5), params.make_random(5)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    c = params.make_random(6)
    assert not np.array_equal(a['w_in'], c['w_in'])


def test_bad_files_are_rejected(tmp_path):
    p = tmp_path / 'x.kns'
    p.write_bytes(b'koala3.0.0' + b'\0' * 100)
    with pytest.raises(ValueError):
        params.read_params(str(p))
