"""
Structural reader of the reference's model container `lib/common/koala_params.pv`.

What can be recovered from the bytes alone -- and is recovered here: the product/version magic, the two 257-entry
int16 feature tables, and the directory of int8 weight blocks `[rows, cols] + rows*cols int8 + per-column trailer`
whose shapes are exactly the KNS-v1 topology (4 x {[271+d, 813], [271, 813] x 3, [271, head]}).
What cannot: the fixed-point conventions (activation Q-formats, the meaning of the per-column trailer bytes, where the
biases and the ~350 KB front-end live), because the engine that interprets them is closed and cannot be run without an
AccessKey.  `to_kns1()` therefore only produces a STRUCTURAL import (int8 / 128, trailers ignored): it has the
reference's capacity and sparsity pattern, not its behaviour, and is meant for sizing/throughput work and as the
starting point for a key holder who can fit the conventions against `tools/ref_capture.py` output.
"""

import struct
from typing import Dict, List, NamedTuple

import numpy as np

from . import params

MAGIC = b'koala3.0.0'


class Block(NamedTuple):
    offset: int          # of the int32 rows field
    rows: int
    cols: int
    weights: np.ndarray  # int8 [rows, cols]
    trailer: np.ndarray  # uint8 [cols]


class PvModel(NamedTuple):
    version: str
    table_mean: np.ndarray   # int16 [257]
    table_scale: np.ndarray  # int16 [257]
    blocks: List[Block]


def read_pv(path: str) -> PvModel:
    b = open(path, 'rb').read()
    if b[:len(MAGIC)] != MAGIC:
        raise ValueError('not a koala 3.0.0 model file: %r' % b[:10])
    t1 = np.frombuffer(b[15:15 + 514], '<i2').copy()
    t2 = np.frombuffer(b[529:529 + 514], '<i2').copy()
    blocks = []
    expect = []
    for s in range(params.STAGES):
        d_in = params.HEADS[s - 1] if s else 0
        expect += [(d_in + params.HIDDEN, params.G3)] + [(params.HIDDEN, params.G3)] * 3 + [(params.HIDDEN, params.HEADS[s])]
    pos = 1043
    for rows, cols in expect:
        pat = struct.pack('<2i', rows, cols)
        o = b.find(pat, pos)
        if o < 0:
            raise ValueError('block [%d, %d] not found after offset %d' % (rows, cols, pos))
        payload = np.frombuffer(b[o + 8:o + 8 + rows * cols], np.int8).reshape(rows, cols).copy()
        trailer = np.frombuffer(b[o + 8 + rows * cols:o + 8 + rows * cols + cols], np.uint8).copy()
        blocks.append(Block(o, rows, cols, payload, trailer))
        pos = o + 8 + rows * cols
    return PvModel(b[5:10].decode(), t1, t2, blocks)


def to_kns1(model: PvModel) -> Dict[str, np.ndarray]:
    """Structural import only (see the module docstring): int8 / 128 weights in KNS1 tensor order, zero biases, feature
    tables rescaled to the magnitude KNS-v1 features have.  Row order of the stage-input matrices is kept as found."""
    t = {name: np.zeros(shape, np.float32) for name, shape in params.tensor_order()}
    t['mean'][:] = model.table_mean.astype(np.float32) / 512.0
    t['scale'][:] = model.table_scale.astype(np.float32) / 4096.0
    eye = np.zeros((params.BINS, params.HIDDEN), np.float32)
    eye[np.arange(params.BINS), np.arange(params.BINS)] = 1.0
    t['w_in'][:] = eye  # the reference's front-end (~350 KB) is not decoded
    it = iter(model.blocks)
    for s in range(params.STAGES):
        for name in ('w_ih_a', 'w_hh_a', 'w_ih_b', 'w_hh_b', 'w_head'):
            blk = next(it)
            t['s%d.%s' % (s, name)][:] = blk.weights.astype(np.float32) / 128.0
    return t


__all__ = ['read_pv', 'to_kns1', 'PvModel', 'Block']
