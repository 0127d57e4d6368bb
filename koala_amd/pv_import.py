"""
Reader and HYPOTHESIS-DRIVEN importer of the reference's model container `lib/common/koala_params.pv`
(SURVEY.md 8f row 1, Appendix B).

What the bytes say (observations, all checked in tests/test_pv_import.py):
  * magic `koala3.0.0`, 5 bytes `01 01 01 00 00`, two 257-entry int16 tables (per-bin feature mean and scale);
  * front-end record at 1043: int32 {2, 2, 4, 1285, 271}, then 1285 x 271 int8 and 271 int8 -- 1285 = 5 x 257, i.e. a
    linear layer over FIVE stacked feature frames (a 5-frame context window) with a per-output int8 vector;
  * per stage: int32 2, then blocks `int32 rows, int32 cols, rows x cols int8, cols int8`, with (rows, cols) =
    (271 + d, 813), (271, 813) x 3, (271, head) -- two GRU layers (W_ih, W_hh each with a per-column vector, which is
    what a bias looks like: small signed values) and the head -- followed by one int16 per stage (3056, 1013, 1379, 1713);
  * 2 trailing bytes are that int16 of the last stage: the file is fully accounted for.
What they do not say: the fixed-point conventions -- how int8 weights, int8 biases and int16 activations scale to real
numbers, the gate order inside the 813 columns, the row order of the stage inputs, the log base and Q-format of the
feature tables, what the per-stage int16 does.  The engine that knows is closed and cannot be run without an AccessKey.

`to_kns1(model, hyp)` therefore maps ALL records into KNS1 tensors under an explicit `Hypothesis` (every guess is a named
field), and `tools/pv_hypotheses.py` scores hypotheses by the only behavioural contract the reference publishes, the
acceptance envelope on its own WAV fixtures (binding/python/test_koala.py:71-114).  KNS-v1's front-end sees one frame, so
the 5-frame context is reduced to one tap or to the sum of the taps (hypothesis field `front_tap`): that alone makes an
exact behavioural import impossible without extending the spec.  Nothing here claims parity; the search result is
recorded as measured (profiles/r02_pv_import_search.json).
"""

import struct
from typing import Dict, List, NamedTuple

import numpy as np

from . import params

MAGIC = b'koala3.0.0'
FRONT_OFFSET = 1043
FRONT_TAPS = 5


class Block(NamedTuple):
    offset: int          # of the int32 rows field
    rows: int
    cols: int
    weights: np.ndarray  # int8 [rows, cols]
    trailer: np.ndarray  # int8 [cols]: per-column vector stored behind the matrix (read as a bias)


class PvModel(NamedTuple):
    version: str
    table_mean: np.ndarray   # int16 [257]
    table_scale: np.ndarray  # int16 [257]
    front: Block             # [1285, 271]
    blocks: List[Block]      # 20: per stage W_ih_a, W_hh_a, W_ih_b, W_hh_b, W_head
    stage_tail: List[int]    # one int16 per stage, stored behind its head block
    front_header: tuple      # the five int32 in front of the front-end matrix
    unread_bytes: int        # bytes of the file no record accounts for (0)


class Hypothesis(NamedTuple):
    """Every convention the bytes do not fix.  Defaults = the most conventional reading."""
    weight_shift: int = 7        # W = int8 * 2^-weight_shift
    bias_shift: int = 5          # b = int8 * 2^-bias_shift
    front_shift: int = 7         # front-end weights
    front_bias_shift: int = 5
    front_tap: int = 4           # which of the 5 stacked frames KNS-v1's one-frame front-end keeps (0..4), -1 = their sum,
                                 # 5 = ALL five, file order = oldest first; 6 = all five, file order = newest first (both give a
                                 # 5-frame front-end that only the CPU oracle can run: oracle/kns_oracle.h)
    mean_div: float = 512.0      # feature mean = table / mean_div ...
    scale_div: float = 4096.0    # ... feature scale = table / scale_div
    log2_features: bool = False  # tables in log2 units: mean and 1/scale are multiplied by ln 2 to reach KNS-v1's ln
    gate_order: str = 'rzn'      # order of the three 271-column groups in a 813-column GRU matrix
    y_first: bool = True         # stage-input rows ordered [y_prev ; e] (False: [e ; y_prev])
    head_shift: int = 7
    head_bias_shift: int = 5
    # round 4: the per-stage int16 behind each head block (3056, 1013, 1379, 1713) as a fixed-point multiplier c = value / 2^tail_q
    # (the reference ships exactly one "elementwise with int16 scalar" kernel, SURVEY.md 2.2 `taabe194`):
    #   'none'      not used (rounds 2-3)
    #   'preact'    scales the head's pre-activation: y_s = sigmoid(c (h W + b))
    #   'out'       scales the head's output where the next stage consumes it: the y rows of stage s + 1's W_ih (the last stage's, the
    #               mask, cannot be folded into KNS1 and is left alone)
    #   'embed'     scales the embedding share of the NEXT stage's input (the e rows of stage s + 1's W_ih)
    tail_mode: str = 'none'
    tail_q: int = 12
    act_q: int = 0               # oracle-only: GEMM outputs re-quantised to saturating int16 with act_q fractional bits (0 = off)


def _block(b: bytes, o: int, rows: int, cols: int) -> Block:
    r, c = struct.unpack('<2i', b[o:o + 8])
    if (r, c) != (rows, cols):
        raise ValueError('block [%d, %d] expected at offset %d, found [%d, %d]' % (rows, cols, o, r, c))
    w = np.frombuffer(b[o + 8:o + 8 + rows * cols], np.int8).reshape(rows, cols).copy()
    t = np.frombuffer(b[o + 8 + rows * cols:o + 8 + rows * cols + cols], np.int8).copy()
    return Block(o, rows, cols, w, t)


def read_pv(path: str) -> PvModel:
    b = open(path, 'rb').read()
    if b[:len(MAGIC)] != MAGIC:
        raise ValueError('not a koala 3.0.0 model file: %r' % b[:10])
    t1 = np.frombuffer(b[15:15 + 514], '<i2').copy()
    t2 = np.frombuffer(b[529:529 + 514], '<i2').copy()
    hdr = struct.unpack('<5i', b[FRONT_OFFSET:FRONT_OFFSET + 20])
    if hdr[3:] != (FRONT_TAPS * params.BINS, params.HIDDEN):
        raise ValueError('unexpected front-end record %r' % (hdr,))
    front = _block(b, FRONT_OFFSET + 12, FRONT_TAPS * params.BINS, params.HIDDEN)
    pos = FRONT_OFFSET + 20 + front.rows * front.cols + front.cols
    blocks, tails = [], []
    for s in range(params.STAGES):
        d_in = params.HEADS[s - 1] if s else 0
        (two,) = struct.unpack('<i', b[pos:pos + 4])
        if two != 2:
            raise ValueError('stage %d does not start with int32 2 at offset %d' % (s, pos))
        pos += 4
        for rows, cols in [(d_in + params.HIDDEN, params.G3)] + [(params.HIDDEN, params.G3)] * 3 + [(params.HIDDEN, params.HEADS[s])]:
            blk = _block(b, pos, rows, cols)
            blocks.append(blk)
            pos += 8 + rows * cols + cols
        tails.append(struct.unpack('<h', b[pos:pos + 2])[0])
        pos += 2
    return PvModel(b[5:10].decode(), t1, t2, front, blocks, tails, hdr, len(b) - pos)


def to_kns1(model: PvModel, hyp: Hypothesis = Hypothesis()) -> Dict[str, np.ndarray]:
    """All 21 records of the reference file as KNS1 tensors under `hyp` (see the module docstring: a hypothesis, not parity)."""
    taps = FRONT_TAPS if hyp.front_tap >= FRONT_TAPS else 1
    t = {name: np.zeros(shape, np.float32) for name, shape in params.tensor_order(taps)}
    ln2 = float(np.log(2.0))
    mean = model.table_mean.astype(np.float64) / hyp.mean_div
    scale = model.table_scale.astype(np.float64) / hyp.scale_div
    if hyp.log2_features:  # (log2 P - m) s = (ln P - m ln 2) (s / ln 2)
        mean, scale = mean * ln2, scale / ln2
    t['mean'][:] = mean
    t['scale'][:] = scale
    fw = model.front.weights.astype(np.float64).reshape(FRONT_TAPS, params.BINS, params.HIDDEN) * 2.0 ** -hyp.front_shift
    if hyp.front_tap >= FRONT_TAPS:
        t['w_in'][:] = (fw if hyp.front_tap == FRONT_TAPS else fw[::-1]).reshape(FRONT_TAPS * params.BINS, params.HIDDEN)
    else:
        t['w_in'][:] = fw.sum(axis=0) if hyp.front_tap < 0 else fw[hyp.front_tap]
    t['b_in'][:] = model.front.trailer.astype(np.float64) * 2.0 ** -hyp.front_bias_shift
    perm = np.concatenate([np.arange(params.HIDDEN) + hyp.gate_order.index(gate) * params.HIDDEN for gate in 'rzn'])
    it = iter(model.blocks)
    for s in range(params.STAGES):
        d_in = params.HEADS[s - 1] if s else 0
        for name in ('ih_a', 'hh_a', 'ih_b', 'hh_b'):
            blk = next(it)
            w = blk.weights.astype(np.float64) * 2.0 ** -hyp.weight_shift
            if name == 'ih_a' and d_in and not hyp.y_first:
                w = np.concatenate([w[params.HIDDEN:], w[:params.HIDDEN]], axis=0)  # file [e ; y] -> KNS1 [y ; e]
            t['s%d.w_%s' % (s, name)][:] = w[:, perm]
            t['s%d.b_%s' % (s, name)][:] = (blk.trailer.astype(np.float64) * 2.0 ** -hyp.bias_shift)[perm]
        blk = next(it)
        t['s%d.w_head' % s][:] = blk.weights.astype(np.float64) * 2.0 ** -hyp.head_shift
        t['s%d.b_head' % s][:] = blk.trailer.astype(np.float64) * 2.0 ** -hyp.head_bias_shift
    if hyp.tail_mode != 'none':
        for s in range(params.STAGES):
            c = float(model.stage_tail[s]) / 2.0 ** hyp.tail_q
            d_out = params.HEADS[s]
            if hyp.tail_mode == 'preact':
                t['s%d.w_head' % s] *= c
                t['s%d.b_head' % s] *= c
            elif s + 1 < params.STAGES:  # (the y rows come first in KNS1's stage input [y_prev ; e])
                rows = slice(0, d_out) if hyp.tail_mode == 'out' else slice(d_out, d_out + params.HIDDEN)
                t['s%d.w_ih_a' % (s + 1)][rows] *= c
    if hyp.act_q:
        t['__act_q__'] = np.int32(hyp.act_q)
    return t


__all__ = ['read_pv', 'to_kns1', 'PvModel', 'Block', 'Hypothesis']
