"""
KNS1 parameter files: the build's own model container.

The reference ships its model as the closed `lib/common/koala_params.pv` (magic `koala3.0.0`, two 257-entry
int16 tables, int8 weight blocks `[271|272|276|311, 813]`, `[271, 813] x 3` per stage and heads
`[271, 1|5|40|257]`).  KNS1 keeps exactly that topology in fp32, little endian:

    magic "KNS1\\0\\0\\0\\0", 14 x u32 {version=1, n_fft=512, hop=256, bins=257, hidden=271, stages=4,
    head[4]={1,5,40,257}, delay_sample=256, 0, 0, 0}
    mean[257] scale[257] w_in[257][271] b_in[271]
    per stage s:  w_ih_a[d_in+271][813] b_ih_a[813] w_hh_a[271][813] b_hh_a[813]
                  w_ih_b[271][813]      b_ih_b[813] w_hh_b[271][813] b_hh_b[813]
                  w_head[271][head[s]]  b_head[head[s]]
    (d_in = head[s-1], 0 for s = 0; w_ih_a rows ordered [y_prev ; e]; 813 columns ordered r|z|n)

This module is host-side tooling (numpy only): it writes/reads the container and synthesises parameter sets
(seeded random for throughput work, a hand-built spectral gate for the acceptance-envelope tests).
"""

import os
import struct
from typing import Dict, Optional

import numpy as np

N_FFT = 512
HOP = 256
BINS = 257
HIDDEN = 271
STAGES = 4
HEADS = (1, 5, 40, 257)
DELAY = 256
G3 = 3 * HIDDEN
MAGIC = b"KNS1\0\0\0\0"


def tensor_order():
    names = [("mean", (BINS,)), ("scale", (BINS,)), ("w_in", (BINS, HIDDEN)), ("b_in", (HIDDEN,))]
    for s in range(STAGES):
        d_in = HEADS[s - 1] if s else 0
        names += [
            ("s%d.w_ih_a" % s, (d_in + HIDDEN, G3)), ("s%d.b_ih_a" % s, (G3,)),
            ("s%d.w_hh_a" % s, (HIDDEN, G3)), ("s%d.b_hh_a" % s, (G3,)),
            ("s%d.w_ih_b" % s, (HIDDEN, G3)), ("s%d.b_ih_b" % s, (G3,)),
            ("s%d.w_hh_b" % s, (HIDDEN, G3)), ("s%d.b_hh_b" % s, (G3,)),
            ("s%d.w_head" % s, (HIDDEN, HEADS[s])), ("s%d.b_head" % s, (HEADS[s],)),
        ]
    return names


def write_params(path: str, tensors: Dict[str, np.ndarray]) -> None:
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<14I", 1, N_FFT, HOP, BINS, HIDDEN, STAGES, *HEADS, DELAY, 0, 0, 0))
        for name, shape in tensor_order():
            a = np.ascontiguousarray(tensors[name], dtype="<f4")
            if a.shape != shape:
                raise ValueError("tensor %s has shape %s, expected %s" % (name, a.shape, shape))
            f.write(a.tobytes())
    os.replace(tmp, path)


def read_params(path: str) -> Dict[str, np.ndarray]:
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("not a KNS1 file: %s" % path)
        hdr = struct.unpack("<14I", f.read(56))
        if hdr[:10] != (1, N_FFT, HOP, BINS, HIDDEN, STAGES) + HEADS:
            raise ValueError("unsupported KNS1 dims: %r" % (hdr,))
        out = {}
        for name, shape in tensor_order():
            n = int(np.prod(shape))
            out[name] = np.frombuffer(f.read(4 * n), dtype="<f4").reshape(shape).copy()
        if f.read(1):
            raise ValueError("trailing bytes in %s" % path)
    return out


def make_random(seed: int = 1234, gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Seeded random parameter set: same dataflow and cost as any trained set; gates stay out of saturation."""
    rng = np.random.default_rng(seed)
    t = {}
    for name, shape in tensor_order():
        if name == "mean":
            a = np.full(shape, -6.0) + 0.5 * rng.standard_normal(shape)
        elif name == "scale":
            a = np.full(shape, 0.25) + 0.02 * rng.standard_normal(shape)
        elif len(shape) == 2:
            lim = gain * (3.0 / shape[0]) ** 0.5
            a = rng.uniform(-lim, lim, size=shape)
        else:
            a = rng.uniform(-0.1, 0.1, size=shape)
        t[name] = a.astype(np.float32)
    return t


def make_gate(noise_log_power: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
    """
    Hand-built "spectral gate" parameter set inside the KNS-v1 topology (no training data exists here).

    features f_k = (log P_k - mean_k) * scale_k are centred on a per-bin threshold (mean_k).  The front-end copies
    bins 0..256 into the first 257 of the 271 embedding units; every GRU layer is configured as a leaky integrator
    of its input (update gate held at a constant by its bias, reset gate open, candidate = tanh(unit-wise copy));
    stage 4's head reads unit k back out as mask_k = sigmoid(gain * h_k).  Heads 1-3 are left at zero weight
    (constant 0.5 outputs, whose feed-forward weights are zero).
    """
    t = {name: np.zeros(shape, np.float32) for name, shape in tensor_order()}
    if noise_log_power is None:
        noise_log_power = np.full(BINS, -4.0)
    t["mean"][:] = noise_log_power
    t["scale"][:] = 1.0
    eye = np.zeros((BINS, HIDDEN), np.float32)
    eye[np.arange(BINS), np.arange(BINS)] = 1.0
    t["w_in"][:] = eye
    copy = np.zeros((HIDDEN, G3), np.float32)
    copy[np.arange(HIDDEN), 2 * HIDDEN + np.arange(HIDDEN)] = 1.0  # candidate n_j <- x_j

    def layer(prefix, d_in, x_gain, z_bias):
        w = np.zeros((d_in + HIDDEN, G3), np.float32)
        w[d_in:, :] = copy * x_gain
        t[prefix + "w_ih_" + layer.tag][:] = w
        b = np.zeros(G3, np.float32)
        b[0:HIDDEN] = 8.0  # reset gate open
        b[HIDDEN:2 * HIDDEN] = z_bias  # update gate: z = sigmoid(z_bias) -> memory
        t[prefix + "b_ih_" + layer.tag][:] = b

    for s in range(STAGES):
        d_in = HEADS[s - 1] if s else 0
        first = s == 0
        layer.tag = "a"
        layer("s%d." % s, d_in, 0.35 if first else 1.2, 0.3 if first else -4.0)
        layer.tag = "b"
        layer("s%d." % s, 0, 1.2, -4.0)
    head = np.zeros((HIDDEN, BINS), np.float32)
    head[np.arange(BINS), np.arange(BINS)] = 14.0
    t["s3.w_head"][:] = head
    t["s3.b_head"][:] = 0.0
    return t


def ensure_params(path: str, kind: str = "random", seed: int = 1234, **kw) -> str:
    """Create the parameter file if it does not exist yet; returns `path`."""
    if not os.path.exists(path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        if kind == "random":
            write_params(path, make_random(seed, **kw))
        elif kind == "gate":
            write_params(path, make_gate(**kw))
        else:
            raise ValueError("unknown parameter kind `%s`" % kind)
    return path


__all__ = ["write_params", "read_params", "make_random", "make_gate", "ensure_params", "tensor_order"]
