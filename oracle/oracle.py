"""
ctypes wrapper of the CPU oracle (oracle/libkns_oracle.so).  TEST INFRASTRUCTURE ONLY -- imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the koala_amd package.
"""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkns_oracle.so")

PREC_FP32 = 0
PREC_BF16 = 1
FRAME = 256
BINS = 257
HIDDEN = 271


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, n) for n in ("kns_oracle.c", "kns_oracle.h", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libkns_oracle.so"])
    return _LIB_PATH


class _Taps(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_float)) for n in ("spectrum", "features", "embed", "heads", "hidden")]


def build_native() -> str:
    """The same source compiled -march=native on THIS host (bench.py's cpu_baseline leg); falls back to the portable build.
    Always rebuilt (-B): a copy made on another machine may use instructions this one lacks (it is .gpurunignore'd as well)."""
    path = os.path.join(_HERE, "libkns_oracle_native.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libkns_oracle_native.so"])
        return path
    except Exception:
        return build()


_lib = None
_libs = {}


def lib(path=None):
    global _lib
    if path is not None:
        if path not in _libs:
            _libs[path] = _bind(C.CDLL(path))
        return _libs[path]
    if _lib is None:
        _lib = _bind(C.CDLL(build()))
    return _lib


def _bind(l):
    l.kns_params_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    l.kns_params_free.argtypes = [C.c_void_p]
    l.kns_oracle_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    l.kns_oracle_delete.argtypes = [C.c_void_p]
    l.kns_oracle_reset.argtypes = [C.c_void_p, C.c_void_p]
    l.kns_oracle_process.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    l.kns_oracle_process_mask.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    l.kns_oracle_process_tap.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(_Taps)]
    l.kns_oracle_set_jitter.argtypes = [C.c_int]
    l.kns_oracle_analysis.argtypes = [C.c_void_p] * 5
    l.kns_oracle_synthesis.argtypes = [C.c_void_p] * 4
    for n in ("kns_exp", "kns_log", "kns_sigmoid", "kns_tanh", "kns_round_bf16", "kns_round_fp16"):
        getattr(l, n).argtypes = [C.c_float]
        getattr(l, n).restype = C.c_float
    return l


def set_jitter(seed: int) -> None:
    """Sensitivity probe (kns_oracle.h): bf16 oracles of this process play a second valid implementation; 0 = off."""
    lib().kns_oracle_set_jitter(int(seed))


def block_size() -> int:
    """Streams per weight pass in the last `Oracle.process` call (reporting only)."""
    return int(lib().kns_oracle_last_block())


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """`num_streams` independent KNS-v1 streams on the host CPU."""

    def __init__(self, model_path: str, num_streams: int = 1, precision: int = PREC_FP32, library: str = None):
        self._l = lib(library)
        self._params = C.c_void_p()
        rc = self._l.kns_params_load(model_path.encode(), precision, C.byref(self._params))
        if rc != 0:
            raise IOError("kns_params_load(%s) failed: %d" % (model_path, rc))
        self._o = C.c_void_p()
        if self._l.kns_oracle_create(self._params, num_streams, C.byref(self._o)) != 0:
            raise RuntimeError("kns_oracle_create failed")
        self.num_streams = num_streams
        self.frame_length = FRAME
        self.delay_sample = self._l.kns_oracle_delay_sample()

    def process(self, pcm: np.ndarray, num_threads: int = 0) -> np.ndarray:
        """pcm int16 [num_streams, T*256] (or [T*256] when num_streams == 1) -> enhanced, same shape."""
        a = np.ascontiguousarray(pcm, dtype=np.int16)
        flat = a.reshape(self.num_streams, -1)
        if flat.shape[1] % FRAME:
            raise ValueError("length must be a multiple of 256")
        out = np.empty_like(flat)
        rc = self._l.kns_oracle_process(self._o, flat.shape[1] // FRAME, _ptr(flat), _ptr(out), num_threads)
        if rc != 0:
            raise RuntimeError("kns_oracle_process failed")
        return out.reshape(a.shape)

    def process_with_mask(self, pcm: np.ndarray, num_threads: int = 0):
        """pcm int16 [num_streams, T*256] -> (enhanced, mask float32 [T, num_streams, 257])"""
        a = np.ascontiguousarray(pcm, dtype=np.int16).reshape(self.num_streams, -1)
        T = a.shape[1] // FRAME
        out = np.empty_like(a)
        mask = np.empty((T, self.num_streams, BINS), np.float32)
        if self._l.kns_oracle_process_mask(self._o, T, _ptr(a), _ptr(out), _ptr(mask), num_threads) != 0:
            raise RuntimeError("kns_oracle_process_mask failed")
        return out, mask

    def process_tap(self, frame: np.ndarray, stream: int = 0):
        """one frame of one stream; returns (enhanced[256], dict of intermediates)"""
        a = np.ascontiguousarray(frame, dtype=np.int16)
        out = np.empty(FRAME, np.int16)
        bufs = {
            "spectrum": np.empty((BINS, 2), np.float32), "features": np.empty(BINS, np.float32),
            "embed": np.empty(HIDDEN, np.float32), "heads": np.empty(1 + 5 + 40 + 257, np.float32),
            "hidden": np.empty((8, HIDDEN), np.float32),
        }
        taps = _Taps(*[bufs[n].ctypes.data_as(C.POINTER(C.c_float)) for n, _ in _Taps._fields_])
        rc = self._l.kns_oracle_process_tap(self._o, stream, _ptr(a), _ptr(out), C.byref(taps))
        if rc != 0:
            raise RuntimeError("kns_oracle_process_tap failed")
        bufs["mask"] = bufs["heads"][46:]
        return out, bufs

    def reset(self, stream_mask=None) -> None:
        m = None
        if stream_mask is not None:
            m = np.ascontiguousarray(stream_mask, dtype=np.uint8)
        self._l.kns_oracle_reset(self._o, _ptr(m) if m is not None else None)

    def analysis(self, hist: np.ndarray, pcm: np.ndarray):
        spec = np.empty((BINS, 2), np.float32)
        feat = np.empty(BINS, np.float32)
        h = np.ascontiguousarray(hist, np.int16)
        p = np.ascontiguousarray(pcm, np.int16)
        self._l.kns_oracle_analysis(self._params, _ptr(h), _ptr(p), _ptr(spec), _ptr(feat))
        return spec, feat

    def delete(self) -> None:
        if self._o:
            self._l.kns_oracle_delete(self._o)
            self._l.kns_params_free(self._params)
            self._o = None

    def __del__(self):
        try:
            self.delete()
        except Exception:
            pass


def synthesis(spectrum: np.ndarray, mask: np.ndarray, tail: np.ndarray):
    """returns (out int16[256]); `tail` float32[256] is updated in place"""
    s = np.ascontiguousarray(spectrum, np.float32)
    m = np.ascontiguousarray(mask, np.float32)
    out = np.empty(FRAME, np.int16)
    assert tail.dtype == np.float32 and tail.flags.c_contiguous
    lib().kns_oracle_synthesis(_ptr(s), _ptr(m), _ptr(tail), _ptr(out))
    return out


def scalar(fn: str, x: np.ndarray) -> np.ndarray:
    f = getattr(lib(), "kns_" + fn)
    return np.array([f(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32).reshape(np.shape(x))
