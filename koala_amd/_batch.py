"""
Python face of the batch extension (include/pv_koala_batch.h): B lock-stepped streams on one MI355X.
The reference has no counterpart (one stream per handle, include/pv_koala.h:65-80); stream b of a batch is
sample-for-sample what its own `Koala` instance would produce.
"""

import os
from ctypes import POINTER, byref, c_char_p, c_double, c_int16, c_int32, c_int64, c_void_p
from typing import Optional

import numpy as np

from ._koala import (KoalaError, KoalaInvalidArgumentError, KoalaIOError, PicovoiceStatuses, load_library,
                     raise_status)

PRECISION_FP32 = 0
PRECISION_BF16 = 1
KERNEL_CLASSES = ('analysis', 'gemm_input', 'gru_recurrent', 'gemm_head', 'synthesis')


class KoalaBatch(object):
    def __init__(self, access_key: str, model_path: str, device: str, library_path: str, num_streams: int,
                 max_frames_per_call: int = 1, precision: str = 'fp32') -> None:
        if not isinstance(access_key, str) or len(access_key) == 0:
            raise KoalaInvalidArgumentError("`access_key` should be a non-empty string.")
        if not os.path.exists(model_path):
            raise KoalaIOError("Could not find model file at `%s`." % model_path)
        if precision not in ('fp32', 'bf16'):
            raise KoalaInvalidArgumentError("`precision` should be `fp32` or `bf16`.")
        lib = load_library(library_path)
        lib.pv_set_sdk(b'python')
        self._lib = lib
        lib.pv_koala_batch_init.argtypes = [c_char_p, c_char_p, c_char_p, c_int32, c_int32, c_int32, POINTER(c_void_p)]
        lib.pv_koala_batch_init.restype = PicovoiceStatuses
        for name, args in (('process_chunk', [c_void_p, c_int32, c_void_p, c_void_p]), ('process_chunk_async', [c_void_p, c_int32, c_void_p, c_void_p]),
                           ('async_wait', [c_void_p, c_int32]), ('reset', [c_void_p, c_void_p]),
                           ('set_stream', [c_void_p, c_void_p]), ('synchronize', [c_void_p]),
                           ('profile_enable', [c_void_p, c_int32]),
                           ('profile_read', [c_void_p, POINTER(c_double), POINTER(c_int64)]),
                           ('delay_sample', [c_void_p, POINTER(c_int32)])):
            fn = getattr(lib, 'pv_koala_batch_' + name)
            fn.argtypes = args
            fn.restype = PicovoiceStatuses
        lib.pv_koala_batch_delete.argtypes = [c_void_p]
        lib.pv_koala_batch_delete.restype = None
        lib.pv_koala_batch_host_alloc.argtypes = [c_int64, POINTER(c_void_p)]
        lib.pv_koala_batch_host_alloc.restype = PicovoiceStatuses
        lib.pv_koala_batch_host_free.argtypes = [c_void_p]
        lib.pv_koala_batch_host_free.restype = None
        self._pinned = []
        lib.pv_koala_batch_debug_read.argtypes = [c_void_p, c_int32, c_void_p, c_int64]
        lib.pv_koala_batch_debug_read.restype = c_int64

        self._handle = c_void_p()
        status = lib.pv_koala_batch_init(access_key.encode(), model_path.encode(), device.encode(), num_streams,
                                         max_frames_per_call, PRECISION_BF16 if precision == 'bf16' else PRECISION_FP32,
                                         byref(self._handle))
        if status is not PicovoiceStatuses.SUCCESS:
            raise_status(lib, status, 'Initialization failed')
        self.num_streams = num_streams
        self.max_frames_per_call = max_frames_per_call
        self.precision = precision
        self.frame_length = lib.pv_koala_frame_length()
        self.sample_rate = lib.pv_sample_rate()
        d = c_int32()
        self._check(lib.pv_koala_batch_delay_sample(self._handle, byref(d)), 'Failed to get delay samples')
        self.delay_sample = d.value

    def _check(self, status, what):
        if status is not PicovoiceStatuses.SUCCESS:
            raise_status(self._lib, status, what)

    def process(self, pcm: np.ndarray) -> np.ndarray:
        """pcm: int16 [num_streams, T*256] in host memory -> enhanced, same shape (synchronous)."""
        a = np.ascontiguousarray(pcm, dtype=np.int16)
        if a.ndim != 2 or a.shape[0] != self.num_streams or a.shape[1] % self.frame_length:
            raise KoalaInvalidArgumentError("expected int16 array of shape [%d, T*%d]" % (self.num_streams, self.frame_length))
        out = np.empty_like(a)
        self._check(self._lib.pv_koala_batch_process_chunk(self._handle, a.shape[1] // self.frame_length,
                                                           a.ctypes.data, out.ctypes.data), 'Processing failed')
        return out

    def alloc_host(self, num_frames: int) -> np.ndarray:
        """int16 [num_streams, num_frames*256] in page-locked host memory (freed by `delete()`): `process()` on such arrays
        lets the GPU's copy engines move the audio directly instead of through a staging copy."""
        n = self.num_streams * num_frames * self.frame_length
        p = c_void_p()
        self._check(self._lib.pv_koala_batch_host_alloc(2 * n, byref(p)), 'Host allocation failed')
        self._pinned.append(p)
        buf = (c_int16 * n).from_address(p.value)
        return np.frombuffer(buf, dtype=np.int16).reshape(self.num_streams, num_frames * self.frame_length)

    def process_into(self, pcm: np.ndarray, enhanced: np.ndarray) -> None:
        """Like `process()`, writing into a caller-provided array (both C-contiguous int16 of the same shape)."""
        for a in (pcm, enhanced):
            if (not isinstance(a, np.ndarray) or a.dtype != np.int16 or not a.flags['C_CONTIGUOUS'] or a.ndim != 2 or
                    a.shape[0] != self.num_streams or a.shape[1] % self.frame_length or a.shape != pcm.shape):
                raise KoalaInvalidArgumentError(
                    "expected C-contiguous int16 arrays of shape [%d, T*%d]" % (self.num_streams, self.frame_length))
        self._check(self._lib.pv_koala_batch_process_chunk(self._handle, pcm.shape[1] // self.frame_length,
                                                           pcm.ctypes.data, enhanced.ctypes.data), 'Processing failed')

    def process_async(self, pcm: np.ndarray, enhanced: np.ndarray) -> None:
        """`process_into()` without the wait, for page-locked arrays (`alloc_host()`): the call is enqueued and returns; up to three are
        in flight, so a caller that rotates over three buffer pairs keeps the link and the GPU busy at once.  `enhanced` is valid
        after `synchronize()` or once `wait(k)` says the call is no longer among the k in flight."""
        for a in (pcm, enhanced):
            if (not isinstance(a, np.ndarray) or a.dtype != np.int16 or not a.flags['C_CONTIGUOUS'] or a.ndim != 2 or
                    a.shape[0] != self.num_streams or a.shape[1] % self.frame_length or a.shape != pcm.shape):
                raise KoalaInvalidArgumentError(
                    "expected C-contiguous int16 arrays of shape [%d, T*%d]" % (self.num_streams, self.frame_length))
        self._check(self._lib.pv_koala_batch_process_chunk_async(self._handle, pcm.shape[1] // self.frame_length,
                                                                 pcm.ctypes.data, enhanced.ctypes.data), 'Processing failed')

    def wait(self, max_in_flight: int = 0) -> None:
        """Blocks until at most `max_in_flight` asynchronous calls are still in flight (0: all done).  Triple buffering: before reusing
        buffer pair n % 3 for call n, `wait(2)` -- call n - 3 has completed, its output may be taken and its input refilled."""
        self._check(self._lib.pv_koala_batch_async_wait(self._handle, max_in_flight), 'wait failed')

    def process_device(self, num_frames: int, pcm_ptr: int, enhanced_ptr: int) -> None:
        """Device pointers (e.g. torch_tensor.data_ptr()) of int16 [num_streams, num_frames*256]; asynchronous."""
        self._check(self._lib.pv_koala_batch_process_chunk(self._handle, num_frames, c_void_p(pcm_ptr),
                                                           c_void_p(enhanced_ptr)), 'Processing failed')

    def reset(self, stream_mask: Optional[np.ndarray] = None) -> None:
        ptr = None
        if stream_mask is not None:
            m = np.ascontiguousarray(stream_mask, dtype=np.uint8)
            if m.shape != (self.num_streams,):
                raise KoalaInvalidArgumentError("`stream_mask` must have one entry per stream")
            ptr = m.ctypes.data
        self._check(self._lib.pv_koala_batch_reset(self._handle, ptr), 'Reset failed')

    def set_stream(self, hip_stream: int) -> None:
        self._check(self._lib.pv_koala_batch_set_stream(self._handle, c_void_p(hip_stream)), 'set_stream failed')

    def synchronize(self) -> None:
        self._check(self._lib.pv_koala_batch_synchronize(self._handle), 'synchronize failed')

    def profile_enable(self, enable: bool = True) -> None:
        self._check(self._lib.pv_koala_batch_profile_enable(self._handle, 1 if enable else 0), 'profile failed')

    def profile_read(self):
        ms = (c_double * 5)()
        n = (c_int64 * 5)()
        self._check(self._lib.pv_koala_batch_profile_read(self._handle, ms, n), 'profile failed')
        return {k: {'ms': ms[i], 'launches': n[i]} for i, k in enumerate(KERNEL_CLASSES)}

    def debug_read(self, what: str, num_frames: int) -> np.ndarray:
        shapes = {'features': (0, (num_frames, self.num_streams, 257)), 'spectrum': (1, (num_frames, self.num_streams, 257, 2)),
                  'mask': (2, (num_frames, self.num_streams, 257)), 'hidden': (3, (8, self.num_streams, 271)),
                  'embed': (4, (num_frames, self.num_streams, 271)),
                  'route': (6, (4,))}  # developer library only: [route, features not stored, mask head fused, spectrum stored]
        code, shape = shapes[what]
        out = np.empty(shape, np.float32)
        n = self._lib.pv_koala_batch_debug_read(self._handle, code, out.ctypes.data, out.size)
        if n != out.size:
            raise KoalaError("debug_read(%s) returned %d, expected %d" % (what, n, out.size))
        return out

    def delete(self) -> None:
        if self._handle:
            self._lib.pv_koala_batch_delete(self._handle)
            self._handle = None
        for p in getattr(self, '_pinned', []):  # arrays from alloc_host() must not be used after this
            self._lib.pv_koala_batch_host_free(p)
        self._pinned = []

    def __del__(self):
        try:
            self.delete()
        except Exception:
            pass


__all__ = ['KoalaBatch', 'PRECISION_FP32', 'PRECISION_BF16', 'KERNEL_CLASSES']
