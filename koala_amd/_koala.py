"""
ctypes binding of the MI355X-native libpv_koala.so -- same surface as the reference binding
(reference binding/python/_koala.py: exceptions :18-84, Koala :87-312, list_hardware_devices :315-339).

Differences from the reference binding, on purpose:
  * an empty `device` raises KoalaInvalidArgumentError (the reference raises NameError, _koala.py:149);
  * a failing device listing raises the mapped KoalaError (the reference hits an unqualified name, :332);
  * `process` also accepts numpy int16 arrays and converts them without a per-sample Python loop.
"""

import os
from ctypes import (POINTER, Structure, byref, c_char_p, c_int, c_int16, c_int32, c_short, c_void_p, cdll)
from typing import Sequence

from ._errors import *  # noqa: F401,F403  (re-exported: the reference keeps the exceptions in this module)
from ._errors import STATUS_TO_EXCEPTION as _STATUS_TO_EXCEPTION
from ._errors import KoalaInvalidArgumentError, KoalaIOError, PicovoiceStatuses



def load_library(library_path: str):
    """dlopen the HIP build; there is no fallback of any kind if it is missing."""
    if not os.path.exists(library_path):
        raise KoalaIOError("Could not find Koala's dynamic library at `%s`." % library_path)
    library = cdll.LoadLibrary(library_path)
    library.pv_set_sdk.argtypes = [c_char_p]
    library.pv_set_sdk.restype = None
    library.pv_get_error_stack.argtypes = [POINTER(POINTER(c_char_p)), POINTER(c_int)]
    library.pv_get_error_stack.restype = PicovoiceStatuses
    library.pv_free_error_stack.argtypes = [POINTER(c_char_p)]
    library.pv_free_error_stack.restype = None
    library.pv_koala_version.argtypes = []
    library.pv_koala_version.restype = c_char_p
    return library


def fetch_error_stack(library) -> Sequence[str]:
    ref = POINTER(c_char_p)()
    depth = c_int()
    status = library.pv_get_error_stack(byref(ref), byref(depth))
    if status is not PicovoiceStatuses.SUCCESS:
        raise _STATUS_TO_EXCEPTION[status](message='Unable to get Koala error state')
    stack = [ref[i].decode('utf-8') for i in range(depth.value)]
    library.pv_free_error_stack(ref)
    return stack


def raise_status(library, status: PicovoiceStatuses, message: str) -> None:
    raise _STATUS_TO_EXCEPTION[status](message=message, message_stack=fetch_error_stack(library))


class Koala(object):
    """One 16 kHz mono stream of the noise suppressor; `process` consumes and returns 256-sample frames."""

    PicovoiceStatuses = PicovoiceStatuses
    _PICOVOICE_STATUS_TO_EXCEPTION = _STATUS_TO_EXCEPTION

    class CKoala(Structure):
        pass

    def __init__(self, access_key: str, model_path: str, device: str, library_path: str) -> None:
        """
        :param access_key: kept for API compatibility; must be a non-empty string (no licence check is made).
        :param model_path: path of a KNS1 parameter file.
        :param device: `best`, `gpu` or `gpu:${GPU_INDEX}`.  `cpu` / `cpu:${NUM_THREADS}` parse but are refused.
        :param library_path: path of libpv_koala.so.
        """
        if not isinstance(access_key, str) or len(access_key) == 0:
            raise KoalaInvalidArgumentError("`access_key` should be a non-empty string.")
        if not os.path.exists(model_path):
            raise KoalaIOError("Could not find model file at `%s`." % model_path)
        if not isinstance(device, str) or len(device) == 0:
            raise KoalaInvalidArgumentError("`device` should be a non-empty string.")

        library = load_library(library_path)
        library.pv_set_sdk('python'.encode('utf-8'))
        self._library = library

        library.pv_koala_init.argtypes = [c_char_p, c_char_p, c_char_p, POINTER(POINTER(self.CKoala))]
        library.pv_koala_init.restype = PicovoiceStatuses
        self._handle = POINTER(self.CKoala)()
        status = library.pv_koala_init(access_key.encode(), model_path.encode(), device.encode(), byref(self._handle))
        if status is not PicovoiceStatuses.SUCCESS:
            raise_status(library, status, 'Initialization failed')

        self._delete_func = library.pv_koala_delete
        self._delete_func.argtypes = [POINTER(self.CKoala)]
        self._delete_func.restype = None

        library.pv_koala_delay_sample.argtypes = [POINTER(self.CKoala), POINTER(c_int32)]
        library.pv_koala_delay_sample.restype = PicovoiceStatuses
        delay = c_int32()
        status = library.pv_koala_delay_sample(self._handle, delay)
        if status is not PicovoiceStatuses.SUCCESS:
            self.delete()
            raise_status(library, status, 'Failed to get delay samples')
        self._delay_sample = delay.value

        self._process_func = library.pv_koala_process
        self._process_func.argtypes = [POINTER(self.CKoala), POINTER(c_short), POINTER(c_short)]
        self._process_func.restype = PicovoiceStatuses
        self._reset_func = library.pv_koala_reset
        self._reset_func.argtypes = [POINTER(self.CKoala)]
        self._reset_func.restype = PicovoiceStatuses

        self._sample_rate = library.pv_sample_rate()
        self._frame_length = library.pv_koala_frame_length()
        self._version = library.pv_koala_version().decode('utf-8')
        self._frame_type = c_short * self._frame_length

    def process(self, pcm: Sequence[int]) -> Sequence[int]:
        """
        One frame in, one (delayed) enhanced frame out.

        :param pcm: `frame_length` 16-bit samples, consecutive with the previous call unless `reset()` was called.
        :return: the enhanced samples that lie `delay_sample` behind the input, as a list of ints.
        """
        if len(pcm) != self._frame_length:
            raise KoalaInvalidArgumentError(
                "Length of input frame %d does not match required frame length %d" % (len(pcm), self._frame_length))
        if hasattr(pcm, 'ctypes') and getattr(pcm, 'dtype', None) == 'int16' and pcm.flags['C_CONTIGUOUS']:
            frame = pcm.ctypes.data_as(POINTER(c_short))
        else:
            frame = self._frame_type(*pcm)
        enhanced = self._frame_type()
        status = self._process_func(self._handle, frame, enhanced)
        if status is not PicovoiceStatuses.SUCCESS:
            raise_status(self._library, status, 'Processing failed')
        return list(enhanced)

    def reset(self) -> None:
        """Back to the state of a new instance; call between non-consecutive pieces of audio."""
        status = self._reset_func(self._handle)
        if status is not PicovoiceStatuses.SUCCESS:
            raise_status(self._library, status, 'Reset failed')

    def delete(self) -> None:
        """Releases the native stream."""
        self._delete_func(self._handle)
        self._handle = None

    @property
    def sample_rate(self) -> int:
        return self._sample_rate

    @property
    def frame_length(self) -> int:
        return self._frame_length

    @property
    def delay_sample(self) -> int:
        """Shift, in samples, between the input stream and the output stream."""
        return self._delay_sample

    @property
    def version(self) -> str:
        return self._version

    def _get_error_stack(self) -> Sequence[str]:
        return fetch_error_stack(self._library)


def list_hardware_devices(library_path: str) -> Sequence[str]:
    library = load_library(library_path)
    fn = library.pv_koala_list_hardware_devices
    fn.argtypes = [POINTER(POINTER(c_char_p)), POINTER(c_int32)]
    fn.restype = PicovoiceStatuses
    devices = POINTER(c_char_p)()
    count = c_int32()
    status = fn(byref(devices), byref(count))
    if status is not PicovoiceStatuses.SUCCESS:
        raise _STATUS_TO_EXCEPTION[status](message='`pv_koala_list_hardware_devices` failed.')
    result = [devices[i].decode() for i in range(count.value)]
    free = library.pv_koala_free_hardware_devices
    free.argtypes = [POINTER(c_char_p), c_int32]
    free.restype = None
    free(devices, count.value)
    return result


from . import _errors  # noqa: E402

__all__ = ['Koala', 'list_hardware_devices', 'load_library', 'fetch_error_stack', 'raise_status'] + [
    n for n in _errors.__all__ if n != 'STATUS_TO_EXCEPTION']
