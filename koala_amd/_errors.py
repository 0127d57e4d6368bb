"""
Status codes of the C ABI and the exception hierarchy they map to.  Names and meaning follow the reference binding
(reference binding/python/_koala.py:18-117: `KoalaError` with `message` / `message_stack`, one subclass per
pv_status_t failure code) so that `except pvkoala.KoalaIOError` keeps working after a switch of package.
"""

from enum import Enum
from typing import Sequence


class PicovoiceStatuses(Enum):
    """pv_status_t (include/picovoice.h)"""
    SUCCESS = 0
    OUT_OF_MEMORY = 1
    IO_ERROR = 2
    INVALID_ARGUMENT = 3
    STOP_ITERATION = 4
    KEY_ERROR = 5
    INVALID_STATE = 6
    RUNTIME_ERROR = 7
    ACTIVATION_ERROR = 8
    ACTIVATION_LIMIT_REACHED = 9
    ACTIVATION_THROTTLED = 10
    ACTIVATION_REFUSED = 11


class KoalaError(Exception):
    """Base class; `message_stack` carries the library's thread-local error stack of the failing call."""

    def __init__(self, message: str = '', message_stack: Sequence[str] = None):
        super().__init__(message)
        self._message = message
        self._message_stack = [] if message_stack is None else message_stack

    def __str__(self):
        head = self._message + (':' if self._message_stack else '')
        return '\n'.join([head] + ['  [%d] %s' % (i, m) for i, m in enumerate(self._message_stack)])

    @property
    def message(self) -> str:
        return self._message

    @property
    def message_stack(self) -> Sequence[str]:
        return self._message_stack


def _subclass(name: str, doc: str):
    return type(name, (KoalaError,), {'__doc__': doc, '__module__': __name__})


KoalaMemoryError = _subclass('KoalaMemoryError', 'PV_STATUS_OUT_OF_MEMORY')
KoalaIOError = _subclass('KoalaIOError', 'PV_STATUS_IO_ERROR')
KoalaInvalidArgumentError = _subclass('KoalaInvalidArgumentError', 'PV_STATUS_INVALID_ARGUMENT')
KoalaStopIterationError = _subclass('KoalaStopIterationError', 'PV_STATUS_STOP_ITERATION')
KoalaKeyError = _subclass('KoalaKeyError', 'PV_STATUS_KEY_ERROR')
KoalaInvalidStateError = _subclass('KoalaInvalidStateError', 'PV_STATUS_INVALID_STATE')
KoalaRuntimeError = _subclass('KoalaRuntimeError', 'PV_STATUS_RUNTIME_ERROR')
KoalaActivationError = _subclass('KoalaActivationError', 'PV_STATUS_ACTIVATION_ERROR')
KoalaActivationLimitError = _subclass('KoalaActivationLimitError', 'PV_STATUS_ACTIVATION_LIMIT_REACHED')
KoalaActivationThrottledError = _subclass('KoalaActivationThrottledError', 'PV_STATUS_ACTIVATION_THROTTLED')
KoalaActivationRefusedError = _subclass('KoalaActivationRefusedError', 'PV_STATUS_ACTIVATION_REFUSED')

STATUS_TO_EXCEPTION = {
    PicovoiceStatuses.OUT_OF_MEMORY: KoalaMemoryError,
    PicovoiceStatuses.IO_ERROR: KoalaIOError,
    PicovoiceStatuses.INVALID_ARGUMENT: KoalaInvalidArgumentError,
    PicovoiceStatuses.STOP_ITERATION: KoalaStopIterationError,
    PicovoiceStatuses.KEY_ERROR: KoalaKeyError,
    PicovoiceStatuses.INVALID_STATE: KoalaInvalidStateError,
    PicovoiceStatuses.RUNTIME_ERROR: KoalaRuntimeError,
    PicovoiceStatuses.ACTIVATION_ERROR: KoalaActivationError,
    PicovoiceStatuses.ACTIVATION_LIMIT_REACHED: KoalaActivationLimitError,
    PicovoiceStatuses.ACTIVATION_THROTTLED: KoalaActivationThrottledError,
    PicovoiceStatuses.ACTIVATION_REFUSED: KoalaActivationRefusedError,
}

__all__ = ['PicovoiceStatuses', 'KoalaError', 'STATUS_TO_EXCEPTION'] + [c.__name__ for c in STATUS_TO_EXCEPTION.values()]
