"""Factory functions (mirrors reference binding/python/_factory.py:27-76, plus `create_batch`)."""

from typing import Optional, Sequence

from ._batch import KoalaBatch
from ._koala import Koala, list_hardware_devices
from ._util import default_library_path, default_model_path


def create(
        access_key: str,
        model_path: Optional[str] = None,
        device: Optional[str] = None,
        library_path: Optional[str] = None) -> Koala:
    """
    Single-stream engine, drop-in for `pvkoala.create`.

    :param access_key: non-empty string, not verified.
    :param model_path: KNS1 parameter file; default: the packaged `lib/koala_params.kns`.
    :param device: `best` (default), `gpu` or `gpu:${GPU_INDEX}`.
    :param library_path: libpv_koala.so; default: the in-tree HIP build.
    """
    return Koala(
        access_key=access_key,
        model_path=default_model_path() if model_path is None else model_path,
        device='best' if device is None else device,
        library_path=default_library_path() if library_path is None else library_path)


def create_batch(
        access_key: str,
        num_streams: int,
        max_frames_per_call: int = 1,
        precision: str = 'fp32',
        model_path: Optional[str] = None,
        device: Optional[str] = None,
        library_path: Optional[str] = None) -> KoalaBatch:
    """`num_streams` independent streams advancing together on one GPU (see KoalaBatch)."""
    return KoalaBatch(
        access_key=access_key,
        model_path=default_model_path() if model_path is None else model_path,
        device='best' if device is None else device,
        library_path=default_library_path() if library_path is None else library_path,
        num_streams=num_streams,
        max_frames_per_call=max_frames_per_call,
        precision=precision)


def available_devices(library_path: Optional[str] = None) -> Sequence[str]:
    """Every string that `create(device=...)` accepts on this machine ("gpu:0 - <name>", ...)."""
    return list_hardware_devices(library_path=default_library_path() if library_path is None else library_path)


__all__ = ['available_devices', 'create', 'create_batch']
