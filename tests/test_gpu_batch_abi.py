"""
The batch extension of the C ABI (include/pv_koala_batch.h) at its edges, called through ctypes exactly as a foreign
binding would: argument errors never touch the device state, every failure leaves a message on the calling thread's
error stack, and the status codes are those of picovoice.h.
"""
import ctypes as C

import numpy as np
import pytest

import koala_amd
from koala_amd._util import default_library_path
from conftest import synth_streams

pytestmark = pytest.mark.gpu

SUCCESS, OUT_OF_MEMORY, IO_ERROR, INVALID_ARGUMENT, RUNTIME_ERROR = 0, 1, 2, 3, 7


@pytest.fixture(scope='module')
def lib():
    import torch  # noqa: F401  (torch's bundled HIP runtime must be loaded first)
    l = C.CDLL(default_library_path())
    l.pv_koala_batch_init.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    l.pv_koala_batch_process_chunk.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    l.pv_koala_batch_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    l.pv_koala_batch_reset.argtypes = [C.c_void_p, C.c_void_p]
    l.pv_koala_batch_delete.argtypes = [C.c_void_p]
    l.pv_koala_batch_num_streams.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    l.pv_koala_batch_delay_sample.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    l.pv_koala_batch_host_alloc.argtypes = [C.c_int64, C.POINTER(C.c_void_p)]
    l.pv_koala_batch_host_free.argtypes = [C.c_void_p]
    l.pv_get_error_stack.argtypes = [C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.c_int32)]
    l.pv_free_error_stack.argtypes = [C.POINTER(C.c_char_p)]
    return l


def stack(lib):
    msgs, n = C.POINTER(C.c_char_p)(), C.c_int32()
    assert lib.pv_get_error_stack(C.byref(msgs), C.byref(n)) in (0, 6)
    out = [msgs[i].decode() for i in range(n.value)]
    if n.value:
        lib.pv_free_error_stack(msgs)
    return out


def open_batch(lib, model, streams=20, frames=4, precision=1, device=b'best'):
    h = C.c_void_p()
    st = lib.pv_koala_batch_init(b'key', model.encode(), device, streams, frames, precision, C.byref(h))
    return st, h


def test_init_argument_errors(lib, random_model):
    for streams, frames, precision in ((0, 4, 1), (-3, 4, 1), (20, 0, 1), (20, 4, 7)):
        st, h = open_batch(lib, random_model, streams, frames, precision)
        assert st == INVALID_ARGUMENT and not h.value and stack(lib)
    h = C.c_void_p()
    assert lib.pv_koala_batch_init(None, random_model.encode(), b'best', 4, 1, 1, C.byref(h)) == INVALID_ARGUMENT
    assert lib.pv_koala_batch_init(b'key', None, b'best', 4, 1, 1, C.byref(h)) == INVALID_ARGUMENT
    assert lib.pv_koala_batch_init(b'key', random_model.encode(), None, 4, 1, 1, C.byref(h)) == INVALID_ARGUMENT
    assert lib.pv_koala_batch_init(b'key', random_model.encode(), b'best', 4, 1, 1, None) == INVALID_ARGUMENT
    assert lib.pv_koala_batch_init(b'key', b'/nonexistent/model.kns', b'best', 4, 1, 1, C.byref(h)) == IO_ERROR
    assert any('model.kns' in m for m in stack(lib))
    assert lib.pv_koala_batch_init(b'key', random_model.encode(), b'cpu', 4, 1, 1, C.byref(h)) == RUNTIME_ERROR
    assert lib.pv_koala_batch_init(b'key', random_model.encode(), b'gpu:99', 4, 1, 1, C.byref(h)) == INVALID_ARGUMENT  # out of range
    assert lib.pv_koala_batch_init(b'key', random_model.encode(), b'tpu', 4, 1, 1, C.byref(h)) == INVALID_ARGUMENT
    lib.pv_koala_batch_delete(None)  # accepted, like pv_koala_delete(NULL)


def test_process_argument_errors_leave_the_state_alone(lib, random_model):
    B, T = 20, 4
    st, h = open_batch(lib, random_model, B, T)
    assert st == SUCCESS
    n, d = C.c_int32(), C.c_int32()
    assert lib.pv_koala_batch_num_streams(h, C.byref(n)) == SUCCESS and n.value == B
    assert lib.pv_koala_batch_delay_sample(h, C.byref(d)) == SUCCESS and d.value == 256
    assert lib.pv_koala_batch_num_streams(None, C.byref(n)) == INVALID_ARGUMENT
    assert lib.pv_koala_batch_delay_sample(h, None) == INVALID_ARGUMENT
    x = synth_streams(B, 2 * T, seed=3)
    a, b = np.ascontiguousarray(x[:, :T * 256]), np.ascontiguousarray(x[:, T * 256:])
    y = np.empty_like(a)
    assert lib.pv_koala_batch_process_chunk(h, T, a.ctypes.data, y.ctypes.data) == SUCCESS
    first = y.copy()
    for frames in (0, -1, T + 1):
        assert lib.pv_koala_batch_process_chunk(h, frames, b.ctypes.data, y.ctypes.data) == INVALID_ARGUMENT
        assert any('num_frames' in m for m in stack(lib))
    assert lib.pv_koala_batch_process_chunk(None, T, b.ctypes.data, y.ctypes.data) == INVALID_ARGUMENT
    assert lib.pv_koala_batch_process_chunk(h, T, None, y.ctypes.data) == INVALID_ARGUMENT
    assert lib.pv_koala_batch_process_chunk(h, T, b.ctypes.data, None) == INVALID_ARGUMENT
    assert lib.pv_koala_batch_process(h, None, y.ctypes.data) == INVALID_ARGUMENT
    import torch
    dev = torch.zeros(B * T * 256, dtype=torch.int16, device='cuda')
    assert lib.pv_koala_batch_process_chunk(h, T, C.c_void_p(dev.data_ptr()), y.ctypes.data) in (INVALID_ARGUMENT, RUNTIME_ERROR)
    assert stack(lib)
    # none of the failed calls advanced the streams: the second chunk continues where the first left off
    assert lib.pv_koala_batch_process_chunk(h, T, b.ctypes.data, y.ctypes.data) == SUCCESS
    ref = koala_amd.create_batch('key', B, T, 'bf16', model_path=random_model)
    assert np.array_equal(ref.process(a), first) and np.array_equal(ref.process(b), y)
    ref.delete()
    assert lib.pv_koala_batch_reset(None, None) == INVALID_ARGUMENT
    lib.pv_koala_batch_delete(h)


def test_host_alloc_edges(lib):
    p = C.c_void_p()
    assert lib.pv_koala_batch_host_alloc(0, C.byref(p)) == INVALID_ARGUMENT and not p.value
    assert lib.pv_koala_batch_host_alloc(-5, C.byref(p)) == INVALID_ARGUMENT
    assert lib.pv_koala_batch_host_alloc(4096, None) == INVALID_ARGUMENT
    assert lib.pv_koala_batch_host_alloc(1 << 20, C.byref(p)) == SUCCESS and p.value
    C.memset(p, 0x5a, 1 << 20)
    lib.pv_koala_batch_host_free(p)
    lib.pv_koala_batch_host_free(None)


def test_overlapping_host_buffers_are_refused_where_the_call_is_pipelined(lib, random_model):
    """include/pv_koala_batch.h: host-pointer calls of >= 4 MiB run as overlapping sub-chunks and need disjoint buffers; smaller
    ones are staged as a whole and may be processed in place."""
    B, T = 300, 32  # 300 x 32 x 512 B = 4.9 MB: pipelined
    st, h = open_batch(lib, random_model, B, T)
    assert st == SUCCESS
    x = synth_streams(B, T, seed=11)
    buf = np.concatenate([x, x], axis=1).copy()  # room for a shifted output inside one allocation
    flat = buf.reshape(-1)
    n = B * T * 256
    assert lib.pv_koala_batch_process_chunk(h, T, flat[:n].ctypes.data, flat[256:].ctypes.data) == RUNTIME_ERROR
    assert any('overlap' in m for m in stack(lib))
    a = np.ascontiguousarray(x)
    y = np.empty_like(a)
    assert lib.pv_koala_batch_process_chunk(h, T, a.ctypes.data, y.ctypes.data) == SUCCESS  # state untouched by the refusal
    ref = koala_amd.create_batch('key', B, T, 'bf16', model_path=random_model)
    assert np.array_equal(ref.process(a), y)
    ref.delete()
    lib.pv_koala_batch_delete(h)
    # small call: staged as a whole, exact aliasing allowed
    st, h = open_batch(lib, random_model, 20, 4)
    s = np.ascontiguousarray(synth_streams(20, 4, seed=12))
    want = koala_amd.create_batch('key', 20, 4, 'bf16', model_path=random_model)
    expect = want.process(s)
    want.delete()
    assert lib.pv_koala_batch_process_chunk(h, 4, s.ctypes.data, s.ctypes.data) == SUCCESS
    assert np.array_equal(s, expect)
    lib.pv_koala_batch_delete(h)


@pytest.mark.parametrize('precision,own_stream', [('fp32', True), ('bf16', True), ('bf16', False)])
def test_asynchronous_host_calls_equal_the_synchronous_path(random_model, precision, own_stream):
    """pv_koala_batch_process_chunk_async: two calls in flight, copies under the neighbours' kernels -- the same samples, bit for bit,
    as the same calls made synchronously; calls of different lengths, an in-place call, a reset and a synchronous call in between."""
    B, Tmax = 70, 12
    ka = koala_amd.create_batch('key', B, Tmax, precision, model_path=random_model)
    ks = koala_amd.create_batch('key', B, Tmax, precision, model_path=random_model)
    if not own_stream:  # the kernels on a caller's stream: the copy streams hang off it by events all the same
        import torch
        side = torch.cuda.Stream()
        ka.set_stream(side.cuda_stream)
    lens = [12, 5, 12, 1, 7, 12, 3]
    xs = [synth_streams(B, t, seed=900 + i) for i, t in enumerate(lens)]
    want = []
    for i, x in enumerate(xs):
        if i == 4:
            ks.reset()
        want.append(ks.process(x))
    bufs = [(ka.alloc_host(Tmax), ka.alloc_host(Tmax)) for _ in range(len(lens))]
    got = []
    for i, (x, t) in enumerate(zip(xs, lens)):
        a, b = bufs[i]
        # contiguous [B, t * 256] windows at the head of the page-locked buffers
        ain = np.frombuffer(a.reshape(-1), np.int16, B * t * 256).reshape(B, t * 256)
        aout = ain if i == 2 else np.frombuffer(b.reshape(-1), np.int16, B * t * 256).reshape(B, t * 256)
        ain[:] = x
        if i == 4:
            ka.reset()
        if i == 5:  # a synchronous call in between first waits for the two in flight
            aout[:] = ka.process(x)
        else:
            ka.process_async(ain, aout)
        got.append(aout)
        if i >= 3:  # at most two calls still in flight: call i - 3 has completed, its output is in place
            ka.wait(2)
            assert np.array_equal(got[i - 3], want[i - 3]), i - 3
    ka.synchronize()
    for i in range(len(lens)):
        assert np.array_equal(got[i], want[i]), i
    # pageable buffers are refused, nothing is processed, the message is on the stack
    x = synth_streams(B, 2, seed=1)
    with pytest.raises(koala_amd.KoalaError):
        ka.process_async(x, np.empty_like(x))
    ka.delete()
    ks.delete()
