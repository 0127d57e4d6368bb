"""
Parity of the HIP engine (through the C ABI of libpv_koala.so) with the CPU oracle on a real MI355X.

Bars (BASELINE.json north_star / DESIGN.md section 5):
  fp32 engine : IDENTICAL to the fp32 oracle -- int16 PCM and the spectrum / feature / embedding / mask / hidden-state taps, value
                for value (since round 4 the spec's FFT is the transform the kernels evaluate, operation for operation)
  bf16 engine : mask within 1e-3 RMS of the fp32 oracle; PCM within 5 LSB of the oracle run with the same
                rounding points (bf16 GEMM operands, fp16 pre-activations); share of samples within 1 LSB: >= 99 % on the real-speech
                histogram of this file (102 400 samples: 99.1 % measured), >= 99.9 % on the bench's timed batch (enforced by bench.py)
                and on the default-model soak below
Size-independent properties are checked at BASELINE's full batch (4096 streams).
"""
import numpy as np
import pytest

import koala_amd
from conftest import model_file, synth_streams
from oracle import oracle

pytestmark = pytest.mark.gpu

# bf16 engine vs the oracle with the same rounding points: the two differ in the last bit of GEMM outputs (the bf16 MFMA truncates
# far-apart addends inside its sums of eight, profiles/r05_mfma_probe.txt) and of the gate / head / log transcendentals (hardware
# v_exp / v_rcp / v_log, <= 1 ulp from the correctly rounded functions the oracle uses), and such a bit now and then flips an fp16 /
# bf16 rounding downstream.  Largest difference ever measured in this suite: see DESIGN.md section 5.
# Bar 5 LSB (the same as __graft_entry__.smoke(), whose white-noise sample measures 4; this suite's own maximum is 3) plus the
# distribution checked wherever a histogram is taken.  Constants, not knobs: nothing in the environment can loosen them.
BF16_TOL = 5
# share of samples within 1 LSB: 99.99 % over the bench's 16.8 M samples of synthetic streams (bench.py fails its run below
# 99.9 %); on real speech through the random-weight model (test_bf16_against_both_oracles, 102 400 samples) 99.1 % measured
BF16_WITHIN_1 = 0.99
# fp32 engine: the oracle's values, every one of them
FP32_TOL = 0
# the -DKNS_DEV build of the same sources: the only library that reads the KOALA_AMD_* developer switches
DEV_LIB = koala_amd.developer_library_path()


def run_oracle(model, x, precision=oracle.PREC_FP32):
    return oracle.Oracle(model, x.shape[0], precision).process(x)


def lsb(a, b):
    return np.abs(a.astype(np.int64) - b.astype(np.int64))


@pytest.mark.parametrize('B,T,calls', [(1, 1, 4), (19, 3, 2), (40, 8, 1), (16, 1, 3)])
def test_fp32_stage_taps_and_pcm(random_model, monkeypatch, B, T, calls):
    # (multi-frame calls do not store the spectrum at all -- the synthesis kernel rebuilds it -- unless the taps are on)
    monkeypatch.setenv('KOALA_AMD_DEBUG_TAPS', '1')
    x = synth_streams(B, T * calls, seed=100 + B)
    kb = koala_amd.create_batch('key', B, T, 'fp32', model_path=random_model, library_path=DEV_LIB)
    streams = [oracle.Oracle(random_model) for _ in range(B)]
    for c in range(calls):
        xc = np.ascontiguousarray(x[:, c * T * 256:(c + 1) * T * 256])
        y = kb.process(xc)
        taps = {k: kb.debug_read(k, T) for k in ('spectrum', 'features', 'embed', 'mask')}
        hidden = kb.debug_read('hidden', T)
        for b in range(B):
            for t in range(T):
                ref, tp = streams[b].process_tap(xc[b, t * 256:(t + 1) * 256])
                for k in ('spectrum', 'features', 'embed', 'mask'):
                    assert np.array_equal(taps[k][t, b], tp[k]), (k, c, b, t, float(np.abs(taps[k][t, b] - tp[k]).max()))
                assert np.array_equal(ref, y[b, t * 256:(t + 1) * 256])
            assert np.array_equal(hidden[:, b], tp['hidden'])
    kb.delete()


@pytest.mark.parametrize('name', ['test', 'noise', 'mixed'])
def test_fp32_single_stream_abi_on_reference_wavs(gate_model, random_model, test_pcm, noise_pcm, name):
    """pv_koala_init/process (the reference ABI, one frame per call) on resources/audio_samples: the oracle's samples, all of them."""
    pcm = {'test': test_pcm, 'noise': noise_pcm,
           'mixed': (test_pcm.astype(int) + noise_pcm).astype(np.int16)}[name]
    n = len(pcm) // 256 * 256
    for model in (gate_model, random_model):
        k = koala_amd.create('key', model_path=model, device='gpu:0')
        out = np.concatenate([np.array(k.process(pcm[i:i + 256]), np.int16) for i in range(0, n, 256)])
        k.delete()
        ref = run_oracle(model, pcm[None, :n])[0]
        assert np.array_equal(out, ref), int(lsb(out, ref).max())


def test_bf16_against_both_oracles(random_model, test_pcm):
    n = 256 * 200
    x = np.stack([test_pcm[:n], test_pcm[4000:4000 + n]])
    kb = koala_amd.create_batch('key', 2, 50, 'bf16', model_path=random_model)
    o32 = [oracle.Oracle(random_model) for _ in range(2)]
    out, mask_err = [], []
    for c in range(4):
        xc = np.ascontiguousarray(x[:, c * 50 * 256:(c + 1) * 50 * 256])
        out.append(kb.process(xc))
        m = kb.debug_read('mask', 50)
        for b in range(2):
            for t in range(50):
                _, tp = o32[b].process_tap(xc[b, t * 256:(t + 1) * 256])
                mask_err.append(m[t, b] - tp['mask'])
    kb.delete()
    out = np.concatenate(out, axis=1)
    rms = float(np.sqrt(np.mean(np.square(np.stack(mask_err)))))
    assert rms < 1e-3, rms  # north_star: floating-point mask path within 1e-3 RMS
    d = lsb(out, run_oracle(random_model, x, oracle.PREC_BF16))
    hist = np.bincount(np.minimum(d.ravel(), 8), minlength=9)
    print('bf16 engine vs bf16-rounding oracle, |diff| histogram 0..8+:', hist.tolist(), 'mask rms vs fp32:', rms)
    assert d.max() <= BF16_TOL and (d == 0).mean() > 0.75 and (d <= 1).mean() > BF16_WITHIN_1
    assert lsb(out, run_oracle(random_model, x)).max() <= 24  # against the unrounded oracle


@pytest.mark.parametrize('B,T', [(20, 4), (64, 7), (4096, 2)])
def test_bf16_results_do_not_vary_from_run_to_run(random_model, B, T):
    """Two fresh handles, the same input: the same bits.  (The gate arithmetic of the bf16 recurrent kernels reads MFMA and
    transcendental results from inline-asm v_fma_mix_f32 instructions, which hipcc's hazard recognizer does not see; without the
    explicit wait states of kns_device.hpp the gate values varied from run to run and were off by ~100 LSB.)"""
    x = synth_streams(B, T, seed=3)
    outs = []
    for _ in range(2):
        kb = koala_amd.create_batch('key', B, T, 'bf16', model_path=random_model)
        outs.append(kb.process(x))
        kb.delete()
    assert np.array_equal(outs[0], outs[1])
    assert lsb(outs[0], run_oracle(random_model, x, oracle.PREC_BF16)).max() <= BF16_TOL


@pytest.mark.parametrize('B,T,calls', [(256, 16, 2), (1024, 4, 2)])  # (host calls below 4 MiB: not cut into sub-chunks)
def test_bf16_mask_rms_at_batch_scale(random_model, B, T, calls):
    """north_star's criterion for the bf16 configuration -- mask within 1e-3 RMS of the floating-point (fp32) path -- over
    every stream, frame and bin of a batch of distinct streams (bench.py reports the same over its 1 024 distinct streams x 64
    frames at 4096 streams: `parity.mask_rms_vs_fp32_oracle`)."""
    x = synth_streams(B, T * calls, seed=77)
    kb = koala_amd.create_batch('key', B, T, 'bf16', model_path=random_model)
    ref = oracle.Oracle(random_model, B, oracle.PREC_FP32)
    sq, n, worst = 0.0, 0, 0.0
    for c in range(calls):
        xc = np.ascontiguousarray(x[:, c * T * 256:(c + 1) * T * 256])
        kb.process(xc)
        got = kb.debug_read('mask', T)
        _, want = ref.process_with_mask(xc)
        d = got.astype(np.float64) - want
        sq += float(np.sum(d * d))
        n += d.size
        worst = max(worst, float(np.abs(d).max()))
    kb.delete()
    rms = np.sqrt(sq / n)
    print('bf16 mask vs fp32 oracle over %d values: rms %.3e, max %.3e' % (n, rms, worst))
    assert rms < 1e-3, rms


@pytest.mark.parametrize('model_kind', ['random', 'random5'])
def test_full_batch_one_frame_calls_equal_one_long_call(random_model, random5_model, model_kind):
    """BASELINE configs[2]'s batch, 4 096 distinct streams: eight one-frame calls (the one-step quad kernel with the narrow heads
    inside, the front-end in the analysis launch and the mask head in the synthesis launch; KNS-v1.1: the feature history rolled
    by the analysis kernel, the one-frame form of the five-frame front-end) against ONE eight-frame call (input GEMM + recurrent
    kernel, weight-stationary narrow GEMMs, recomputed spectrum): the same PCM, bit for bit -- a stream's samples do not depend on
    how its frames were grouped into calls.  Size-independent property: no oracle involved."""
    model = random_model if model_kind == 'random' else random5_model
    B, T = 4096, 8
    x = synth_streams(B, 2 * T, seed=77)
    ka = koala_amd.create_batch('key', B, T, 'bf16', model_path=model)
    kb = koala_amd.create_batch('key', B, T, 'bf16', model_path=model)
    for c in range(2):  # the second round starts from non-trivial state on both sides
        xc = np.ascontiguousarray(x[:, c * T * 256:(c + 1) * T * 256])
        whole = ka.process(xc)
        parts = np.concatenate([kb.process(np.ascontiguousarray(xc[:, t * 256:(t + 1) * 256])) for t in range(T)], axis=1)
        assert np.array_equal(whole, parts), (c, int(lsb(whole, parts).max()))
    ka.delete()
    kb.delete()


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_result_does_not_depend_on_batch_slot_or_chunking(random_model, precision):
    x = synth_streams(3, 12, seed=5)
    solo = []
    for s in range(3):
        kb = koala_amd.create_batch('key', 1, 1, precision, model_path=random_model)
        solo.append(np.concatenate([kb.process(np.ascontiguousarray(x[s:s + 1, i * 256:(i + 1) * 256]))
                                    for i in range(12)], axis=1)[0])
        kb.delete()
    big = np.tile(x, (17, 1))[:49]  # 49 streams: the three inputs repeated over many slots and m-tiles
    kb = koala_amd.create_batch('key', 49, 12, precision, model_path=random_model)
    y = kb.process(big)
    kb.delete()
    for i in range(49):
        assert np.array_equal(y[i], solo[i % 3]), i
    kb = koala_amd.create_batch('key', 49, 5, precision, model_path=random_model)
    parts = [kb.process(np.ascontiguousarray(big[:, a * 256:b * 256])) for a, b in ((0, 5), (5, 7), (7, 12))]
    kb.delete()
    assert np.array_equal(np.concatenate(parts, axis=1), y)


def test_reset_full_and_masked(random_model):
    x = synth_streams(20, 6, seed=8)
    kb = koala_amd.create_batch('key', 20, 6, 'fp32', model_path=random_model)
    a = kb.process(x)
    kb.reset()
    assert np.array_equal(kb.process(x), a)  # reference test_reset: bit-identical second pass
    m = np.zeros(20, np.uint8)
    m[[0, 7, 19]] = 1
    kb.reset(m)
    b = kb.process(x)
    kb.delete()
    for s in range(20):
        assert np.array_equal(a[s], b[s]) == bool(m[s])


def test_device_pointers_on_caller_stream(random_model):
    torch = pytest.importorskip('torch')
    x = synth_streams(33, 4, seed=2)
    kb = koala_amd.create_batch('key', 33, 4, 'fp32', model_path=random_model)
    host = kb.process(x)
    kb.reset()
    dx = torch.from_numpy(x).cuda()
    dy = torch.zeros_like(dx)
    kb.set_stream(torch.cuda.current_stream().cuda_stream)
    kb.process_device(4, dx.data_ptr(), dy.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(dy.cpu().numpy(), host)
    kb.set_stream(0)
    kb.delete()


def test_device_call_in_place(random_model):
    """`enhanced` == `pcm` for a multi-frame device-pointer call: the synthesis kernel rebuilds spectra from the input, so the
    engine must notice the overlap and keep the spectrum instead; the samples equal those of a call with separate buffers."""
    torch = pytest.importorskip('torch')
    x = synth_streams(40, 12, seed=14)
    for precision in ('fp32', 'bf16'):
        kb = koala_amd.create_batch('key', 40, 6, precision, model_path=random_model)
        want = np.concatenate([kb.process(np.ascontiguousarray(x[:, c * 1536:(c + 1) * 1536])) for c in range(2)], axis=1)
        kb.reset()
        got = []
        for c in range(2):
            d = torch.from_numpy(np.ascontiguousarray(x[:, c * 1536:(c + 1) * 1536])).cuda()
            torch.cuda.synchronize()
            kb.process_device(6, d.data_ptr(), d.data_ptr())
            kb.synchronize()
            got.append(d.cpu().numpy())
        kb.delete()
        assert np.array_equal(np.concatenate(got, axis=1), want)


def test_host_call_in_place_through_the_pipelined_path(random_model):
    """`enhanced` == `pcm` for a host-memory call large enough (>= 4 MiB, more frames than one sub-chunk) to be pipelined in
    sub-chunks: exact aliasing is legal -- chunk c's copy-out lands on chunk c's own input columns, which are staged by then --
    and gives the samples of a call with separate buffers; a PARTIAL overlap is refused with a message."""
    B, T = 512, 32
    x = synth_streams(B, T, seed=41)
    kb = koala_amd.create_batch('key', B, T, 'bf16', model_path=random_model)
    want = kb.process(x)
    kb.reset()
    buf = x.copy()
    kb.process_into(buf, buf)
    assert np.array_equal(buf, want)
    kb.reset()
    pin = kb.alloc_host(T)
    pin[:] = x
    kb.process_into(pin, pin)
    assert np.array_equal(pin, want)
    big = np.zeros((B + 1, T * 256), np.int16)
    a, b = big[:B], big[1:]
    a[:] = x
    with pytest.raises(koala_amd.KoalaError) as e:
        kb.process_into(a, b)
    assert 'overlap' in str(e.value)
    kb.delete()


@pytest.mark.parametrize('B,Tmax,T,chunk', [(33, 32, 32, '16'), (33, 32, 21, '16'), (20, 8, 7, '4'), (48, 16, 16, '3'),
                                            (16, 32, 32, '0'), (5, 2, 2, '1'), (33, 64, 64, '16'), (21, 64, 57, '16:5,7,16,3')])
def test_host_pointer_calls_are_pipelined_without_changing_results(random_model, monkeypatch, B, Tmax, T, chunk):
    """Host-pointer calls run as overlapping sub-chunks (pageable buffers through staging slots, page-locked ones by direct
    strided copies): both must equal the device-pointer call bit for bit, across calls and for ragged chunk counts."""
    torch = pytest.importorskip('torch')
    # small calls are not split unless told to; 'chunk:a,b,c' also dictates the sub-chunk lengths (default: short chunks at both ends of
    # the call, long ones in the middle -- Engine::host_schedule)
    monkeypatch.setenv('KOALA_AMD_HOST_CHUNK', chunk.split(':')[0])
    if ':' in chunk:
        monkeypatch.setenv('KOALA_AMD_HOST_SCHED', chunk.split(':')[1])
    x = synth_streams(B, 2 * T, seed=5)
    kb = koala_amd.create_batch('key', B, Tmax, 'bf16', model_path=random_model, library_path=DEV_LIB)
    dx = torch.from_numpy(x).cuda()
    dy = torch.zeros_like(dx[:, :T * 256].contiguous())
    ref = []
    for c in range(2):
        dxc = dx[:, c * T * 256:(c + 1) * T * 256].contiguous()
        torch.cuda.synchronize()  # the engine runs on its own stream here
        kb.process_device(T, dxc.data_ptr(), dy.data_ptr())
        kb.synchronize()
        ref.append(dy.cpu().numpy().copy())
    kb.reset()
    pageable = [kb.process(np.ascontiguousarray(x[:, c * T * 256:(c + 1) * T * 256])) for c in range(2)]
    kb.reset()
    pin_in, pin_out = kb.alloc_host(T), kb.alloc_host(T)
    pinned = []
    for c in range(2):
        pin_in[:] = x[:, c * T * 256:(c + 1) * T * 256]
        pin_out[:] = -1
        kb.process_into(pin_in, pin_out)
        pinned.append(pin_out.copy())
    kb.delete()
    for c in range(2):
        assert np.array_equal(pageable[c], ref[c])
        assert np.array_equal(pinned[c], ref[c])


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_full_batch_unity_mask_is_a_pure_delay(unity_model, precision):
    """BASELINE batch (4096 streams x 32 frames): with mask == 1 the STFT/iSTFT pair must return the input delayed
    by 256 samples, bit for bit, for every stream -- independent of the oracle's speed."""
    B, T = 4096, 32
    rng = np.random.default_rng(0)
    x = rng.integers(-32768, 32768, size=(B, T * 256), dtype=np.int64).astype(np.int16)
    kb = koala_amd.create_batch('key', B, T, precision, model_path=unity_model)
    y = kb.process(x)
    y2 = kb.process(x)
    kb.delete()
    assert np.array_equal(y[:, 256:], x[:, :-256]) and not y[:, :256].any()
    assert np.array_equal(y2[:, :256], x[:, -256:]) and np.array_equal(y2[:, 256:], x[:, :-256])


def test_full_batch_matches_oracle_on_sampled_streams(random_model):
    B, T = 4096, 8
    base = synth_streams(64, T, seed=77)
    x = np.tile(base, (B // 64, 1))
    kb = koala_amd.create_batch('key', B, T, 'fp32', model_path=random_model)
    y = kb.process(x)
    kb.delete()
    ref = run_oracle(random_model, base)
    assert np.array_equal(y[:64], ref)
    for blk in range(1, B // 64):  # every replica of the 64 inputs is bit-identical, whatever its slot
        assert np.array_equal(y[blk * 64:(blk + 1) * 64], y[:64])


def test_mute_model_gives_silence():
    model = model_file('mute')
    x = synth_streams(5, 4, seed=1)
    kb = koala_amd.create_batch('key', 5, 4, 'fp32', model_path=model)
    assert not kb.process(x).any()
    kb.delete()


@pytest.mark.parametrize('precision,B,T', [('bf16', 1000, 5), ('bf16', 520, 3), ('fp32', 300, 2)])
def test_ragged_large_batches_take_the_fallback_kernels(random_model, precision, B, T):
    """Stream counts that are not multiples of 16 and m-tile counts that are not multiples of 256/512: these go through
    the non-split weight-stationary GEMM and the generic kernels.  Every replica of the 8 distinct inputs must be
    bit-identical wherever it sits, and the first 8 streams must match the oracle."""
    base = synth_streams(8, T, seed=31)
    x = np.tile(base, (B // 8 + 1, 1))[:B]
    kb = koala_amd.create_batch('key', B, T, precision, model_path=random_model)
    y = kb.process(x)
    y2 = kb.process(x)
    kb.delete()
    for i in range(B):
        assert np.array_equal(y[i], y[i % 8]), i
        assert np.array_equal(y2[i], y2[i % 8]), i
    o = oracle.Oracle(random_model, 8, oracle.PREC_BF16 if precision == 'bf16' else oracle.PREC_FP32)
    ref = np.concatenate([o.process(base), o.process(base)], axis=1)
    got = np.concatenate([y[:8], y2[:8]], axis=1)
    assert lsb(got, ref).max() <= (BF16_TOL if precision == 'bf16' else FP32_TOL)


@pytest.mark.parametrize('kind', ['random', 'gate', 'adaptive'])
def test_against_committed_golden_vectors(kind):
    """The engine against tests/golden/kns_v1_golden.npz (written by tools/make_golden.py from the oracle): no oracle
    run involved."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'kns_v1_golden.npz'))
    model = model_file(kind)
    for precision, tol in (('fp32', FP32_TOL), ('bf16', BF16_TOL)):
        kb = koala_amd.create_batch('key', 3, 16, precision, model_path=model)
        y = np.concatenate([kb.process(np.ascontiguousarray(g['pcm'][:, c * 4096:(c + 1) * 4096])) for c in range(3)], axis=1)
        kb.delete()
        d = lsb(y, g['%s_%s' % (kind, precision)])
        assert d.max() <= tol, (precision, int(d.max()))


_SWITCH_SCRIPT = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(tests)r)
import torch  # noqa: F401  (first: see conftest)
import koala_amd
from koala_amd.workload import synth_streams
h = hashlib.sha256()
for precision, B, T in (('bf16', 4096, 4), ('bf16', 4096, 1), ('bf16', 272, 3), ('bf16', 320, 5), ('fp32', 512, 2), ('bf16', 272, 1),
                        ('fp32', 48, 1), ('bf16', 960, 1), ('fp32', 48, 7), ('fp32', 250, 8), ('bf16', 512, 6),  # (the last three: the wavefront route)
                        ('bf16', 1040, 37), ('bf16', 2048, 32)):  # (the layer pipeline over sub-chunks of frames, mid-size batches)
    x = np.tile(synth_streams(16, 2 * T, seed=9), ((B + 15) // 16, 1))[:B]
    kb = koala_amd.create_batch('key', B, T, precision, model_path=%(model)r, library_path=%(lib)r)
    for c in range(2):
        h.update(kb.process(np.ascontiguousarray(x[:, c * T * 256:(c + 1) * T * 256])).tobytes())
    kb.delete()
# the five-frame front-end (KNS-v1.1): weight-stationary over workgroup triples vs the generic GEMM, ragged segment ends
for B, T in ((272, 9), (48, 37), (1024, 8), (272, 2), (64, 1), (4096, 1)):  # (the last three: the one-frame form of the kernel)
    x = np.tile(synth_streams(16, 2 * T, seed=11), ((B + 15) // 16, 1))[:B]
    kb = koala_amd.create_batch('key', B, T, 'bf16', model_path=%(model5)r, library_path=%(lib)r)
    for c in range(2):
        h.update(kb.process(np.ascontiguousarray(x[:, c * T * 256:(c + 1) * T * 256])).tobytes())
    kb.delete()
print('DIGEST', h.hexdigest())
'''


def test_alternative_kernels_give_identical_pcm(random_model, random5_model):
    """Every A/B switch selects other kernels for the same arithmetic (weight-streaming vs resident recurrent kernels, the
    generic vs weight-stationary GEMMs, stored vs recomputed spectrum, ...): the PCM must not change by a bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    # the product library (which reads no switch) is one arm, the developer build under each of its switches the others
    script = _SWITCH_SCRIPT % {'root': root, 'tests': os.path.join(root, 'tests'), 'model': random_model,
                               'model5': random5_model, 'lib': koala_amd.default_library_path()}
    out = subprocess.run([sys.executable, '-c', script], env=dict(os.environ, KOALA_AMD_GEMM_GENERIC='1'),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    digests['product'] = [ln for ln in out.stdout.splitlines() if ln.startswith('DIGEST')][-1]
    script = _SWITCH_SCRIPT % {'root': root, 'tests': os.path.join(root, 'tests'), 'model': random_model, 'model5': random5_model,
                               'lib': DEV_LIB}
    for switch in ('', 'KOALA_AMD_GRU_STREAM', 'KOALA_AMD_GEMM_GENERIC',
                   'KOALA_AMD_GEMM_NO_WSR', 'KOALA_AMD_NO_SMALL', 'KOALA_AMD_NO_GRAPH', 'KOALA_AMD_STORE_SPECTRUM',
                   'KOALA_AMD_DEBUG_TAPS', 'KOALA_AMD_NO_QUAD', 'KOALA_AMD_NO_HEAD_FUSE', 'KOALA_AMD_NO_STFT_FUSE', 'KOALA_AMD_WAVE_MT=0',
                   'KOALA_AMD_WAVE_MT=4096', 'KOALA_AMD_WAVE_GROUP=2', 'KOALA_AMD_NO_SPIN_WAIT', 'KOALA_AMD_PIPE_MT=0', 'KOALA_AMD_PIPE_CHUNK=8',
                   'KOALA_AMD_PIPE_WHOLE_STFT', 'KOALA_AMD_NO_WEIGHT_CACHE'):  # (KOALA_AMD_NO_QUAD: one-frame calls of large
        # batches through input GEMM + recurrent kernel instead of the one-step quad kernel, kns_gruq.hip; KOALA_AMD_WAVE_MT: multi-frame
        # calls never / always as a wavefront over (layer, frame), kns_gru.hip gru_wave_kernel; _GROUP: its m-tiles per workgroup;
        # KOALA_AMD_NO_SPIN_WAIT: one-frame host calls wait in hipStreamSynchronize instead of spinning on the frame's completion word;
        # KOALA_AMD_PIPE_MT=0: mid-size batches never as a layer pipeline over sub-chunks of frames, _CHUNK: its frames per sub-chunk,
        # _WHOLE_STFT: its analysis and synthesis as whole-call launches instead of per sub-chunk)
        env = dict(os.environ)
        if switch:
            env[switch.split('=')[0]] = switch.split('=')[1] if '=' in switch else '1'
        out = subprocess.run([sys.executable, '-c', script], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, (switch, out.stderr[-2000:])
        digests[switch] = [ln for ln in out.stdout.splitlines() if ln.startswith('DIGEST')][-1]
    assert len(set(digests.values())) == 1, digests


@pytest.mark.parametrize('precision,B,Tmax,calls', [('fp32', 21, 6, 24), ('bf16', 21, 6, 24), ('fp32', 290, 4, 24),
                                                    ('bf16', 4096, 4, 10), ('bf16', 4100, 3, 8), ('fp32', 4096, 2, 6)])
def test_random_call_sequences_keep_the_stream_state_straight(random_model, precision, B, Tmax, calls):
    """A soak over the state bookkeeping (history / overlap-add / hidden-state ping-pong buffers, the single-frame graph,
    the wavefront route of multi-frame calls, masked resets): random chunk lengths, host and device pointers, random per-stream resets --
    the engine must track the oracle, which is driven through the same sequence, call by call."""
    torch = pytest.importorskip('torch')
    rng = np.random.default_rng(42)
    prec = oracle.PREC_BF16 if precision == 'bf16' else oracle.PREC_FP32
    tol = BF16_TOL if precision == 'bf16' else FP32_TOL
    kb = koala_amd.create_batch('key', B, Tmax, precision, model_path=random_model)
    ref = oracle.Oracle(random_model, B, prec)
    worst = 0
    for call in range(calls):
        T = int(rng.integers(1, Tmax + 1))
        x = synth_streams(B, T, seed=1000 + call)
        if call % 7 == 3:
            mask = (rng.random(B) < 0.3).astype(np.uint8)
            kb.reset(mask)
            ref.reset(mask)
        elif call == calls // 2:
            kb.reset()
            ref.reset()
        if call % 3 == 0:
            dx = torch.from_numpy(x).cuda()
            dy = torch.zeros_like(dx)
            torch.cuda.synchronize()
            kb.process_device(T, dx.data_ptr(), dy.data_ptr())
            kb.synchronize()
            y = dy.cpu().numpy()
        else:
            y = kb.process(x)
        want = ref.process(x)
        worst = max(worst, int(lsb(y, want).max()))
        assert lsb(y, want).max() <= tol, (call, T)
    kb.delete()


@pytest.mark.parametrize('precision,B,T', [('fp32', 48, 37), ('bf16', 48, 37), ('bf16', 300, 19), ('fp32', 16, 64),
                                           ('bf16', 1040, 9)])
def test_long_chunks_of_small_and_odd_batches(random_model, precision, B, T):
    """Odd frame counts and small or ragged batches in ONE call: the synthesis kernel's time segments (each replays a frame
    to rebuild its overlap-add tail, and they shrink with the stream count), the wavefront over (layer, frame) with ragged
    m-tile groups and the fallback GEMMs all see shapes the throughput configuration never produces."""
    x = synth_streams(B, T, seed=77)
    kb = koala_amd.create_batch('key', B, T, precision, model_path=random_model)
    y = kb.process(x)
    y2 = kb.process(x)  # a second call continues the streams
    kb.delete()
    ref = oracle.Oracle(random_model, B, oracle.PREC_BF16 if precision == 'bf16' else oracle.PREC_FP32)
    want, want2 = ref.process(x), ref.process(x)
    tol = BF16_TOL if precision == 'bf16' else FP32_TOL
    assert lsb(y, want).max() <= tol and lsb(y2, want2).max() <= tol


def test_distinct_handles_on_distinct_threads(random_model):
    """The threading contract of DESIGN.md section 1: a handle has one caller at a time, distinct handles are independent
    -- here four of them (single-stream graph path and batch path, both precisions) run concurrently from four threads."""
    import threading
    x = synth_streams(24, 40, seed=11)
    want = {p: oracle.Oracle(random_model, 24, oracle.PREC_BF16 if p == 'bf16' else oracle.PREC_FP32).process(x)
            for p in ('fp32', 'bf16')}
    results, errors = {}, []

    def batch_worker(name, precision):
        try:
            kb = koala_amd.create_batch('key', 24, 8, precision, model_path=random_model)
            results[name] = np.concatenate([kb.process(np.ascontiguousarray(x[:, c * 2048:(c + 1) * 2048])) for c in range(5)], axis=1)
            kb.delete()
        except Exception as e:  # noqa: BLE001
            errors.append((name, repr(e), getattr(e, 'message_stack', None)))

    def single_worker(name, stream):
        try:
            k = koala_amd.create('key', model_path=random_model)
            results[name] = np.concatenate([np.array(k.process(x[stream, f * 256:(f + 1) * 256]), np.int16) for f in range(40)])
            k.delete()
        except Exception as e:  # noqa: BLE001
            errors.append((name, repr(e), getattr(e, 'message_stack', None)))

    threads = [threading.Thread(target=batch_worker, args=('b32', 'fp32')), threading.Thread(target=batch_worker, args=('b16', 'bf16')),
               threading.Thread(target=single_worker, args=('s0', 0)), threading.Thread(target=single_worker, args=('s5', 5))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert np.array_equal(results['b32'], want['fp32']) and lsb(results['b16'], want['bf16']).max() <= BF16_TOL
    assert np.array_equal(results['s0'], want['fp32'][0]) and np.array_equal(results['s5'], want['fp32'][5])


@pytest.mark.parametrize('precision,B,T', [('bf16', 32768, 16), ('bf16', 20000, 8), ('fp32', 12288, 4)])
def test_batches_far_beyond_the_bench_size(random_model, precision, B, T):
    """Eight times the bench batch (and sizes that are no multiple of anything): 64-bit addressing of every workspace, the
    GEMM stage arithmetic at other m-tile counts.  64 inputs are replicated over all slots; slot 0..63 is checked against
    the oracle and every replica must equal it bit for bit."""
    base = synth_streams(64, 2 * T, seed=123)
    reps = (B + 63) // 64
    kb = koala_amd.create_batch('key', B, T, precision, model_path=random_model)
    ref = oracle.Oracle(random_model, 64, oracle.PREC_BF16 if precision == 'bf16' else oracle.PREC_FP32)
    for c in range(2):
        chunk = np.ascontiguousarray(base[:, c * T * 256:(c + 1) * T * 256])
        y = kb.process(np.tile(chunk, (reps, 1))[:B])
        want = ref.process(chunk)
        assert lsb(y[:64], want).max() <= (BF16_TOL if precision == 'bf16' else FP32_TOL)
        full = (B // 64) * 64
        assert np.array_equal(y[:full].reshape(B // 64, 64, -1), np.broadcast_to(y[:64], (B // 64, 64, y.shape[1])))
        if B > full:
            assert np.array_equal(y[full:], y[:B - full])
    kb.delete()


def test_handles_release_what_they_hold(random_model):
    """Create / use / delete in a loop (single-stream and batch handles, both call paths): device memory must come back --
    buffers, streams, events and captured graphs are all owned by the handle."""
    torch = pytest.importorskip('torch')
    x = synth_streams(64, 4, seed=3)
    frame = x[0, :256]

    def cycle():
        k = koala_amd.create('key', model_path=random_model)
        k.process(frame)
        k.process(frame)
        k.delete()
        kb = koala_amd.create_batch('key', 64, 4, 'bf16', model_path=random_model)
        kb.process(x)
        kb.process(np.ascontiguousarray(x[:, :256]))
        pin = kb.alloc_host(4)
        pin[:] = x
        kb.process_into(pin, kb.alloc_host(4))
        kb.delete()

    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(25):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, 'device memory shrank by %d MiB over 25 create/delete cycles' % ((free0 - free1) >> 20)


def test_host_calls_on_a_caller_stream(random_model):
    """`set_stream` + host pointers: the pipelined path computes on the caller's stream (its copy streams hang off it by
    events) and single-frame calls fall back from the graph to plain launches; results as on the handle's own stream."""
    torch = pytest.importorskip('torch')
    B, T = 4096, 32
    x = np.tile(synth_streams(64, T + 1, seed=21), (B // 64, 1))
    kb = koala_amd.create_batch('key', B, T, 'bf16', model_path=random_model)
    own = [kb.process(np.ascontiguousarray(x[:, :T * 256])), kb.process(np.ascontiguousarray(x[:, T * 256:]))]
    kb.reset()
    s = torch.cuda.Stream()
    kb.set_stream(s.cuda_stream)
    mine = [kb.process(np.ascontiguousarray(x[:, :T * 256])), kb.process(np.ascontiguousarray(x[:, T * 256:]))]
    kb.set_stream(0)
    kb.delete()
    assert np.array_equal(own[0], mine[0]) and np.array_equal(own[1], mine[1])


@pytest.mark.parametrize('T,calls', [(1, 3), (32, 2)])
def test_baseline_config1_b256_fp32_at_its_stated_size(random_model, T, calls):
    """BASELINE configs[1] exactly: 256 streams, fp32 mask network -- one frame per call (16 m-tiles: the single-launch
    low-latency layers) and 32 frames per call (the wavefront route: kns_gru.hip, gru_wave_kernel).  Every stream against the oracle."""
    B = 256
    x = synth_streams(B, T * calls, seed=256)
    kb = koala_amd.create_batch('key', B, T, 'fp32', model_path=random_model)
    ref = oracle.Oracle(random_model, B)
    for c in range(calls):
        xc = np.ascontiguousarray(x[:, c * T * 256:(c + 1) * T * 256])
        d = lsb(kb.process(xc), ref.process(xc))
        assert d.max() == 0, (c, int(d.max()))
    kb.delete()


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
@pytest.mark.parametrize('B,T', [(256, 1), (272, 1), (944, 1), (960, 1), (976, 1), (1792, 1), (3072, 1), (3088, 1), (4090, 1), (4096, 1), (4112, 1), (8192, 1), (3072, 2),
                                 (3088, 2), (768, 5), (784, 5), (1024, 3), (1040, 3), (4112, 2)])
def test_dispatch_boundaries(random_model, precision, B, T):
    """The engine switches kernel families between one-frame calls below and above 192 m-tiles (bf16) / 256 m-tiles (fp32)
    (low-latency layer kernel vs input GEMM + recurrent kernel; 16 m-tiles was the edge in round 1), in bf16 already at
    59 -> 60 m-tiles when the m-tiles make whole quads (one-step fused quad kernel; 61 m-tiles do not); calls of several frames
    change between the wavefront route and the chunked kernels at 48 / 64 m-tiles (bf16) and 256 (fp32), kns_engine.cpp
    run_device().  Both sides of every edge, two calls each, every stream against the oracle."""
    base = synth_streams(128, 2 * T, seed=B)
    x = np.tile(base, ((B + 127) // 128, 1))[:B]
    kb = koala_amd.create_batch('key', B, T, precision, model_path=random_model)
    ref = oracle.Oracle(random_model, 128, oracle.PREC_BF16 if precision == 'bf16' else oracle.PREC_FP32)
    tol = BF16_TOL if precision == 'bf16' else FP32_TOL
    for c in range(2):
        y = kb.process(np.ascontiguousarray(x[:, c * T * 256:(c + 1) * T * 256]))
        want = ref.process(np.ascontiguousarray(base[:, c * T * 256:(c + 1) * T * 256]))
        assert lsb(y[:128], want).max() <= tol
        for i in range(128, B):  # replicas in other m-tiles (the last one ragged at 272 / 3088) are bit-identical
            assert np.array_equal(y[i], y[i % 128]), i
    kb.delete()


@pytest.mark.parametrize('precision,B,T,route', [('bf16', 16, 1, 1), ('bf16', 944, 1, 1), ('bf16', 960, 1, 3), ('bf16', 976, 1, 1),
                                                 ('bf16', 3072, 1, 3), ('bf16', 3088, 1, 0), ('bf16', 4096, 1, 3), ('bf16', 64, 4, 4),
                                                 ('bf16', 768, 8, 4), ('bf16', 784, 8, 0), ('bf16', 1024, 4, 4), ('bf16', 1024, 5, 0), ('bf16', 1040, 2, 0),
                                                 ('bf16', 4096, 4, 0), ('bf16', 1024, 64, 5), ('bf16', 1024, 31, 0), ('bf16', 2304, 32, 5), ('bf16', 2320, 32, 0), ('bf16', 784, 32, 5),
                                                 ('fp32', 256, 1, 1), ('fp32', 4096, 1, 1), ('fp32', 4112, 1, 0),
                                                 ('fp32', 64, 2, 4), ('fp32', 256, 32, 4), ('fp32', 4096, 2, 4), ('fp32', 4112, 2, 0)])
def test_dispatch_routes(random_model, precision, B, T, route):
    """The dispatch table at the head of Engine::run_device (kns_engine.cpp), row by row: the developer build says which kernel
    family the last call took (0 chunked, 1 low-latency layer kernel, 2 the same frame by frame, 3 one-step quad kernel, 4 wavefront over
    (layer, frame), 5 chunked kernels as a layer pipeline over two sub-chunks of frames) and what
    rode inside other launches."""
    torch = pytest.importorskip('torch')
    kb = koala_amd.create_batch('key', B, T, precision, model_path=random_model, library_path=DEV_LIB)
    x = torch.from_numpy(synth_streams(B, T, seed=1)).cuda()  # (device pointers: a large host call would be cut into sub-chunks)
    y = torch.zeros_like(x)
    torch.cuda.synchronize()
    kb.process_device(T, x.data_ptr(), y.data_ptr())
    kb.synchronize()
    got = kb.debug_read('route', T)
    kb.delete()
    assert int(got[0]) == route, got.tolist()
    fused = precision == 'bf16' and T == 1
    # [1]: the features stayed out of the call's feature buffer (only a several-frame front-end's one-frame calls: history roll);
    # [2]: the mask head rode in the synthesis launch; [3]: the spectrum was stored
    assert not bool(got[1]) and bool(got[2]) == fused and bool(got[3]) == (T == 1)


# bf16 on the DEFAULT model (the hand-built adaptive-floor gate, koala_amd.params.make_adaptive_gate): the suite's bars, like every other
# model.  This is tools/soak.py's failing case of round 4 as a test (profiles/r04_soak.txt: worst 35 LSB against a bar of 5).  Two things
# closed it in round 5 (DESIGN.md sections 2.3, 2.4, 5): the model's gain from a band level to a bin's mask was capped at 8 (it was ~70:
# a hard gate that turned one flipped bf16 rounding of an operand into tens of LSB) -- worst 5-7 LSB; and the features' logarithm became
# a polynomial shared by engine and oracle, which makes the bf16 FEATURES bit-identical on both sides -- worst 3 LSB, 99.9997 % within 1.
BF16_DEFAULT_MODEL_TOL = 5
BF16_DEFAULT_MODEL_WITHIN_1 = 0.9999


def test_bf16_default_model_soak(gate_model, test_pcm, noise_pcm):
    """2 048 streams x 80 random calls (chunk lengths 1 .. 4, host / device / in-place pointers, masked and full resets) on the
    default model in bf16 against the oracle with the same rounding points, and the reference's envelope
    (binding/python/test_koala.py:71-114) through the same engine over the same kind of call sequence."""
    torch = pytest.importorskip('torch')
    rng = np.random.default_rng(7)
    B, Tmax, calls = 2048, 4, 80
    kb = koala_amd.create_batch('key', B, Tmax, 'bf16', model_path=gate_model)
    ref = oracle.Oracle(gate_model, B, oracle.PREC_BF16)
    worst, within1, n = 0, 0, 0
    for call in range(calls):
        T = int(rng.integers(1, Tmax + 1))
        x = synth_streams(B, T, seed=5000 + call)
        r = rng.random()
        if r < 0.15:
            m = (rng.random(B) < 0.4).astype(np.uint8)
            kb.reset(m)
            ref.reset(m)
        elif r < 0.2:
            kb.reset()
            ref.reset()
        mode = rng.random()
        if mode < 0.3:
            dx = torch.from_numpy(x).cuda()
            dy = torch.zeros_like(dx)
            torch.cuda.synchronize()
            kb.process_device(T, dx.data_ptr(), dy.data_ptr())
            kb.synchronize()
            y = dy.cpu().numpy()
        elif mode < 0.45:
            dx = torch.from_numpy(x).cuda()
            torch.cuda.synchronize()
            kb.process_device(T, dx.data_ptr(), dx.data_ptr())
            kb.synchronize()
            y = dx.cpu().numpy()
        else:
            y = kb.process(x)
        d = lsb(y, ref.process(x))
        worst = max(worst, int(d.max()))
        within1 += int((d <= 1).sum())
        n += d.size
    kb.delete()
    print('default model, bf16, %d streams x %d calls: worst %d LSB, %.4f %% within 1 LSB' % (B, calls, worst, 100.0 * within1 / n))
    assert worst <= BF16_DEFAULT_MODEL_TOL, worst
    assert within1 / n >= BF16_DEFAULT_MODEL_WITHIN_1
    # the envelope, through the bf16 engine, in calls of 1 .. 4 frames: speech, noise, speech + noise as three streams
    nfr = len(test_pcm) // 256
    t, z = test_pcm[:nfr * 256], noise_pcm[:nfr * 256]
    x = np.stack([t, z, np.clip(t.astype(int) + z, -32768, 32767).astype(np.int16)])
    kb = koala_amd.create_batch('key', 3, Tmax, 'bf16', model_path=gate_model)
    outs, f = [], 0
    while f < nfr:
        T = min(int(rng.integers(1, Tmax + 1)), nfr - f)
        outs.append(kb.process(np.ascontiguousarray(x[:, f * 256:(f + T) * 256])))
        f += T
    kb.delete()
    y = np.concatenate(outs, axis=1)

    def rms(a):
        return np.sqrt(np.mean((a.reshape(-1, 256).astype(np.float64) / 32768.0) ** 2, axis=1))
    for i, want in enumerate((t, None, t)):
        out = rms(y[i])
        dev = out.copy() if want is None else np.concatenate([out[:1], np.abs(out[1:] - rms(want)[:-1])])
        assert dev.max() < 0.02, (i, float(dev.max()))


def _round_bf16(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32)


@pytest.mark.parametrize('B,T', [(19, 3), (40, 8), (16, 1)])
def test_bf16_features_are_the_oracles_bits(random_model, monkeypatch, B, T):
    """The bf16 configuration is specified to a tolerance downstream of its GEMMs and gates, but its FEATURES are exact: the fp32 FFT is
    the spec's operation for operation and the logarithm is a short polynomial written twice (kns_log_fast) -- so the engine's bf16
    feature operands equal the oracle's features rounded to bf16, bit for bit (round 5: with the hardware v_log_f32 a rounding flipped
    now and then, and a gate-like model turned that into its largest PCM differences)."""
    monkeypatch.setenv('KOALA_AMD_DEBUG_TAPS', '1')
    x = synth_streams(B, T, seed=321 + B)
    kb = koala_amd.create_batch('key', B, T, 'bf16', model_path=random_model, library_path=DEV_LIB)
    kb.process(x)
    got = kb.debug_read('features', T)
    kb.delete()
    for b in range(B):
        o = oracle.Oracle(random_model, 1, oracle.PREC_BF16)
        for t in range(T):
            _, tp = o.process_tap(x[b, t * 256:(t + 1) * 256])
            assert np.array_equal(got[t, b], _round_bf16(tp['features'])), (b, t)


def _envelope_ok(out, want):
    """The reference's acceptance criterion (binding/python/test_koala.py:71-114): per-frame RMS of the enhanced signal within 0.02 of the
    RMS of the expected one, one frame (the engine's delay) later; `want` None = pure noise in: every frame's RMS below 0.02."""
    def rms(a):
        return np.sqrt(np.mean((a.reshape(-1, 256).astype(np.float64) / 32768.0) ** 2, axis=1))
    o = rms(out)
    dev = o.copy() if want is None else np.concatenate([o[:1], np.abs(o[1:] - rms(want)[:-1])])
    return float(dev.max())


@pytest.mark.parametrize('name', ['test', 'noise', 'mixed'])
def test_bf16_single_stream_abi_on_reference_wavs(gate_model, random_model, test_pcm, noise_pcm, monkeypatch, name):
    """pv_koala_init / pv_koala_process with KOALA_AMD_PRECISION=bf16 -- the entry path BASELINE configs[4]'s bf16 latency is quoted on
    (pv_api.cpp default_precision(); hipGraph replay of the one-frame pipeline) -- on resources/audio_samples, default and random
    model: the first half of the file, ONE pv_koala_reset in the middle of the signal, then the whole file.  Every frame against the
    oracle with the same rounding points driven through the same sequence (the suite's bf16 bars); after the reset the first half must
    come out exactly as before it (the reference's test_reset, binding/python/test_koala.py:116-129); and on the default model the
    whole-file pass meets the reference's envelope (binding/python/test_koala.py:71-114)."""
    monkeypatch.setenv('KOALA_AMD_PRECISION', 'bf16')
    pcm = {'test': test_pcm, 'noise': noise_pcm,
           'mixed': np.clip(test_pcm.astype(int) + noise_pcm, -32768, 32767).astype(np.int16)}[name]
    nfr = len(pcm) // 256
    n, cut = nfr * 256, (nfr // 2) * 256
    for model in (gate_model, random_model):
        k = koala_amd.create('key', model_path=model, device='gpu:0')
        head = np.concatenate([np.array(k.process(pcm[i:i + 256]), np.int16) for i in range(0, cut, 256)])
        k.reset()  # in the middle of the signal: frame cut / 256 is never seen by the first pass
        out = np.concatenate([np.array(k.process(pcm[i:i + 256]), np.int16) for i in range(0, n, 256)])
        k.delete()
        o = oracle.Oracle(model, 1, oracle.PREC_BF16)
        ref_head = o.process(pcm[None, :cut])[0]
        o.reset()
        ref = o.process(pcm[None, :n])[0]
        d = lsb(np.concatenate([head, out]), np.concatenate([ref_head, ref]))
        print('%s, %s: max %d LSB, %.4f %% within 1' % (name, 'default' if model == gate_model else 'random', int(d.max()),
                                                         100.0 * (d <= 1).mean()))
        assert d.max() <= BF16_TOL, int(d.max())
        assert (d <= 1).mean() >= (BF16_DEFAULT_MODEL_WITHIN_1 if model == gate_model else BF16_WITHIN_1)
        assert np.array_equal(out[:cut], head)  # reset == new instance, bit for bit
        if model == random_model and name != 'noise':  # and not the fp32 engine by accident: its samples are the fp32 oracle's
            assert not np.array_equal(head, run_oracle(model, pcm[None, :cut])[0])
        if model == gate_model:
            assert _envelope_ok(out, {'test': test_pcm, 'noise': None, 'mixed': test_pcm}[name][:n] if name != 'noise' else None) < 0.02


def test_baseline_config2_b4096_t64_bf16_at_its_exact_shape(random_model):
    """BASELINE configs[2] exactly as bench.py times it: 4 096 streams x 64 frames per call, bf16 GEMMs + fp32 FFT, device pointers --
    two consecutive calls, 512 distinct streams (every one of them against the bf16-rounding oracle; the other slots hold replicas
    that must be bit-identical), and the bench line's distribution bar (>= 99.9 % of the samples within 1 LSB)."""
    torch = pytest.importorskip('torch')
    B, T, distinct = 4096, 64, 512
    base = synth_streams(distinct, 2 * T, seed=4096)
    kb = koala_amd.create_batch('key', B, T, 'bf16', model_path=random_model)
    ref = oracle.Oracle(random_model, distinct, oracle.PREC_BF16)
    for c in range(2):
        chunk = np.ascontiguousarray(base[:, c * T * 256:(c + 1) * T * 256])
        dx = torch.from_numpy(np.tile(chunk, (B // distinct, 1))).cuda()
        dy = torch.zeros_like(dx)
        torch.cuda.synchronize()
        kb.process_device(T, dx.data_ptr(), dy.data_ptr())
        kb.synchronize()
        y = dy.cpu().numpy()
        d = lsb(y[:distinct], ref.process(chunk))
        hist = np.bincount(np.minimum(d.ravel(), 8), minlength=9)
        print('4096 x 64 bf16, call %d: |gpu - oracle| histogram 0..8+: %s' % (c, hist.tolist()))
        assert d.max() <= BF16_TOL, (c, int(d.max()))
        assert (d <= 1).mean() >= 0.999, (c, float((d <= 1).mean()))
        assert np.array_equal(y.reshape(B // distinct, distinct, -1), np.broadcast_to(y[:distinct], (B // distinct, distinct, y.shape[1])))
    kb.delete()


def test_many_single_stream_handles_share_one_weight_image(random_model, gate_model):
    """The reference's contract is one handle per stream (include/pv_koala.h:26-63): 64 pv_koala_init handles alive at once, driven from
    four threads, each on its own signal -- every frame of every handle is the fp32 oracle's; the handles share ONE device image of the
    packed weights per (model, precision) (kns_engine.cpp, WeightImage), so the 2nd .. 64th cost their state and workspace only; a
    handle on another model or precision gets its own image; and the image goes when its last handle does."""
    import threading
    torch = pytest.importorskip('torch')
    nh, frames = 64, 12
    x = synth_streams(nh, frames, seed=64)
    want = run_oracle(random_model, x)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    first = koala_amd.create('key', model_path=random_model, device='gpu:0')
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    handles = [first] + [koala_amd.create('key', model_path=random_model, device='gpu:0') for _ in range(nh - 1)]
    torch.cuda.synchronize()
    free2 = torch.cuda.mem_get_info()[0]
    per_later = (free1 - free2) / float(nh - 1)
    print('first handle %.1f MiB, each of the next %d: %.2f MiB' % ((free0 - free1) / 2.0 ** 20, nh - 1, per_later / 2.0 ** 20))
    # the fp32 weight image alone is 29 MB: a later handle must cost far less than that (allocation granularity included)
    assert per_later < 8 << 20, per_later
    other = koala_amd.create('key', model_path=gate_model, device='gpu:0')  # another model: its own image, its own results
    outs, errors = [None] * nh, []

    def worker(lo, hi):
        try:
            for f in range(frames):
                for i in range(lo, hi):
                    y = np.array(handles[i].process(x[i, f * 256:(f + 1) * 256]), np.int16)
                    outs[i] = y if f == 0 else np.concatenate([outs[i], y])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    threads = [threading.Thread(target=worker, args=(q * nh // 4, (q + 1) * nh // 4)) for q in range(4)]
    for t in threads:
        t.start()
    o = np.concatenate([np.array(other.process(x[0, f * 256:(f + 1) * 256]), np.int16) for f in range(frames)])
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(nh):
        assert np.array_equal(outs[i], want[i]), i
    assert np.array_equal(o, run_oracle(gate_model, x[:1])[0])
    for h in handles[:-1]:
        h.delete()
    # the last handle still works after 63 others released the image
    last = np.array(handles[-1].process(x[nh - 1, :256]), np.int16)
    ref = oracle.Oracle(random_model, 1)
    ref.process(x[nh - 1:nh])
    assert np.array_equal(last, ref.process(x[nh - 1:nh, :256])[0])
    handles[-1].delete()
    other.delete()
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < 64 << 20  # everything came back, the shared images included


@pytest.mark.parametrize('B,T', [(1024, 64), (1040, 41), (2048, 32), (800, 57), (2560, 64)])
def test_mid_size_batches_through_the_layer_pipeline(random_model, B, T):
    """Mid-size batches (49 .. 144 m-tiles, >= 32 frames per call) run their (layer, sub-chunk of frames) grid as a wavefront over two
    streams (kns_engine.cpp run_device, kRoutePipelined): the same kernels on slices of the same buffers.  Two calls (the second one
    continues the streams from ping-pong state buffers the first one left in either parity), 128 distinct streams against the oracle,
    every replica identical, and device pointers on a caller's stream."""
    torch = pytest.importorskip('torch')
    base = synth_streams(128, 2 * T, seed=B + T)
    x = np.tile(base, ((B + 127) // 128, 1))[:B]
    kb = koala_amd.create_batch('key', B, T, 'bf16', model_path=random_model, library_path=DEV_LIB)
    ref = oracle.Oracle(random_model, 128, oracle.PREC_BF16)
    st = torch.cuda.Stream()
    kb.set_stream(st.cuda_stream)
    for c in range(2):
        dx = torch.from_numpy(np.ascontiguousarray(x[:, c * T * 256:(c + 1) * T * 256])).cuda()
        dy = torch.zeros_like(dx)
        torch.cuda.synchronize()
        kb.process_device(T, dx.data_ptr(), dy.data_ptr())
        kb.synchronize()
        if c == 0:
            assert int(kb.debug_read('route', T)[0]) == (5 if B <= 2304 else 0)
        y = dy.cpu().numpy()
        want = ref.process(np.ascontiguousarray(base[:, c * T * 256:(c + 1) * T * 256]))
        assert lsb(y[:128], want).max() <= BF16_TOL
        for i in range(128, B):
            assert np.array_equal(y[i], y[i % 128]), i
    kb.set_stream(0)
    kb.delete()


@pytest.mark.parametrize('kind', ['random', 'gate'])
def test_bf16_engine_against_the_round4_anchor(kind):
    """The bf16 engine against golden vectors that the round-5 refit of the oracle could not move: round 4's oracle (fmaf-chain GEMMs,
    polynomial e^x, full feature logarithm; tests/golden/kns_v1_golden_r4_bf16.npz, see tests/test_oracle.py).  The engine and that older
    restatement are two valid roundings of the bf16 configuration: within the suite's 5-LSB bar and >= 98 % within 1 LSB (measured: max 3,
    98.9 % -- against round 5's refitted oracle the same engine measures 99.99 %: the refit moved the oracle towards the device's sums of
    eight and correctly rounded functions, and this anchor is what shows by how much)."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'kns_v1_golden.npz'))
    r4 = np.load(os.path.join(GOLDEN, 'kns_v1_golden_r4_bf16.npz'))
    kb = koala_amd.create_batch('key', 3, 16, 'bf16', model_path=model_file(kind))
    y = np.concatenate([kb.process(np.ascontiguousarray(g['pcm'][:, c * 4096:(c + 1) * 4096])) for c in range(3)], axis=1)
    kb.delete()
    d = lsb(y, r4['%s_bf16' % kind])
    print(kind, 'engine vs round-4 oracle: max %d LSB, %.3f %% within 1' % (int(d.max()), 100.0 * (d <= 1).mean()))
    assert d.max() <= BF16_TOL and (d <= 1).mean() >= 0.98


@pytest.mark.parametrize('kind', ['random', 'adaptive'])
def test_engine_against_the_independent_numpy_restatements(kind, test_pcm, noise_pcm):
    """The engine held to the SPEC without the C oracle in between: the float64 numpy restatements of tests/test_oracle.py (KNS-v1 from
    DESIGN.md section 2; the bf16 configuration's rounding points from section 2.2), which share no code with oracle/kns_oracle.c.  fp32
    engine: within 1 LSB (float32 against float64 round-off at the final rounding), >= 99 % identical; bf16 engine: the suite's 5-LSB bar,
    >= 97 % within 1."""
    from test_oracle import _numpy_kns_v1, _numpy_kns_v1_bf16
    model = model_file(kind)
    a = 30 * 256
    pcm = (test_pcm[a:a + 60 * 256].astype(int) + noise_pcm[a:a + 60 * 256]).astype(np.int16)
    for precision, restatement, tol, share in (('fp32', _numpy_kns_v1, 1, 0.99), ('bf16', _numpy_kns_v1_bf16, BF16_TOL, 0.97)):
        want, _ = restatement(model, pcm)
        kb = koala_amd.create_batch('key', 1, 20, precision, model_path=model)
        got = np.concatenate([kb.process(np.ascontiguousarray(pcm[None, c * 5120:(c + 1) * 5120])) for c in range(3)], axis=1)[0]
        kb.delete()
        d = lsb(got, want)
        print(kind, precision, 'engine vs numpy restatement: max %d LSB, identical %.4f, within 1: %.4f' % (int(d.max()), float((d == 0).mean()), float((d <= 1).mean())))
        assert d.max() <= tol and (d <= (0 if precision == 'fp32' else 1)).mean() >= share, (precision, int(d.max()))
