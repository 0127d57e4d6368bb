"""
Synthetic-noise check of the default model (koala_amd.params.make_adaptive_gate): seeded white, pink and low-frequency "rumble" noise
at two levels, alone and mixed with the speech fixture -- and one kind it is NOT expected to handle: speech-like babble (amplitude-
modulated formant resonances).  SINCE ROUND 5 THIS IS A TUNING SET, not a hold-out: tools/gate_search.py reads these very bars in its
cost (the file keeps its name; the set no search has seen is tests/test_validation.py).  CPU oracle; the GPU engine is checked against
the same oracle sample for sample elsewhere.

Bars (levels 0.01 and 0.03 RMS, i.e. around and above the reference's noise fixture at 0.023): stationary noise alone is
suppressed by >= 20 dB after 0.5 s (measured 21.3-28.9 dB with round 6's constants, adaptive-gate-v4; round 5's v3: 16.7-23.5 dB; round 4's
hard gate: 21-33 dB, at the price of a bf16 path that amplified single rounding flips -- DESIGN.md section 2.4) and by >= 9 dB in the first
four frames, while the floor tracker is still coming down from its closed start (measured 9.6-10.7 dB; the reference's own test checks
those frames against an absolute 0.02 RMS, test_koala.py:94-95); with speech on top, speech-active frames keep >= 85 % (median) of the clean
speech's RMS (measured 0.87-0.97).  Babble: what is measured is recorded (10 dB), the bar is only "does no harm".
"""
import numpy as np
import pytest
from scipy.signal import lfilter

from conftest import model_file
from oracle import oracle


def rms(x, axis=-1):
    return np.sqrt(np.mean((np.asarray(x, np.float64) / 32768.0) ** 2, axis=axis))


def synth_noise(kind, n, rng):
    w = rng.standard_normal(n)
    if kind == 'white':
        return w
    if kind == 'pink':
        return lfilter([0.049922035, -0.095993537, 0.050612699, -0.004408786],
                       [1, -2.494956002, 2.017265875, -0.522189400], w)
    if kind == 'rumble':
        hum = lfilter([1], [1, -1.97, 0.9704], rng.standard_normal(n))
        return hum / np.std(hum) + 0.15 * w
    if kind == 'babble':
        t = np.arange(n) / 16000.0
        b = np.zeros(n)
        for i, (f0, r) in enumerate([(300, 0.97), (600, 0.96), (1100, 0.95), (1800, 0.95), (2600, 0.94), (3500, 0.94)]):
            th = 2 * np.pi * f0 / 16000
            x = lfilter([1], [1, -2 * r * np.cos(th), r * r], rng.standard_normal(n))
            b += x / np.std(x) * (0.6 + 0.4 * np.sin(2 * np.pi * (1.3 + 0.7 * i) * t + rng.uniform(0, 6.28)))
        return b
    raise ValueError(kind)


def run_case(kind, level, test_pcm, engine=None):
    """`engine(pcm[2, n]) -> enhanced[2, n]`; default: the CPU oracle with the default model"""
    n = len(test_pcm) // 256 * 256
    rng = np.random.default_rng(777)
    x = synth_noise(kind, n, rng)
    noise = np.clip(np.rint(x / np.std(x) * level * 32768), -32768, 32767).astype(np.int16)
    mix = np.clip(test_pcm[:n].astype(int) + noise, -32768, 32767).astype(np.int16)
    y = (engine or oracle.Oracle(model_file('adaptive'), 2).process)(np.stack([noise, mix]))
    clean = rms(test_pcm[:n].reshape(-1, 256))
    out = rms(y.reshape(2, -1, 256))
    active = clean[:-1] > 0.03
    return {
        'steady_db': 20 * np.log10(rms(noise[8000:]) / max(rms(y[0][8000 + 256:]), 1e-9)),
        'first_frames_db': 20 * np.log10(rms(noise[:1024]) / max(rms(y[0][256:1280]), 1e-9)),
        'speech_ratio': float(np.median(out[1][1:][active] / clean[:-1][active])),
    }


@pytest.mark.parametrize('kind', ['white', 'pink', 'rumble'])
@pytest.mark.parametrize('level', [0.01, 0.03])
def test_stationary_holdout_noise(kind, level, test_pcm):
    r = run_case(kind, level, test_pcm)
    print(kind, level, r)
    assert r['steady_db'] >= 20.0, r
    assert r['first_frames_db'] >= 9.0, r
    assert r['speech_ratio'] >= 0.85, r


def test_babble_is_out_of_reach_of_a_spectral_gate(test_pcm):
    """Speech-like noise looks like speech to a level detector: measured ~5-6 dB; the bar is that speech is not damaged."""
    r = run_case('babble', 0.01, test_pcm)
    print('babble', r)
    assert r['steady_db'] >= 3.0 and r['speech_ratio'] >= 0.9, r


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_holdout_noise_through_the_gpu_engine(precision, test_pcm):
    """The same hold-out cases through libpv_koala.so on the MI355X (the CPU-only cases above are deselected in the GPU job):
    the figures to quote for the default model are these, not the envelope it was tuned on."""
    import koala_amd
    kb = koala_amd.create_batch('key', 2, 73, precision, model_path=model_file('adaptive'))

    def engine(x):
        kb.reset()
        return np.concatenate([kb.process(np.ascontiguousarray(x[:, i:i + 73 * 256])) for i in range(0, x.shape[1], 73 * 256)], axis=1)
    for kind in ('white', 'pink', 'rumble'):
        for level in (0.01, 0.03):
            r = run_case(kind, level, test_pcm, engine)
            print(precision, kind, level, r)
            assert r['steady_db'] >= 20.0 and r['first_frames_db'] >= 9.0 and r['speech_ratio'] >= 0.85, (kind, level, r)
    r = run_case('babble', 0.01, test_pcm, engine)
    print(precision, 'babble', r)
    assert r['steady_db'] >= 3.0 and r['speech_ratio'] >= 0.9, r
    kb.delete()


ROUND4_HARD_GATE = dict(c0=0.9, s=51.25, bz=2.75, kappa=1.0, g=22.275, thr=0.24, z_d=0.1, g2=1.521, z_b=0.64, g3=8.25, b3=-0.85,
                        spread=0.416, thr_lf=0.405, zb_rel=3.6, hang=0.3)


def test_default_model_is_insensitive_to_the_last_bits_of_the_bf16_configuration(tmp_path, test_pcm, noise_pcm):
    """The bf16 configuration is specified to a tolerance: two valid implementations differ in the last bit of a transcendental or a
    GEMM output now and then, and such a bit occasionally flips a bf16 / fp16 rounding downstream.  What that does to the PCM is a
    property of the MODEL (its gain from an operand to a bin's mask).  The oracle plays the second implementation
    (kns_oracle_set_jitter); the default model must stay within the suite's bf16 bar with room to spare for the larger sample the
    GPU soak takes, and round 4's hard gate -- kept here as the counter-example -- must not."""
    from koala_amd import params
    from koala_amd.workload import synth_streams

    def distance(model):
        x = synth_streams(128, 100, seed=5000)
        n = 100 * 256
        t, z = test_pcm, noise_pcm
        for i, w in enumerate((t, z, (t.astype(int) + z).astype(np.int16))):
            x[i] = np.resize(w[:len(w) // 256 * 256], n)
        oracle.set_jitter(0)
        ref = oracle.Oracle(model, x.shape[0], oracle.PREC_BF16).process(x)
        try:
            oracle.set_jitter(11)
            y = oracle.Oracle(model, x.shape[0], oracle.PREC_BF16).process(x)
        finally:
            oracle.set_jitter(0)
        d = np.abs(y.astype(np.int64) - ref.astype(np.int64))
        return int(d.max()), float((d <= 1).mean())
    worst, within1 = distance(model_file('adaptive'))
    hard = str(tmp_path / 'hard.kns')
    params.write_params(hard, params.make_adaptive_gate(**ROUND4_HARD_GATE))
    hworst, hwithin1 = distance(hard)
    print('default model: worst %d LSB, %.4f %% within 1 | round-4 hard gate: worst %d LSB, %.4f %% within 1' % (worst, 100 * within1, hworst, 100 * hwithin1))
    assert worst <= 4 and within1 >= 0.9995, (worst, within1)
    assert hworst >= 2 * worst, (hworst, worst)


def test_rising_noise_level_is_a_known_limitation():
    """What the default model does NOT do, recorded so that nobody has to find out: its floor tracker falls fast and rises very slowly
    (that asymmetry is what protects speech), so a noise level that steps UP is treated like speech for a long time -- white noise at
    0.01 RMS is suppressed by ~29 dB, and after a +6 dB step by ~10 dB still 18 s later (round 5's constants: 19 and 7 dB).  Since round 6 the
    constant search scores a rising level (tools/gate_search.py, `rising`), which bought those 3 dB and rejects constants that freeze the
    floor; the tracker's structure still does not follow a step within seconds.  The bar: never amplifies, the level BEFORE the step is
    handled like any stationary noise, and >= 8 dB are kept after it."""
    rng = np.random.default_rng(5)
    n = 16000 * 24 // 256 * 256
    g = np.where(np.arange(n) < 16000 * 4, 0.01, 0.02)
    x = np.clip(np.rint(rng.standard_normal(n) * g * 32768), -32768, 32767).astype(np.int16)
    y = oracle.Oracle(model_file('adaptive'), 1).process(x[None, :])[0]
    fi, fo = rms(x.reshape(-1, 256)), rms(y.reshape(-1, 256))
    sup = 20 * np.log10(fi[:-1] / np.maximum(fo[1:], 1e-9))
    before, late = float(np.median(sup[125:187])), float(np.median(sup[-187:-62]))
    print('white noise 0.01 RMS: %.1f dB; 18 s after a +6 dB step: %.1f dB' % (before, late))
    assert before >= 20.0
    assert late >= 8.0 and late < before  # recorded: the floor has not followed
