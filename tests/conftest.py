import os
import sys
import wave

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
BUILD = os.path.join(ROOT, 'build')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # PyTorch-ROCm wheels bundle their own HIP/HSA runtime; a process that uses both torch and libpv_koala.so must
    # let torch load first so that both bind to ONE runtime (two runtimes in a process cannot both open the GPU).
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def load_wav(name):
    with wave.open(os.path.join(GOLDEN, name), 'rb') as f:
        assert f.getnchannels() == 1 and f.getsampwidth() == 2 and f.getframerate() == 16000
        return np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16).copy()


def model_file(kind, seed=1234):
    from koala_amd import params
    # (the default model's file carries the version of its constants: a stale build/ directory must not serve round 4's hard gate)
    name = '%s_%d.kns' % (kind, seed) if kind.startswith('random') else 'adaptive_v4.kns' if kind == 'adaptive' else '%s.kns' % kind
    if kind == 'gate':  # round 1's fixture-calibrated gate: its threshold is derived HERE from the reference's noise fixture
        return params.ensure_params(os.path.join(BUILD, name), kind, seed,
                                    threshold=params.noise_prior(load_wav('noise.wav')) + 2.15)
    return params.ensure_params(os.path.join(BUILD, name), kind, seed)


def synth_streams(num_streams, num_frames, seed=0):
    from koala_amd.workload import synth_streams as gen
    return gen(num_streams, num_frames, seed)


@pytest.fixture(scope='session')
def test_pcm():
    return load_wav('test.wav')


@pytest.fixture(scope='session')
def noise_pcm():
    return load_wav('noise.wav')


@pytest.fixture(scope='session')
def random_model():
    return model_file('random', 1234)


@pytest.fixture(scope='session')
def random5_model():
    """KNS-v1.1: random weights behind the reference file's five-frame front-end ([1285, 271])."""
    return model_file('random5', 1234)


@pytest.fixture(scope='session')
def gate_model():
    """The default model: the hand-built spectral gate with an adaptive noise floor (nothing derived from an audio file)."""
    return model_file('adaptive')


@pytest.fixture(scope='session')
def prior_gate_model():
    """Round 1's gate, whose fixed threshold is the mean spectrum of the reference's noise fixture (kept for comparison)."""
    return model_file('gate')


@pytest.fixture(scope='session')
def unity_model():
    return model_file('unity')


@pytest.fixture(scope='session')
def native_library():
    from koala_amd import _util
    return _util.build_native()


def frame_rms(pcm):
    return float(np.sqrt(np.mean((np.asarray(pcm, np.float64) / 32768.0) ** 2)))
