"""Constant search for the default model (koala_amd.params.make_adaptive_gate), CPU only.  Scores a candidate on
  (1) the reference's acceptance envelope (binding/python/test_koala.py:71-114) on test / noise / mixed: max per-frame |rms(out) - rms(ref)|,
  (2) the hold-out bars of tests/test_holdout.py,
  (3) its sensitivity to the last bits of the bf16 configuration (tools/model_sensitivity.py: plain against jittered oracle),
and walks the constants to minimise (3) while (1) < 0.0185 and (2) hold.   python tools/gate_search.py [evaluations] [seed]"""
import json
import multiprocessing as mp
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_wav  # noqa: E402
from koala_amd import params  # noqa: E402
from koala_amd.workload import synth_streams  # noqa: E402
from oracle import oracle  # noqa: E402

RANGES = {  # name: (lo, hi, log-scale?)
    'g': (3.0, 24.0, True), 's': (6.0, 55.0, True), 'g2': (0.8, 3.0, True), 'g3': (2.5, 9.0, True), 'b3': (-2.0, 0.0, False),
    'thr': (-0.1, 0.45, False), 'z_d': (float(os.environ.get('GATE_ZD_MIN', '0.05')), 0.8, False), 'z_b': (0.3, 0.9, False), 'zb_rel': (1.0, 5.0, False), 'bz': (1.0, 4.0, False),
    'kappa': (0.8, 1.1, False), 'spread': (0.0, 0.8, False), 'thr_lf': (0.0, 0.6, False), 'hang': (0.0, 0.5, False), 'c0': (0.0, 1.0, False),
    'mask_spread': (0, 6, False), 'mu': (-9.5, -3.0, False),
}


def frame_rms(x):
    return np.sqrt(np.mean((x.reshape(-1, 256).astype(np.float64) / 32768.0) ** 2, axis=1))


def envelope(model, test_pcm, noise_pcm):
    n = len(test_pcm) // 256 * 256
    t, z = test_pcm[:n], noise_pcm[:n]
    x = np.stack([t, z, np.clip(t.astype(int) + z, -32768, 32767).astype(np.int16)])
    y = oracle.Oracle(model, 3).process(x, 1)
    ref = [t, np.zeros_like(t), t]
    dev = 0.0
    for i in range(3):
        out = frame_rms(y[i])
        want = frame_rms(ref[i])
        dev = max(dev, float(np.abs(out[1:] - want[:-1]).max()), float(out[0]))  # delay of one frame; frame 0 of the output: silence expected
    return dev


def holdout(model, test_pcm):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import test_holdout as th
    worst = {'steady_db': 99.0, 'first_frames_db': 99.0, 'speech_ratio': 9.0}
    for kind in ('white', 'pink', 'rumble'):
        for level in (0.01, 0.03):
            r = th.run_case(kind, level, test_pcm, lambda x: oracle.Oracle(model, 2).process(x, 1))
            for k in worst:
                worst[k] = min(worst[k], float(r[k]))
    worst['rise_db'] = rising(model)
    return worst


def rising(model):
    """Round 6: the case round 5's degenerate winner (bz = 4) hid from every score -- white noise at 0.01 RMS that steps UP by 6 dB after 3 s;
    median suppression 5 .. 7 s after the step (a floor tracker that cannot follow treats the louder noise like speech)."""
    rng = np.random.default_rng(5)
    n = 16000 * 10 // 256 * 256
    g = np.where(np.arange(n) < 16000 * 3, 0.01, 0.02)
    x = np.clip(np.rint(rng.standard_normal(n) * g * 32768), -32768, 32767).astype(np.int16)
    y = oracle.Oracle(model, 1).process(x[None, :], 1)[0]
    fi, fo = frame_rms(x), frame_rms(y)
    sup = 20 * np.log10(fi[:-1] / np.maximum(fo[1:], 1e-9))
    return float(np.median(sup[500:]))


def sensitivity(model, x):
    oracle.set_jitter(0)
    ref = oracle.Oracle(model, x.shape[0], oracle.PREC_BF16).process(x, 1)
    worst, big, sq, n = 0, 0, 0.0, 0
    for seed in (11, 23):
        oracle.set_jitter(seed)
        y = oracle.Oracle(model, x.shape[0], oracle.PREC_BF16).process(x, 1)
        d = np.abs(y.astype(np.int64) - ref.astype(np.int64))
        worst = max(worst, int(d.max()))
        big += int((d > 2).sum())
        sq += float((d.astype(np.float64) ** 2).sum())
        n += d.size
    oracle.set_jitter(0)
    # (the maximum and the count of large differences are rare-event statistics of a small sample; the RMS is what a search can descend)
    return worst, big, (sq / n) ** 0.5


_ctx = {}


def evaluate(kw):
    if not _ctx:
        _ctx['t'] = load_wav('test.wav')
        _ctx['z'] = load_wav('noise.wav')
        x = synth_streams(192, 100, seed=5000)
        n = 100 * 256
        t, z = _ctx['t'], _ctx['z']
        for i, w in enumerate((t, z, (t.astype(int) + z).astype(np.int16))):
            x[i] = np.resize(w[:len(w) // 256 * 256], n)
        _ctx['x'] = x
    kw = dict(kw)
    kw['mask_spread'] = int(round(kw.get('mask_spread', 0)))
    fd, path = tempfile.mkstemp(suffix='.kns', dir='/tmp')
    os.close(fd)
    try:
        params.write_params(path, params.make_adaptive_gate(**kw))
        env = envelope(path, _ctx['t'], _ctx['z'])
        res = {'env': env}
        res['gain'] = kw.get('g', DEFAULTS['g']) * kw.get('g2', DEFAULTS['g2']) * kw.get('g3', DEFAULTS['g3']) / 4.0
        if env < 0.03:  # worth the rest
            if not os.environ.get('GATE_GAIN_CAP'):
                res['sens'], res['big'], res['rms'] = sensitivity(path, _ctx['x'])
            if env < 0.02:
                res['hold'] = holdout(path, _ctx['t'])
                h = res['hold']
                if os.environ.get('GATE_GAIN_CAP') and os.environ.get('GATE_SENS') and h['steady_db'] >= 15.5 and h['first_frames_db'] >= 8.5 and h['speech_ratio'] >= 0.855 and env < 0.0192:
                    res['sens'], res['big'], res['rms'] = sensitivity(path, _ctx['x'])  # only for candidates that meet the functional bars
    finally:
        os.unlink(path)
    return kw, res


def cost(res):
    c = max(0.0, res['env'] - 0.0185) * 4000.0  # the envelope is the hard constraint
    cap = os.environ.get('GATE_GAIN_CAP')
    if cap:
        # GATE_GAIN_CAP: bound the gain of the chain level -> detector -> layer B -> mask (g x g2 x g3 / 4 per unit of x) instead of
        # scoring the sensitivity: what one flipped bf16 rounding (~0.002 in x) can do to a bin's mask is that gain times the flip
        c += 30.0 * max(0.0, res['gain'] - float(cap)) + res['env'] * 100.0
        h = res.get('hold')
        if h:
            steady_goal = float(os.environ.get('GATE_STEADY_DB', '16'))
            c += 10 * max(0.0, steady_goal - h['steady_db']) + 10 * max(0.0, 9.0 - h['first_frames_db']) + 200 * max(0.0, 0.87 - h['speech_ratio'])
            if os.environ.get('GATE_RISE_DB'):  # the rising-noise case inside the score
                c += 10 * max(0.0, float(os.environ['GATE_RISE_DB']) - h['rise_db'])
        else:
            c += 300.0  # (no hold-out figures = the envelope failed: never cheaper than a candidate that was scored on them)
        if os.environ.get('GATE_SENS'):
            c += (200.0 * res['rms']) if 'rms' in res else 40.0
        return c
    if 'sens' not in res:
        return c + 1000.0
    c += res['sens'] + 0.01 * res['big']
    h = res.get('hold')
    if h:
        c += 10 * max(0.0, 16.0 - h['steady_db']) + 10 * max(0.0, 9.0 - h['first_frames_db']) + 200 * max(0.0, 0.87 - h['speech_ratio'])
    else:
        c += 50.0
    return c


def mutate(kw, rng, width):
    out = dict(kw)
    names = list(RANGES)
    for name in rng.choice(names, size=rng.integers(1, 4), replace=False):
        lo, hi, logs = RANGES[name]
        v = out.get(name, DEFAULTS[name])
        if name == 'mask_spread':
            v = int(np.clip(v + rng.integers(-2, 3), lo, hi))
        elif logs:
            v = float(np.clip(v * np.exp(width * rng.standard_normal()), lo, hi))
        else:
            v = float(np.clip(v + width * (hi - lo) * rng.standard_normal(), lo, hi))
        out[name] = v
    return out


import inspect  # noqa: E402
DEFAULTS = {k: v.default for k, v in inspect.signature(params.make_adaptive_gate).parameters.items()}


def main():
    evals = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    start = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else {}
    rng = np.random.default_rng(seed)
    pool = mp.Pool(int(os.environ.get('GATE_SEARCH_PROCS', '7')))
    best_kw, best = evaluate(start)
    best_c = cost(best)
    print('start', json.dumps(best), 'cost %.2f' % best_c, flush=True)
    done = 0
    log = open(os.path.join(ROOT, 'build', 'gate_search_%d.jsonl' % seed), 'a')
    while done < evals:
        width = 0.25 if done < evals // 2 else 0.1
        cands = [mutate(best_kw, rng, width) for _ in range(14)]
        for kw, res in pool.imap_unordered(evaluate, cands):
            c = cost(res)
            done += 1
            log.write(json.dumps({'kw': kw, 'res': res, 'cost': c}) + '\n')
            if c < best_c:
                best_c, best_kw, best = c, kw, res
                print('%5d cost %.2f %s %s' % (done, c, json.dumps(res), json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in kw.items() if v != DEFAULTS.get(k)})), flush=True)
        log.flush()
    print('BEST', json.dumps(best_kw), json.dumps(best))


if __name__ == '__main__':
    main()
