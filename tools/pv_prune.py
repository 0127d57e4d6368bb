#!/usr/bin/env python3
"""
BUILD-CONTAINER ONLY (reads /root/reference/lib/common/koala_params.pv; nothing of it is copied into the repository).

Prunes the import-hypothesis space of koala_amd/pv_import.py with what the BYTES constrain, before anything is scored on
the acceptance envelope (round 2 scored 2 400 hypotheses on the envelope alone -- the weakest signal there is):
  S1  gate order      the three 271-column groups of the 813-byte trailers (read as biases): a GRU's update gate is the one
                      that is biased to hold its state, i.e. whose bias stands apart from the other two, consistently over the
                      16 GRU blocks; an order that does not put that group on `z` is dropped
  S2  weight shift    the pre-activations of every GRU layer, simulated on the reference's own WAVs: shifts under which the
                      gates are saturated (|x| > 8 for most units) or inert (std < 0.05) are dropped
  S3  bias shift      bias spread against the spread of the weight-driven part of the same pre-activation: a bias shift under
                      which the biases dwarf (> 8 x) or vanish against (< 1/64 x) the signal is dropped
  S4  per-stage int16 (3056, 1013, 1379, 1713) against the simulated range of each stage's head pre-activation and output:
                      recorded (no reading of it is part of the Hypothesis yet)
  S5  feature tables  mean / scale divisors under which the normalised features of the WAVs leave [-8, 8] or collapse are dropped
The surviving sub-space is then sampled and scored like round 2's (envelope + the two gain descriptors), with the five-frame
front-end (front_tap = 5 / 6) always on.  Everything is written to profiles/r03_pv_import_search.json.
"""
import itertools
import json
import multiprocessing as mp
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
PV = '/root/reference/lib/common/koala_params.pv'
from koala_amd import params, pv_import  # noqa: E402
import pv_hypotheses  # noqa: E402

H = params.HIDDEN


def features_of_fixtures(mean, scale):
    from conftest import load_wav
    out = []
    for name in ('test.wav', 'noise.wav'):
        x = load_wav(name).astype(np.float64) / 32768.0
        n = len(x) // 256 * 256
        win = np.sin(np.pi * np.arange(512) / 512)
        fr = np.stack([x[i:i + 512] * win for i in range(0, n - 512 + 1, 256)])
        logp = np.log(np.abs(np.fft.rfft(fr, axis=1)) ** 2 + 1e-10)
        out.append((logp - mean) * scale)
    return np.concatenate(out)


def main():
    m = pv_import.read_pv(PV)
    rec = {'tool': 'tools/pv_prune.py', 'statistics': {}}

    # ---- S1: which 271-column group of the trailers is "different"
    groups = {'ih': np.zeros(3), 'hh': np.zeros(3)}
    per_block = []
    for bi, blk in enumerate(m.blocks):
        if blk.cols != params.G3:
            continue
        t = blk.trailer.astype(np.float64).reshape(3, H)
        kind = 'ih' if (bi % 5) in (0, 2) else 'hh'
        groups[kind] += t.mean(axis=1)
        per_block.append({'block': bi, 'kind': kind, 'group_means': [round(float(v), 2) for v in t.mean(axis=1)],
                          'group_stds': [round(float(v), 2) for v in t.std(axis=1)]})
    total = groups['ih'] + groups['hh']
    # the update gate: the group whose summed bias (b_ih + b_hh is what the gate sees) is furthest from the other two
    dist = [abs(total[i] - np.mean(np.delete(total, i))) for i in range(3)]
    z_group = int(np.argmax(dist))
    votes = [int(np.argmax([abs(b['group_means'][i] - np.mean(np.delete(b['group_means'], i))) for i in range(3)])) for b in per_block]
    consistent = float(np.mean([v == z_group for v in votes]))
    orders = [''.join(p) for p in itertools.permutations('rzn')]
    keep_orders = [o for o in orders if o[z_group] == 'z'] if consistent >= 0.6 else orders
    rec['statistics']['S1_gate_order'] = {
        'summed_group_means_ih_plus_hh': [round(float(v), 2) for v in total], 'distinct_group': z_group,
        'blocks_voting_for_it': consistent, 'per_block': per_block, 'orders_kept': keep_orders,
        'verdict': 'discriminates' if len(keep_orders) < 6 else 'does not discriminate (no group stands apart in >= 60 % of the blocks)'}

    # ---- S5: feature tables on the reference's WAVs
    keep_feat = []
    feat_stats = []
    for md in (256.0, 512.0, 1024.0):
        for sd in (2048.0, 4096.0, 8192.0):
            for l2 in (False, True):
                ln2 = np.log(2.0)
                mean, scale = m.table_mean / md, m.table_scale / sd
                if l2:
                    mean, scale = mean * ln2, scale / ln2
                f = features_of_fixtures(mean, scale)
                st = {'mean_div': md, 'scale_div': sd, 'log2': l2, 'feature_mean': round(float(f.mean()), 3),
                      'feature_std': round(float(f.std()), 3), 'frac_outside_8': round(float((np.abs(f) > 8).mean()), 4)}
                ok = st['frac_outside_8'] < 0.01 and 0.1 < st['feature_std'] < 4.0 and abs(st['feature_mean']) < 3.0
                st['kept'] = bool(ok)
                feat_stats.append(st)
                if ok:
                    keep_feat.append((md, sd, l2))
    rec['statistics']['S5_feature_tables'] = {'combinations': feat_stats, 'kept': len(keep_feat), 'of': len(feat_stats)}

    # ---- S2 / S3: pre-activation spread of the first GRU layer per (front shift, weight shift, bias shift), features as kept
    md, sd, l2 = keep_feat[0] if keep_feat else (512.0, 4096.0, False)
    mean, scale = m.table_mean / md, m.table_scale / sd
    if l2:
        mean, scale = mean * np.log(2.0), scale / np.log(2.0)
    f = features_of_fixtures(mean, scale)
    stack = np.concatenate([np.concatenate([np.zeros((k, 257)), f[:len(f) - k]]) for k in (4, 3, 2, 1, 0)], axis=1)  # oldest first
    fw = m.front.weights.astype(np.float64)
    w_ih = m.blocks[0].weights.astype(np.float64)
    spread, keep_fs, keep_ws, keep_bs = [], [], [], []
    for fs in range(5, 14):  # (round 2 searched 5 .. 9 only: almost all of that range drives the gates into saturation, see the record)
        e = stack @ (fw * 2.0 ** -fs)
        e_std = float(e.std())
        ok_f = 0.05 < e_std < 8.0
        if ok_f:
            keep_fs.append(fs)
        for ws in range(5, 13):
            pre = e @ (w_ih * 2.0 ** -ws)
            st = {'front_shift': fs, 'weight_shift': ws, 'embedding_std': round(e_std, 3), 'preact_std': round(float(pre.std()), 3),
                  'frac_saturated_abs_gt_8': round(float((np.abs(pre) > 8).mean()), 4)}
            st['kept'] = bool(ok_f and 0.05 < st['preact_std'] and st['frac_saturated_abs_gt_8'] < 0.5)
            spread.append(st)
    keep_pairs = [(s['front_shift'], s['weight_shift']) for s in spread if s['kept']]
    bias = m.blocks[0].trailer.astype(np.float64)
    bias_stats = []
    ref_std = np.median([s['preact_std'] for s in spread if s['kept']]) if keep_pairs else 1.0
    for bs in (3, 4, 5, 6, 7):
        r = float((bias * 2.0 ** -bs).std() / max(ref_std, 1e-9))
        ok = 1.0 / 64 < r < 8.0
        bias_stats.append({'bias_shift': bs, 'bias_std_over_median_preact_std': round(r, 3), 'kept': bool(ok)})
        if ok:
            keep_bs.append(bs)
    rec['statistics']['S2_weight_shift'] = {'first_layer_spread': spread, 'front_weight_pairs_kept': len(keep_pairs), 'of': 72,
                                             'of_round2s_25_pairs_kept': sum(1 for a, b in keep_pairs if a <= 9 and b <= 9)}
    rec['statistics']['S3_bias_shift'] = {'per_shift': bias_stats, 'kept': keep_bs}

    # ---- S4: the per-stage int16
    rec['statistics']['S4_stage_int16'] = {
        'values': m.stage_tail, 'as_q12': [round(v / 4096.0, 4) for v in m.stage_tail], 'as_q15': [round(v / 32768.0, 5) for v in m.stage_tail],
        'head_widths': list(params.HEADS),
        'note': 'no monotone relation to the head widths (1, 5, 40, 257) or to the head weights\' row norms; ratios between stages 3.02 / '
                '0.73 / 0.81 -- recorded, not part of any hypothesis yet',
        'head_weight_rms': [round(float(np.sqrt(np.mean(m.blocks[5 * s + 4].weights.astype(np.float64) ** 2))), 2) for s in range(4)]}

    # ---- the pruned space, and what is left of round 2's
    full = 1
    for v in pv_hypotheses.SPACE.values():
        full *= len(dict.fromkeys(v))
    space = dict(pv_hypotheses.SPACE)
    space['gate_order'] = keep_orders
    space['bias_shift'] = keep_bs or space['bias_shift']
    space['front_tap'] = [5, 6]  # the five-frame front-end, both stacking orders (the GPU engine runs it since round 3)
    r2_pairs = [pr for pr in keep_pairs if pr[0] <= 9 and pr[1] <= 9]
    pruned = len(keep_orders) * len(space['bias_shift']) * len(r2_pairs or [0]) * len(keep_feat or [0]) * 2 * 5 * 2 * 4 * 4
    rec['space'] = {'round2_size': full, 'pruned_size_within_round2s_ranges': int(pruned), 'factor': round(full / max(pruned, 1), 1),
                    'extended': 'front_shift up to 13 and weight_shift up to 12 added: the statistics say that is where unsaturated gates are',
                    'kept': {'gate_order': keep_orders, 'bias_shift': space['bias_shift'], 'front_weight_pairs': keep_pairs,
                             'feature_tables': keep_feat, 'front_tap': [5, 6]}}

    # ---- score a sample of the pruned space (same scoring as round 2)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    rnd = random.Random(20260929)
    hyps = []
    while len(hyps) < n and keep_pairs and keep_feat:
        fs, ws = rnd.choice(keep_pairs)
        md, sd, l2 = rnd.choice(keep_feat)
        hyps.append(dict(weight_shift=ws, front_shift=fs, bias_shift=rnd.choice(space['bias_shift'] + [8, 9]),
                         front_bias_shift=rnd.choice(pv_hypotheses.SPACE['front_bias_shift']), front_tap=rnd.choice([5, 6]),
                         mean_div=md, scale_div=sd, log2_features=l2, gate_order=rnd.choice(keep_orders),
                         y_first=rnd.choice([True, False]), head_shift=rnd.choice(pv_hypotheses.SPACE['head_shift']),
                         head_bias_shift=rnd.choice(pv_hypotheses.SPACE['head_bias_shift'])))
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        results = pool.map(pv_hypotheses.evaluate, hyps, chunksize=8)
    score = lambda d: max(d['speech'], d['noise'], d['mixed'])  # noqa: E731
    scored = sorted(results, key=lambda r: score(r[1]))
    scores = np.array([score(r[1]) for r in scored]) if scored else np.array([1.0])
    useful = [r for r in results if r[1]['noise_gain_db'] <= -6.0 and abs(r[1]['speech_gain_db']) <= 3.0]
    rec['search'] = {
        'hypotheses_tried': len(results), 'passing': int((scores < 0.02).sum()), 'suppressor_like': len(useful),
        'score_quantiles': {q: float(np.quantile(scores, float(q))) for q in ('0.0', '0.01', '0.1', '0.5', '0.9')},
        'best': [{'hypothesis': h, 'metrics': d} for h, d in scored[:8]],
        'best_suppressor_like': [{'hypothesis': h, 'metrics': d} for h, d in sorted(useful, key=lambda r: r[1]['noise_gain_db'])[:5]]}
    r2 = json.load(open(os.path.join(ROOT, 'profiles', 'r02_pv_import_search.json')))
    rec['round2'] = {k: r2[k] for k in ('hypotheses_tried', 'passing', 'suppressor_like', 'score_quantiles')}
    rec['default_hypothesis'] = r2['default_hypothesis']
    path = os.path.join(ROOT, 'profiles', 'r03_pv_import_search.json')
    json.dump(rec, open(path, 'w'), indent=1)
    print(json.dumps({'space': rec['space'], 'S1': {k: rec['statistics']['S1_gate_order'][k] for k in ('summed_group_means_ih_plus_hh', 'distinct_group', 'blocks_voting_for_it', 'orders_kept')},
                      'S3': rec['statistics']['S3_bias_shift'], 'S5_kept': rec['statistics']['S5_feature_tables']['kept'],
                      'search': {k: rec['search'][k] for k in ('hypotheses_tried', 'passing', 'suppressor_like', 'score_quantiles')}}, indent=1))


if __name__ == '__main__':
    main()
