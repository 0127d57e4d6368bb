#!/usr/bin/env python3
"""
bench.py -- throughput of the pv_koala_process hot path on MI355X (BASELINE.json metric: 16 kHz frames/sec/GPU at
batch 4096, configs[2]: bf16 mask GEMMs on MFMA + fp32 FFT).

One "step" = one pv_koala_batch_process_chunk call: 4096 streams x 64 frames (1.024 s of audio per stream) per GPU,
int16 PCM already resident in HBM, enhanced PCM left in HBM.  N > 1: one process per GPU (torchrun), each rank owns
its own 4096 streams (weak scaling, no data-path collective); the only collective is the final RCCL all-reduce of
{frames, max elapsed}.

Prints ONE JSON line (rank 0).  Besides the driver's fields it carries
  roofline      dominant kernel (by device time) priced against its bound: algorithmic work per launch / mean launch
                duration measured with HIP events on the engine's stream in a second, identical pass
  stages        the same for every kernel class
  cpu_baseline  the CPU oracle (a "port": plain-C restatement, OpenMP over streams) on a bounded sample of the same
                workload on this box's host cores -- a reported baseline, not the target
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3}  # dense peaks, no sparsity

H = 271
G3 = 813
HEADS = (1, 5, 40, 257)
# algorithmic work per stream-frame (unpadded dims)
BYTES_ANALYSIS = 512 + 512 + 512 + 2056 + 1028
BYTES_SYNTHESIS = 2056 + 1028 + 1024 + 1024 + 512
MAC_GEMM_IN = (271 + 272 + 276 + 311 + 4 * 271) * G3      # 8 input-side GEMMs (W_ih)
MAC_GRU = 8 * H * G3                                      # 8 recurrent GEMMs (W_hh)
MAC_HEAD = 257 * H + H * sum(HEADS)                       # front-end + 4 heads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--prime-seconds', type=float, default=0.5,
                    help='untimed run of the same step before the warmup steps: the MI355X takes a few hundred ms of load to '
                         'leave its idle clocks (sclk ~100 MHz), and the workload then runs against the 1400 W board power cap')
    ap.add_argument('--streams', type=int, default=4096, help='streams per GPU')
    ap.add_argument('--frames', type=int, default=64, help='frames per stream per call (64 = 1.02 s of audio)')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--library', default=None, help='alternative libpv_koala.so (developer A/B runs)')
    ap.add_argument('--dist-backend', default='nccl', help='nccl (= RCCL over xGMI) for real runs; gloo only to test the '
                    'multi-rank code path on a single-GPU box together with KOALA_BENCH_SHARE_GPU=1')
    args = ap.parse_args()

    import numpy as np
    import torch

    import koala_amd
    from koala_amd import params
    from koala_amd.sharding import aggregate_throughput, shard_range
    from koala_amd.workload import synth_streams

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if os.environ.get('KOALA_BENCH_SHARE_GPU'):
        local_rank = 0  # test mode: every rank drives GPU 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    else:
        torch.cuda.set_device(local_rank)
    if args.gpus != world and rank == 0 and world > 1:
        print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)

    koala_amd.build_native()
    model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
    B, T = args.streams, args.frames
    first, _ = shard_range(B * world, rank, world)

    # synthetic input: 64 distinct seeded streams tiled over the batch (generation cost, not data-path cost)
    base = synth_streams(64, T, seed=1234, first_stream=(first % 4096))
    x = np.tile(base, ((B + 63) // 64, 1))[:B]
    dx = torch.from_numpy(x).cuda()
    dy = torch.empty_like(dx)

    kb = koala_amd.create_batch('bench', B, T, args.precision, model_path=model, device='gpu:%d' % local_rank,
                               library_path=args.library)
    kb.set_stream(torch.cuda.current_stream().cuda_stream)

    def step():
        kb.process_device(T, dx.data_ptr(), dy.data_ptr())

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    t_prime = time.perf_counter()
    while time.perf_counter() - t_prime < args.prime_seconds:  # clock ramp, untimed (see --prime-seconds)
        step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    frames_total, elapsed_max = aggregate_throughput(B * T * args.steps, elapsed)
    value = frames_total / elapsed_max

    # ---- per-kernel pass (same workload, HIP events around every launch on the engine's stream)
    prof_steps = max(2, min(args.steps, 20))
    kb.profile_enable(True)
    for _ in range(prof_steps):
        step()
    prof = kb.profile_read()
    kb.profile_enable(False)
    frames_per_launch = B * T
    work = {
        'analysis': ('hbm', BYTES_ANALYSIS * frames_per_launch, 1),
        'gemm_input': ('mfma', 2.0 * MAC_GEMM_IN / 8 * frames_per_launch, 8),
        'gru_recurrent': ('mfma', 2.0 * MAC_GRU / 8 * frames_per_launch, 8),
        'gemm_head': ('mfma', 2.0 * MAC_HEAD / 5 * frames_per_launch, 5),
        'synthesis': ('hbm', BYTES_SYNTHESIS * frames_per_launch, 1),
    }
    stages = {}
    dev_ms = sum(v['ms'] for v in prof.values()) / prof_steps
    for name, (bound, per_launch, launches) in work.items():
        n = prof[name]['launches']
        if n == 0:  # e.g. small batches: the input GEMMs run inside the frame-by-frame recurrent launches
            continue
        actual = n / float(prof_steps)  # launches per step as measured (small batches step the layers frame by frame)
        per_launch = per_launch * launches / actual
        if name == 'gru_recurrent' and prof['gemm_input']['launches'] == 0:
            per_launch += 2.0 * MAC_GEMM_IN * frames_per_launch / actual
        ms = prof[name]['ms'] / n
        if bound == 'hbm':
            achieved, peak, unit = per_launch / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, 'GB/s'
        else:
            achieved, peak, unit = per_launch / (ms * 1e-3) / 1e12, MFMA_PEAK_TFLOPS[args.precision], 'TFLOP/s'
        stages[name] = {'bound': bound, 'achieved': round(achieved, 3), 'peak': peak, 'unit': unit,
                        'frac': round(achieved / peak, 5), 'avg_launch_ms': round(ms, 5),
                        'launches_per_step': round(actual, 2) if actual != int(actual) else int(actual),
                        'share_of_device_time': round(ms * actual / dev_ms, 4) if dev_ms else None}
    dominant = max(stages, key=lambda k: stages[k]['avg_launch_ms'] * stages[k]['launches_per_step'])
    roofline = dict(stages[dominant])
    roofline['kernel'] = dominant
    # HBM bytes per launch come from separate rocprofv3 --pmc passes of this same command (tools/pmc_run.sh ->
    # profiles/*_pmc.json; FETCH_SIZE doubled per the gfx950 correction); they cannot be collected inside a timed run
    roofline['traffic'] = None
    try:
        import glob
        pmc_files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc.json')))
        pmc = json.load(open(pmc_files[-1])) if pmc_files else {}
        wl = pmc.pop('_workload', None)
        if wl == {'streams_per_gpu': B, 'frames_per_call': T, 'dtype': args.precision}:
            for entry in pmc.values():
                if entry.get('class') == dominant and 'hbm_bytes' in entry:
                    roofline['traffic'] = entry['hbm_bytes']
                    roofline['traffic_source'] = os.path.basename(pmc_files[-1])
            for name in stages:
                for entry in pmc.values():
                    if entry.get('class') == name and 'hbm_bytes' in entry:
                        stages[name]['traffic'] = entry['hbm_bytes']
    except Exception:
        pass

    # ---- parity spot check inside the bench: the timed engine vs the CPU oracle on a few streams
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        ncores = os.cpu_count() or 1
        # calibrate, then size the sample for ~10-20 s of CPU work
        o = oracle.Oracle(model, 1024, oracle.PREC_FP32)
        xs = np.ascontiguousarray(np.tile(base, (16, 1))[:, :8 * 256])
        c0 = time.perf_counter()
        o.process(xs)
        rate = 1024 * 8 / (time.perf_counter() - c0)
        ns = int(min(16384, max(64, (rate * 15) // (T * 64) * 64)))
        o = oracle.Oracle(model, ns, oracle.PREC_FP32)
        xs = np.ascontiguousarray(np.tile(base, ((ns + 63) // 64, 1))[:ns])
        c0 = time.perf_counter()
        ref = o.process(xs)
        dt = time.perf_counter() - c0
        cpu = {'value': round(ns * T / dt, 1), 'unit': 'frames/s', 'cores': ncores, 'kind': 'port',
               'sample': '%d streams x %d frames of the same synthetic workload, oracle/kns_oracle.c fp32, OpenMP '
                         'over stream blocks, %.1f s' % (ns, T, dt)}
        # and use it as a checker of what was just timed (first call after a reset)
        kb.reset()
        step()
        torch.cuda.synchronize()
        got = dy[:64].cpu().numpy().astype(np.int64)
        want = oracle.Oracle(model, 64, oracle.PREC_BF16 if args.precision == 'bf16' else oracle.PREC_FP32).process(
            np.ascontiguousarray(x[:64])).astype(np.int64)
        cpu['gpu_vs_oracle_max_lsb'] = int(np.abs(got - want).max())

    kb.delete()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    line = {
        'metric': '16 kHz frames/sec (batch 4096 streams per GPU)',
        'value': round(value, 1),
        'unit': 'frames/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': round(elapsed_max / args.steps * 1e3, 4),
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': args.precision,
        'data': 'synthetic (seeded AR(2)+AM speech-like int16 streams; seeded random KNS1 parameters)',
        'config': {
            'workload': 'BASELINE configs[2]: batch=%d streams/GPU x %d frames/call, %s mask GEMMs on MFMA + fp32 FFT'
                        % (B, T, args.precision),
            'streams_per_gpu': B, 'frames_per_call': T, 'global_streams': B * world,
            'parallelism': 'streams sharded over %d GPU(s), no data-path collective' % world,
        },
        'real_time_factor': round(elapsed_max / (args.steps * T * 256 / 16000.0) / B, 9),
        'frames_per_sec_per_gpu': round(value / world, 1),
        'roofline': roofline,
        'stages': stages,
        'cpu_baseline': cpu,
    }
    print(json.dumps(line))


if __name__ == '__main__':
    main()
