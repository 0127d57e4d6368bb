#!/usr/bin/env python3
"""
bench.py -- throughput of the pv_koala_process hot path on MI355X (BASELINE.json metric: 16 kHz frames/sec/GPU at
batch 4096, configs[2]: bf16 mask GEMMs on MFMA + fp32 FFT).

One "step" = one pv_koala_batch_process_chunk call: 4096 streams x 64 frames (1.024 s of audio per stream) per GPU,
int16 PCM already resident in HBM, enhanced PCM left in HBM.

Multi-GPU: one process per GPU, each rank owns its own 4096 streams (weak scaling, no data-path
collective); the only collective is the final RCCL all-reduce of {frames, max elapsed}.  `python bench.py --gpus N`
launches the N ranks ITSELF (one child per GPU, LOCAL_RANK = GPU index, rendezvous on 127.0.0.1); started by
torchrun / torch.distributed.run it uses the ranks it was given.  `n_gpus` in the output is the world size RCCL saw,
and a WORLD_SIZE that differs from --gpus is an error (exit status 2), never a silent single-GPU run.

Prints ONE JSON line (rank 0).  Besides the driver's fields it carries
  roofline      dominant kernel (by device time) priced against its bound: algorithmic work per launch / mean launch
                duration measured with HIP events on the engine's stream in a second, identical pass
  stages        the same for every kernel class (HBM-bound stages also against the bytes they were measured to move)
  cpu_baseline  the CPU oracle (a "port": plain-C restatement, OpenMP over stream blocks) on a bounded sample of the
                same workload on this box's host cores, in a child process bound one thread per physical core
                (`cores` = threads used; `noisy` when the box never went quiet) -- a reported baseline, not the target
  extra         the other BASELINE operating points, timed in the same run (N = 1 only): configs[1] (256 streams,
                fp32), configs[4] (one stream, one frame per pv_koala_process call: p50/p99), one frame per call at
                4096 streams, the host-pointer (PCIe-inclusive) paths -- synchronous calls on pageable and page-locked
                buffers, asynchronous calls rotating over three page-locked buffer pairs -- and the |GPU - oracle|
                histogram of the timed batch's first call
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3}  # dense peaks, no sparsity

H = 271
G3 = 813
HEADS = (1, 5, 40, 257)
# algorithmic work per stream-frame (unpadded dims).  STFT stages: the dataflow of this engine since round 2 (DESIGN.md
# section 6; SURVEY.md 8d priced a dataflow that stored the spectrum: 4 620 / 5 644 B): no stored spectrum in multi-frame calls
# (the synthesis kernel rebuilds it from the PCM), operand-typed features, history and overlap-add tail stay on chip inside a call
BYTES_ANALYSIS = {'bf16': 512 + 257 * 2, 'fp32': 512 + 257 * 4}      # PCM in, features out
BYTES_SYNTHESIS = 512 + 257 * 4 + 512                                # PCM in, fp32 mask in, PCM out
MAC_GEMM_IN = (271 + 272 + 276 + 311 + 4 * 271) * G3      # 8 input-side GEMMs (W_ih)
MAC_GRU = 8 * H * G3                                      # 8 recurrent GEMMs (W_hh)
MAC_HEAD = 257 * H + H * sum(HEADS)                       # front-end + 4 heads
# bf16 configuration since round 4: the (linear) front-end is folded into the four stage-input GEMMs, whose embedding rows (271)
# become feature rows (257): no front-end launch, 3 599 151 MAC per stream-frame instead of 3 714 326; the two
# narrow heads of stages 0 and 1 are computed inside their layer-B recurrent launches
MAC_GEMM_IN_FOLDED = (257 + 258 + 262 + 297 + 4 * 271) * G3
MAC_HEAD_FOLDED = H * (HEADS[2] + HEADS[3])  # (the two narrow heads, 271 x 6 MAC, ride in their recurrent launches: counted there)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--prime-seconds', type=float, default=0.5,
                    help='untimed run of the same step before the warmup steps: the MI355X takes a few hundred ms of load to '
                         'leave its idle clocks (sclk ~100 MHz), and the workload then runs against the 1400 W board power cap')
    ap.add_argument('--sustain-seconds', type=float, default=4.0,
                    help='untimed stretch of the same step after the timed region during which the sustained sclk / board power '
                         'are sampled (0: skip)')
    ap.add_argument('--streams', type=int, default=4096, help='streams per GPU')
    ap.add_argument('--frames', type=int, default=64, help='frames per stream per call (64 = 1.02 s of audio)')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-leg-child', action='store_true', help='internal: the cpu_baseline leg in its own (core-bound) process')
    ap.add_argument('--no-extra', action='store_true', help='skip the other operating points (profiling runs)')
    ap.add_argument('--no-live-traffic', action='store_true', help='do not run the two rocprofv3 --pmc passes that measure HBM bytes '
                    'per launch (then the last committed profiles/*_pmc.json is quoted, marked as not measured in this run)')
    ap.add_argument('--library', default=None, help='alternative libpv_koala.so (developer A/B runs)')
    ap.add_argument('--dist-backend', default='nccl', help='nccl (= RCCL over xGMI) for real runs; gloo only to test the '
                    'multi-rank code path on a single-GPU box together with KOALA_BENCH_SHARE_GPU=1')
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start one child per GPU and relay rank 0's line."""
    import koala_amd
    koala_amd.build_native()  # once, before the ranks exist
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    deadline = time.time() + 3600
    alive = list(procs)
    while alive and time.time() < deadline:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:  # one rank failed: the others would wait in a collective forever
                    q.terminate()
        time.sleep(0.05)
    for p in alive:
        p.kill()
        rc = rc or 124
    return rc


class BoardSampler(object):
    """sclk and board power of one GPU while a stretch of the workload runs: `rocm-smi --showclocks --showpower` called from a
    host thread (each call takes a few hundred ms; nothing is launched on the GPU).  amdgpu's sysfs files (pp_dpm_sclk,
    hwmon power1_*) were tried first and do not follow the load on this platform (2.39 GHz / 0.3 kW while rocm-smi and the
    throughput say 1.98 GHz / 1.39 kW).  Every field is None where rocm-smi is missing."""
    PEAK_SCLK_MHZ = 2400.0  # MI355X peak engine clock: what the 2.5 PFLOP/s dense bf16 figure is quoted at

    def __init__(self, index):
        import threading
        self.index = index
        self.samples = []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self.t0 = time.perf_counter()

    def _read(self):
        import re
        mhz, watts = None, None
        try:
            out = subprocess.run(['rocm-smi', '-d', str(self.index), '--showclocks', '--showpower'], capture_output=True,
                                 text=True, timeout=10).stdout
            m = re.search(r'sclk clock level[^(]*\((\d+)Mhz\)', out)
            if m:
                mhz = float(m.group(1))
            m = re.search(r'Power \(W\):\s*([0-9.]+)', out)
            if m:
                watts = float(m.group(1))
        except Exception:
            pass
        return mhz, watts

    def _run(self):
        while not self._stop.is_set():
            t = time.perf_counter() - self.t0
            self.samples.append(self._read() + (t,))
            self._stop.wait(0.1)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._thread.join()

    def summary(self, after_seconds=0.0):
        """over the samples STARTED later than `after_seconds` (the clock settles over a second or two of load)"""
        import statistics
        mhz = [m for m, _, t in self.samples if m and t >= after_seconds]
        w = [p for _, p, t in self.samples if p and t >= after_seconds]
        return {'sclk_MHz_median': statistics.median(mhz) if mhz else None, 'sclk_MHz_min': min(mhz) if mhz else None,
                'board_power_W_mean': round(sum(w) / len(w), 1) if w else None, 'samples': len(mhz),
                'peak_sclk_MHz': self.PEAK_SCLK_MHZ, 'source': 'rocm-smi --showclocks --showpower'}


PMC_CLASS = {'analysis_kernel': 'analysis', 'gemm_ws2_kernel<0, 4>': 'gemm_input', 'gru_resident8_kernel': 'gru_recurrent',
             'synthesis_kernel': 'synthesis', 'gemm_wsr_kernel<2, 2>': 'gemm_head'}


def live_traffic(args):
    """HBM bytes per launch of each class's main kernel, measured NOW: two rocprofv3 passes of this same command (a few steps, no
    extras), one hardware counter each -- FETCH_SIZE, WRITE_SIZE in KiB, kernel-trace only, as MI355X_MICROARCH.md's
    HBM section prescribes (separate --pmc passes; FETCH_SIZE doubled: on gfx950 this rocprofv3 reports half of a wide
    streaming read).  Returns {class: bytes} or None when rocprofv3 is missing or a pass fails (the caller then quotes the
    committed profile)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if not shutil.which('rocprofv3'):
        return None
    per = {}
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        with tempfile.TemporaryDirectory(dir='/tmp') as d:
            cmd = ['rocprofv3', '--kernel-trace', '--pmc', ctr, '-d', d, '-o', 'p', '--', sys.executable, os.path.abspath(__file__),
                   '--steps', '3', '--warmup', '1', '--prime-seconds', '0', '--sustain-seconds', '0', '--no-cpu-baseline', '--no-extra',
                   '--no-live-traffic', '--streams', str(args.streams), '--frames', str(args.frames), '--precision', args.precision]
            if args.library:
                cmd += ['--library', args.library]
            env = dict(os.environ, TMPDIR='/tmp')
            for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
                env.pop(k, None)
            try:
                proc = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                        start_new_session=True)
                try:
                    rc = proc.wait(timeout=90)
                except subprocess.TimeoutExpired:  # the profiler and the bench under it, not only the wrapper
                    import signal
                    os.killpg(proc.pid, signal.SIGKILL)
                    proc.wait()
                    return None
                dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith('.db')]
                if rc != 0 or not dbs:
                    return None
                rows = sqlite3.connect(dbs[0]).execute(
                    'select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name',
                    (ctr,)).fetchall()
            except Exception:
                return None
            for name, total, n in rows:
                for frag, cls in PMC_CLASS.items():
                    if frag in name and n:
                        per.setdefault(cls, {})[ctr] = total / n
    out = {}
    for cls, v in per.items():
        if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            out[cls] = int(2 * v['FETCH_SIZE'] * 1024) + int(v['WRITE_SIZE'] * 1024)
    return out or None


def time_steps(fn, sync, steps, warmup):
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return time.perf_counter() - t0


def physical_cores():
    """distinct (package, core) pairs of the CPUs this process may run on; falls back to the logical count"""
    try:
        seen = set()
        for c in os.sched_getaffinity(0):
            base = '/sys/devices/system/cpu/cpu%d/topology/' % c
            seen.add((open(base + 'physical_package_id').read().strip(), open(base + 'core_id').read().strip()))
        return max(1, len(seen))
    except Exception:
        return os.cpu_count() or 1


def cpu_leg_child(T, distinct):
    """The cpu_baseline leg proper, in its own process (started by main() with the OpenMP binding in its environment): the oracle
    ("port") on a bounded sample of the bench workload -- calibrated for ~15 s -- plus one thread for scale.  Prints one JSON line."""
    import numpy as np
    from koala_amd import params
    from koala_amd.workload import synth_streams
    from oracle import oracle
    model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
    native = oracle.build_native()
    threads = int(os.environ.get('OMP_NUM_THREADS', '0')) or (os.cpu_count() or 1)
    base = synth_streams(distinct, T, seed=1234)
    xs = np.ascontiguousarray(np.tile(base, ((1024 + distinct - 1) // distinct, 1))[:1024, :8 * 256])
    o = oracle.Oracle(model, 1024, oracle.PREC_FP32, library=native)
    o.process(xs, threads)  # (first touch, thread start-up)
    c0 = time.perf_counter()
    o.process(xs, threads)
    rate = 1024 * 8 / (time.perf_counter() - c0)
    ns = int(min(32768, max(64, (rate * 15) // (T * 64) * 64)))
    o = oracle.Oracle(model, ns, oracle.PREC_FP32, library=native)
    xs = np.ascontiguousarray(np.tile(base, ((ns + distinct - 1) // distinct, 1))[:ns])
    c0 = time.perf_counter()
    o.process(xs, threads)
    dt = time.perf_counter() - c0
    block = oracle.block_size()
    n1 = 64  # one thread, for scale: the multi-thread figure is a weak baseline (shared weights, 64-stream blocks per thread)
    o1 = oracle.Oracle(model, n1, oracle.PREC_FP32, library=native)
    x1 = np.ascontiguousarray(np.tile(base, ((n1 + distinct - 1) // distinct, 1))[:n1, :16 * 256])
    c1 = time.perf_counter()
    o1.process(x1, 1)
    rate1 = n1 * 16 / (time.perf_counter() - c1)
    print(json.dumps({'frames_per_s': round(ns * T / dt, 1), 'threads': threads, 'streams': ns, 'seconds': round(dt, 1), 'block': block,
                      'one_thread_frames_per_s': round(rate1, 1),
                      'build': '-O3 -march=native (this host)' if native.endswith('_native.so') else '-O3 -march=x86-64-v3'}))


def host_busy_fraction(seconds):
    """share of all host cores' time that was not idle over the next `seconds` (/proc/stat, first line)"""
    def read():
        v = [int(x) for x in open('/proc/stat').readline().split()[1:]]
        return sum(v), v[3] + (v[4] if len(v) > 4 else 0)  # total, idle + iowait
    try:
        t0, i0 = read()
        time.sleep(seconds)
        t1, i1 = read()
        return 1.0 - (i1 - i0) / max(1, t1 - t0)
    except Exception:
        return 0.0


def machine_state():
    """What resources/scripts/machine-state.sh of the reference logs around its perf runs: load and memory."""
    st = {}
    try:
        st['loadavg_1m'] = float(open('/proc/loadavg').read().split()[0])
        mem = dict((l.split(':')[0], int(l.split()[1])) for l in open('/proc/meminfo') if ':' in l)
        st['mem_used_pct'] = round(100.0 * (1.0 - mem['MemAvailable'] / float(mem['MemTotal'])), 1)
    except Exception:
        pass
    return st


def extra_points(args, torch, np, koala_amd, model, kb, x, dx, dy, base, local_rank):
    """The other BASELINE operating points, on the same box in the same run (rank 0 of a single-GPU run)."""
    from koala_amd import params
    from koala_amd.workload import synth_streams
    B, T = args.streams, args.frames
    out = {}
    sync = torch.cuda.synchronize
    dev = 'gpu:%d' % local_rank

    # -- one frame per call at the bench batch (the reference's calling convention, batched)
    d1 = dx[:, :256].contiguous()
    o1 = torch.empty_like(d1)
    dt = time_steps(lambda: kb.process_device(1, d1.data_ptr(), o1.data_ptr()), sync, 300, 30)
    out['streaming_T1'] = {'workload': '%d streams x 1 frame per call, %s, device-resident' % (B, args.precision),
                           'frames_per_s': round(B * 300 / dt, 1), 'ms_per_call': round(dt / 300 * 1e3, 4)}

    # -- the same calling convention with twice the streams on the GPU (two m-tiles' worth of quads per CU in turn: the launch
    # and weight-prologue costs of a frame step are shared by more streams)
    k8 = koala_amd.create_batch('bench', 2 * B, 1, args.precision, model_path=model, device=dev, library_path=args.library)
    k8.set_stream(torch.cuda.current_stream().cuda_stream)
    d8 = torch.cat([d1, d1])
    o8 = torch.empty_like(d8)
    dt = time_steps(lambda: k8.process_device(1, d8.data_ptr(), o8.data_ptr()), sync, 300, 30)
    k8.delete()
    out['streaming_T1_2x_streams'] = {'workload': '%d streams x 1 frame per call, %s, device-resident' % (2 * B, args.precision),
                                      'frames_per_s': round(2 * B * 300 / dt, 1), 'ms_per_call': round(dt / 300 * 1e3, 4)}

    # -- host-pointer (PCIe-inclusive) path at the bench size: pageable numpy arrays, then page-locked ones
    y = np.empty_like(x)
    dt = time_steps(lambda: kb.process_into(x, y), lambda: None, 4, 1)
    out['host_pageable'] = {'workload': '%d x %d frames per call, caller buffers in pageable host memory' % (B, T),
                            'frames_per_s': round(B * T * 4 / dt, 1)}
    px, py = kb.alloc_host(T), kb.alloc_host(T)
    px[:] = x
    dt = time_steps(lambda: kb.process_into(px, py), lambda: None, 4, 1)
    out['host_pinned'] = {'workload': '%d x %d frames per call, caller buffers page-locked (pv_koala_batch_host_alloc), synchronous calls'
                                      % (B, T), 'frames_per_s': round(B * T * 4 / dt, 1)}
    # ... and the throughput-oriented caller: three page-locked buffer pairs, pv_koala_batch_process_chunk_async -- one call's copies run
    # under its neighbours' kernels (a synchronous call cannot hide its first copy-in and last copy-out).  (engine on its own stream
    # here: the asynchronous path chains three streams by events)
    try:
        ka = koala_amd.create_batch('bench', B, T, args.precision, model_path=model, device=dev, library_path=args.library)
        pairs = [(ka.alloc_host(T), ka.alloc_host(T)) for _ in range(3)]
        for a, _ in pairs:
            a[:] = x
        n_async = [0]

        def async_step():
            a, b = pairs[n_async[0] % 3]
            ka.process_async(a, b)
            n_async[0] += 1
        dt = time_steps(async_step, ka.synchronize, 36, 6)  # (the pipeline's fill and drain -- one exposed copy each way -- spread over 36 calls)
        out['host_pinned_async'] = {'workload': '%d x %d frames per call, three page-locked buffer pairs in rotation, '
                                                'pv_koala_batch_process_chunk_async (three calls in flight)' % (B, T),
                                    'frames_per_s': round(B * T * 36 / dt, 1)}
        ka.delete()
    except Exception as e:  # (a library without the entry point: developer A/B runs against older builds)
        out['host_pinned_async'] = {'error': str(e)[:200]}

    # -- KNS-v1.1: the same batch with the reference model file's front-end topology (a linear layer over FIVE stacked feature
    # frames, koala_params.pv record [1285, 271]: + 4 x 257 x 271 = 278 588 MAC per stream-frame, 3 992 914 in all)
    m5 = params.ensure_params(os.path.join(ROOT, 'build', 'random5_1234.kns'), 'random5', 1234)
    k5 = koala_amd.create_batch('bench', B, T, args.precision, model_path=m5, device=dev, library_path=args.library)
    k5.set_stream(torch.cuda.current_stream().cuda_stream)
    dt = time_steps(lambda: k5.process_device(T, dx.data_ptr(), dy.data_ptr()), sync, 100, 10)
    k5.profile_enable(True)
    for _ in range(10):
        k5.process_device(T, dx.data_ptr(), dy.data_ptr())
    p5 = k5.profile_read()
    k5.profile_enable(False)
    dt1 = time_steps(lambda: k5.process_device(1, d1.data_ptr(), o1.data_ptr()), sync, 300, 30)  # ... and one frame per call
    k5.delete()
    out['front_taps5'] = {'workload': 'KNS-v1.1 (five-frame front-end, random weights), %d streams x %d frames per call, %s' % (B, T, args.precision),
                          'frames_per_s': round(B * T * 100 / dt, 1), 'ms_per_call': round(dt / 100 * 1e3, 4),
                          'mac_per_stream_frame': MAC_GEMM_IN + MAC_GRU + MAC_HEAD + 4 * 257 * H,
                          'gemm_head_class_ms_per_call': round(p5['gemm_head']['ms'] / 10, 4),
                          'one_frame_per_call_frames_per_s': round(B * 300 / dt1, 1), 'one_frame_per_call_ms': round(dt1 / 300 * 1e3, 4)}

    # -- BASELINE configs[1]: 256 streams, fp32 mask network
    for T1 in (32, 1):
        k1 = koala_amd.create_batch('bench', 256, T1, 'fp32', model_path=model, device=dev, library_path=args.library)
        k1.set_stream(torch.cuda.current_stream().cuda_stream)
        a = torch.from_numpy(np.ascontiguousarray(np.tile(base, (4, 1))[:256, :T1 * 256])).cuda()
        b = torch.empty_like(a)
        n = 60 if T1 > 1 else 300
        dt = time_steps(lambda: k1.process_device(T1, a.data_ptr(), b.data_ptr()), sync, n, 10)
        fps1 = 256 * T1 * n / dt
        tflops = fps1 * 2.0 * (MAC_GEMM_IN + MAC_GRU + MAC_HEAD) / 1e12
        out['config1_fp32_b256_T%d' % T1] = {
            'workload': 'BASELINE configs[1]: 256 streams x %d frame(s) per call, fp32 mask net, device-resident' % T1,
            'frames_per_s': round(fps1, 1), 'ms_per_call': round(dt / n * 1e3, 4),
            # the whole mask network against the fp32 matrix peak: the fp32 configuration is the bit-exact correctness path (fp32 =
            # oracle, value for value); T = 32 runs as a wavefront over (layer, frame), T = 1 as one launch per layer (DESIGN.md
            # section 6, profiles/r04_wavefront.txt)
            'roofline': {'bound': 'mfma', 'achieved': round(tflops, 2), 'peak': MFMA_PEAK_TFLOPS['fp32'], 'unit': 'TFLOP/s',
                         'frac': round(tflops / MFMA_PEAK_TFLOPS['fp32'], 4), 'flop_per_stream_frame': 2 * (MAC_GEMM_IN + MAC_GRU + MAC_HEAD)}}
        k1.delete()

    # -- the +-1 LSB configuration AT THE HEADLINE BATCH: 4096 streams x 64 frames per call with the fp32 mask network -- the engine that is the
    # oracle bit for bit (north_star: "output within +-1 LSB"; the bf16 headline is specified to a tolerance: <= 3 LSB, 99.998 % within 1)
    if args.precision == 'bf16' and not args.no_cpu_baseline:
        try:
            from oracle import oracle  # (the checker of this leg's output, not the thing timed)
            kf = koala_amd.create_batch('bench', B, T, 'fp32', model_path=model, device=dev, library_path=args.library)
            kf.set_stream(torch.cuda.current_stream().cuda_stream)
            yf = torch.empty_like(dx)
            kf.process_device(T, dx.data_ptr(), yf.data_ptr())
            sync()
            nchk = min(64, base.shape[0])
            want = oracle.Oracle(model, nchk, oracle.PREC_FP32).process(np.ascontiguousarray(x[:nchk]))
            dmax = int(np.abs(yf[:nchk].cpu().numpy().astype(np.int64) - want).max())
            nf = 8
            dt = time_steps(lambda: kf.process_device(T, dx.data_ptr(), yf.data_ptr()), sync, nf, 2)
            kf.delete()
            fpsf = B * T * nf / dt
            tflops = fpsf * 2.0 * (MAC_GEMM_IN + MAC_GRU + MAC_HEAD) / 1e12
            out['fp32_b4096_T64'] = {
                'workload': '%d streams x %d frames per call, fp32 mask net + fp32 FFT, device-resident' % (B, T),
                'frames_per_s': round(fpsf, 1), 'ms_per_call': round(dt / nf * 1e3, 3),
                'roofline': {'bound': 'mfma', 'achieved': round(tflops, 2), 'peak': MFMA_PEAK_TFLOPS['fp32'], 'unit': 'TFLOP/s',
                             'frac': round(tflops / MFMA_PEAK_TFLOPS['fp32'], 4), 'flop_per_stream_frame': 2 * (MAC_GEMM_IN + MAC_GRU + MAC_HEAD)},
                'parity': {'streams': nchk, 'frames': T, 'oracle': 'kns_oracle.c, fp32', 'max_lsb': dmax, 'bar_max_lsb': 0, 'pass': dmax == 0}}
        except Exception as e:  # noqa: BLE001
            out['fp32_b4096_T64'] = {'error': repr(e)[:300]}

    # -- mid-size batches: the layers of a call as a wavefront over two sub-chunks of frames (round 6; kns_engine.cpp, kRoutePipelined)
    for Bm in (1024, 2048):
        try:
            km = koala_amd.create_batch('bench', Bm, T, args.precision, model_path=model, device=dev, library_path=args.library)
            km.set_stream(torch.cuda.current_stream().cuda_stream)
            xm = dx[:Bm].contiguous()
            ym = torch.empty_like(xm)
            dt = time_steps(lambda: km.process_device(T, xm.data_ptr(), ym.data_ptr()), sync, 60, 10)
            km.set_stream(0)
            km.delete()
            out['mid_b%d_T%d' % (Bm, T)] = {'workload': '%d streams x %d frames per call, %s, device-resident' % (Bm, T, args.precision),
                                            'frames_per_s': round(Bm * T * 60 / dt, 1), 'ms_per_call': round(dt / 60 * 1e3, 4)}
        except Exception as e:  # noqa: BLE001
            out['mid_b%d_T%d' % (Bm, T)] = {'error': repr(e)[:200]}

    # -- the many-files mode's shape (koala_amd/demo/koala_demo_file.py: a few files x 32 frames per call): 16 streams, both precisions
    for prec in ('bf16', 'fp32'):
        k1 = koala_amd.create_batch('bench', 16, 32, prec, model_path=model, device=dev, library_path=args.library)
        k1.set_stream(torch.cuda.current_stream().cuda_stream)
        a = torch.from_numpy(np.ascontiguousarray(base[:16, :32 * 256])).cuda()
        b = torch.empty_like(a)
        dt = time_steps(lambda: k1.process_device(32, a.data_ptr(), b.data_ptr()), sync, 100, 10)
        out['files_b16_T32_%s' % prec] = {'workload': '16 streams x 32 frames per call, %s, device-resident (wavefront route)' % prec,
                                          'frames_per_s': round(16 * 32 * 100 / dt, 1), 'ms_per_call': round(dt / 100 * 1e3, 4),
                                          'real_time_factor': round(dt / 100 / (16 * 32 * 0.016), 6)}
        k1.delete()

    # -- BASELINE configs[4]: one stream, one frame per pv_koala_process call (hipGraph replay, host buffers)
    frame = np.ascontiguousarray(base[0, :256])
    for prec in ('fp32', 'bf16'):
        os.environ['KOALA_AMD_PRECISION'] = prec
        k = koala_amd.create('bench', model_path=model, device=dev, library_path=args.library)
        for _ in range(200):
            k.process(frame)
        lat = np.empty(3000)
        for i in range(3000):
            t0 = time.perf_counter()
            k.process(frame)
            lat[i] = time.perf_counter() - t0
        k.delete()
        out['config4_b1_%s' % prec] = {
            'workload': 'BASELINE configs[4]: 1 stream, 1 frame per pv_koala_process call through the Python binding '
                        '(numpy frame in, list out), %s' % prec,
            'p50_us': round(float(np.percentile(lat, 50)) * 1e6, 1), 'p99_us': round(float(np.percentile(lat, 99)) * 1e6, 1),
            'frames_per_s': round(1.0 / float(np.mean(lat)), 1),
            'real_time_factor': round(float(np.mean(lat)) / 0.016, 5)}
    os.environ.pop('KOALA_AMD_PRECISION', None)

    # -- BASELINE configs[0], the one number the reference publishes a bound for: seconds to push the 365 full frames of
    # resources/audio_samples/test.wav through Koala.process() (binding/python/test_koala_perf.py:42-58: mean of the timed
    # iterations after warm-up; CI ceiling 0.8 s on cpu:1, .github/workflows/python-perf.yml:45-53).  Here on gpu:N with the
    # default model: one pv_koala_process call (hipGraph replay) per frame.
    try:
        import wave
        with wave.open(os.path.join(ROOT, 'tests', 'golden', 'test.wav'), 'rb') as f:
            wav = np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16)
        nf = len(wav) // 256
        frames = [wav[i * 256:(i + 1) * 256] for i in range(nf)]
        res = {}
        for prec in ('fp32', 'bf16'):
            os.environ['KOALA_AMD_PRECISION'] = prec
            k = koala_amd.create('bench', device=dev, library_path=args.library)
            secs = []
            for it in range(21):
                t0 = time.perf_counter()
                for fr in frames:
                    k.process(fr)
                if it:  # iteration 0 is the warm-up
                    secs.append(time.perf_counter() - t0)
            k.delete()
            res[prec] = round(float(np.mean(secs)), 5)
        os.environ.pop('KOALA_AMD_PRECISION', None)
        out['config0_testwav_loop'] = {
            'workload': 'BASELINE configs[0] loop: %d frames of tests/golden/test.wav through Koala.process(), one frame per '
                        'call, mean of 20 iterations after 1 warm-up, device %s, default model' % (nf, dev),
            'seconds_fp32': res['fp32'], 'seconds_bf16': res['bf16'], 'reference_ci_ceiling_seconds_cpu1': 0.8,
            'frames_per_s_fp32': round(nf / res['fp32'], 1), 'real_time_factor_fp32': round(res['fp32'] / (nf * 0.016), 5)}
    except Exception as e:  # the fixture travels with the repository; never fail the bench line over an extra
        out['config0_testwav_loop'] = {'error': repr(e)}
    return out


def main():
    args = parse_args()
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world_env = int(env_world or '1')
    if world_env != args.gpus:
        print('bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a number for a different job than the one asked for'
              % (args.gpus, world_env), file=sys.stderr)
        sys.exit(2)

    import numpy as np
    import torch

    import koala_amd
    from koala_amd import params
    from koala_amd.sharding import aggregate_throughput, shard_range
    from koala_amd.workload import synth_streams

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    share_gpu = bool(os.environ.get('KOALA_BENCH_SHARE_GPU'))
    if share_gpu:
        local_rank = 0  # test mode: every rank drives GPU 0
    if not share_gpu and torch.cuda.device_count() < (local_rank + 1):
        print('bench.py: rank %d needs GPU %d but only %d visible' % (rank, local_rank, torch.cuda.device_count()),
              file=sys.stderr)
        sys.exit(2)
    world = 1
    collective = None
    if world_env > 1 or 'RANK' in os.environ:  # started by a launcher (torchrun, or launch_ranks above): even one rank joins a group
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.dist_backend)
        world = dist.get_world_size()  # what the communicator was actually built with
        collective = dist.get_backend()
    else:
        torch.cuda.set_device(local_rank)

    # one rank per GPU on one node: every rank keeps to its own slice of the host cores (its Python thread, the engine's
    # pageable-path copy threads and RCCL's proxy thread then never migrate onto a sibling's cores).  KOALA_BENCH_NO_AFFINITY=1
    # leaves the scheduler alone.
    cores_per_rank = None
    if world > 1 and not os.environ.get('KOALA_BENCH_NO_AFFINITY') and hasattr(os, 'sched_setaffinity'):
        try:
            cores = sorted(os.sched_getaffinity(0))
            local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
            lr = int(os.environ.get('LOCAL_RANK', '0'))
            if len(cores) >= local_world:
                mine = cores[lr * len(cores) // local_world:(lr + 1) * len(cores) // local_world]
                os.sched_setaffinity(0, mine)
                cores_per_rank = len(mine)
        except OSError:
            pass

    koala_amd.build_native()
    model = params.ensure_params(os.path.join(ROOT, 'build', 'random_1234.kns'), 'random', 1234)
    B, T = args.streams, args.frames
    first, _ = shard_range(B * world, rank, world)

    # synthetic input: `distinct` seeded streams tiled over the batch (generation cost, not data-path cost)
    distinct = min(B, 1024 if (rank == 0 and world == 1 and not args.no_cpu_baseline) else 64)
    base = synth_streams(distinct, T, seed=1234, first_stream=(first % 4096))
    x = np.ascontiguousarray(np.tile(base, ((B + distinct - 1) // distinct, 1))[:B])
    dx = torch.from_numpy(x).cuda()
    dy = torch.empty_like(dx)

    kb = koala_amd.create_batch('bench', B, T, args.precision, model_path=model, device='gpu:%d' % local_rank,
                               library_path=args.library)
    # torch's default stream is the NULL stream, which this ABI reads as "the handle's own stream": the bench runs on a stream of its
    # own instead, so that the engine's kernels and the HIP events between the steps below sit on ONE stream
    bench_stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    torch.cuda.set_stream(bench_stream)
    kb.set_stream(bench_stream.cuda_stream)

    def step():
        kb.process_device(T, dx.data_ptr(), dy.data_ptr())

    def barrier():
        if collective:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # the first call after creation is also the parity sample (checked against the oracle further down)
    step()
    torch.cuda.synchronize()
    first_call = dy[:distinct].cpu().numpy() if (rank == 0 and world == 1) else None
    # ... and its masks (the mask head's output buffer, read back through the debug tap; only the parity leg uses them)
    first_mask = kb.debug_read('mask', T)[:, :distinct] if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

    # ---- priming, untimed (see --prime-seconds): the board's clock / power state follows the load with a lag of tens of milliseconds,
    # and a burst that starts from an idle or half-ramped board runs its first steps 5-10 % slower than steady state (round 5: the
    # driver's 20-step run measured 90.2 M frames/s while a 4-s stretch of the same step in the same process ran at 98.4 M).  So the
    # step is run BACK TO BACK -- no host synchronisation between steps: that idle gap is what lets the clock sag -- in groups of eight
    # with a HIP event between steps, until three consecutive steps agree within 1 % (and at least --prime-seconds have passed, at
    # most 6 s); --warmup and --steps follow as given.  The board's sclk / power at the end of priming go on the line.
    def timed_group(n):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            step()
            ev[i + 1].record()
        return ev

    def group_ms(ev):
        return [ev[i].elapsed_time(ev[i + 1]) for i in range(len(ev) - 1)]

    prime = {'steps': 0, 'converged': False}
    board_before = None
    t_prime = time.perf_counter()
    if args.prime_seconds > 0:
        sampler = BoardSampler(local_rank) if rank == 0 else None
        if sampler:
            sampler.__enter__()
        pending, hist = timed_group(8), []
        while True:
            nxt = timed_group(8)  # enqueued before the previous group is waited for: the GPU never idles
            pending[-1].synchronize()
            hist = (hist + group_ms(pending))[-8:]
            prime['steps'] += 8
            pending = nxt
            dt = time.perf_counter() - t_prime
            last3 = hist[-3:]
            prime['converged'] = max(last3) <= 1.01 * min(last3)
            if (prime['converged'] and dt >= args.prime_seconds) or dt > 6.0:
                break
        if sampler:
            # (the sampler's last rocm-smi call takes a few hundred ms to return: the step keeps running meanwhile -- an idle board here
            # is exactly what priming is meant to avoid)
            sampler._stop.set()
            while sampler._thread.is_alive():
                nxt = timed_group(8)
                pending[-1].synchronize()
                hist = (hist + group_ms(pending))[-8:]
                prime['steps'] += 8
                pending = nxt
            sampler._thread.join()
        prime['last_steps_ms'] = [round(v, 4) for v in hist[-3:]]
        if sampler:
            sm = sampler.summary(after_seconds=max(0.0, (time.perf_counter() - t_prime) - 1.0))
            board_before = {'sclk_MHz': sm['sclk_MHz_median'], 'board_power_W': sm['board_power_W_mean'], 'samples': sm['samples'],
                            'when': 'last second of priming (rocm-smi from a host thread while the step runs)'}
        prime['steps'] += 8  # the group still in flight: it runs straight into the warm-up steps below
    prime['seconds'] = round(time.perf_counter() - t_prime, 2)
    if collective and args.prime_seconds > 0:
        # several ranks: priming ends at a different moment on every rank, and waiting for the slowest one at the barrier below would
        # leave the other boards idle for tenths of a second -- the sag priming exists to avoid.  So the ranks are aligned HERE, run a
        # fixed stretch of the step back to back again (~60 ms), and meet the barrier in front of the timed region within a millisecond
        barrier()
        for _ in range(24):
            step()
        prime['steps'] += 24
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    timed_ev = timed_group(args.steps) if args.steps <= 256 else None  # (an event record costs ~2 us against a ~2.6 ms step)
    if timed_ev is None:
        for _ in range(args.steps):
            step()
    barrier()
    elapsed = time.perf_counter() - t0
    per_step_ms = [round(v, 4) for v in group_ms(timed_ev)] if timed_ev else None
    # A further untimed stretch of the same step gives the figures the MFMA fractions are also quoted against
    # (`frac_at_sustained_sclk`: the board runs this workload AT its 1.4 kW cap, ~2.0 GHz instead of 2.4) and the throughput over
    # four seconds; since round 6's priming the timed region starts from that same state.
    sustained = None
    if rank == 0 and args.sustain_seconds > 0:
        with BoardSampler(local_rank) as sb:
            n_s, t_s = 0, time.perf_counter()
            while time.perf_counter() - t_s < args.sustain_seconds:
                for _ in range(50):
                    step()
                torch.cuda.synchronize()
                n_s += 50
            dt_s = time.perf_counter() - t_s
        sustained = sb.summary(after_seconds=args.sustain_seconds / 2.0)
        sustained['seconds'] = round(dt_s, 2)
        sustained['frames_per_s'] = round(B * T * n_s / dt_s, 1)
    frames_total, elapsed_max = aggregate_throughput(B * T * args.steps, elapsed)
    value = frames_total / elapsed_max

    # ---- per-kernel pass (same workload, HIP events around every launch on the engine's stream)
    prof_steps = max(2, min(args.steps, 20))
    kb.profile_enable(True)
    torch.cuda.synchronize()
    t_prof = time.perf_counter()
    for _ in range(prof_steps):
        step()
    prof = kb.profile_read()  # (synchronises)
    instrumented_ms_per_step = (time.perf_counter() - t_prof) / prof_steps * 1e3
    kb.profile_enable(False)
    frames_per_launch = B * T
    folded = args.precision == 'bf16'
    mac_in, mac_head, head_launches = (MAC_GEMM_IN_FOLDED, MAC_HEAD_FOLDED, 2) if folded else (MAC_GEMM_IN, MAC_HEAD, 5)
    mac_gru = MAC_GRU + (H * (HEADS[0] + HEADS[1]) if folded else 0)
    work = {
        'analysis': ('hbm', BYTES_ANALYSIS[args.precision] * frames_per_launch, 1),
        'gemm_input': ('mfma', 2.0 * mac_in / 8 * frames_per_launch, 8),
        'gru_recurrent': ('mfma', 2.0 * mac_gru / 8 * frames_per_launch, 8),
        'gemm_head': ('mfma', 2.0 * mac_head / head_launches * frames_per_launch, head_launches),
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
            per_launch += 2.0 * mac_in * frames_per_launch / actual
        ms = prof[name]['ms'] / n
        if bound == 'hbm':
            achieved, peak, unit = per_launch / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, 'GB/s'
        else:
            achieved, peak, unit = per_launch / (ms * 1e-3) / 1e12, MFMA_PEAK_TFLOPS[args.precision], 'TFLOP/s'
        stages[name] = {'bound': bound, 'achieved': round(achieved, 3), 'peak': peak, 'unit': unit,
                        'frac': round(achieved / peak, 5), 'avg_launch_ms': round(ms, 5),
                        'launches_per_step': round(actual, 2) if actual != int(actual) else int(actual),
                        'share_of_device_time': round(ms * actual / dev_ms, 4) if dev_ms else None}
        clk = (sustained or {}).get('sclk_MHz_median')
        if bound == 'mfma' and clk:
            # the same fraction against the matrix peak AT THE CLOCK THE BOARD SUSTAINS under its power cap with this workload
            stages[name]['frac_at_sustained_sclk'] = round(achieved / (peak * clk / BoardSampler.PEAK_SCLK_MHZ), 5)
        if name in ('analysis', 'synthesis'):
            # limited by VALU issue, not by HBM: `frac` prices the bytes the dataflow needs, and
            # `frac_of_peak_by_traffic` (below) the bytes the kernel was measured to move
            stages[name]['limited_by'] = 'VALU issue'

    # ---- CPU baseline + parity check of what was timed (rank 0 of a single-GPU run only).  BEFORE the rocprofv3 passes and the
    # other operating points: their host threads leave a load average of 15-30 behind, and the CPU leg wants a quiet box
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        ncores = os.cpu_count() or 1
        # the timed leg runs the oracle compiled for THIS host's instruction set (-march=native, built here and now; the parity
        # checks below use the portable build that travels with the repository: same source, same results bit for bit) and
        # starts once the box is quiet -- the 1-minute load average below 4, waited for at most 45 s (it decays with a 60 s time
        # constant: after the GPU legs' host threads it starts around 15-30)
        # The timed leg runs in a CHILD process (cpu_leg_child below): its OpenMP runtime starts from a clean environment -- one thread
        # per physical core, bound (OMP_PLACES=cores, OMP_PROC_BIND=close: a block's scratch is first touched, hence allocated, on the
        # NUMA node of the core that works on it), no torch thread pools beside it -- on the oracle compiled for THIS host's instruction
        # set (-march=native, built here and now; the parity checks below use the portable build that travels with the repository:
        # same source, same results bit for bit).  It starts once the box is quiet (below), waited for at most 45 s; a leg that had to start on a busy box
        # says so: "noisy": true.
        # (quiet = the host's cores at least 97 % idle over the last second, read from /proc/stat: the 1-minute load average this used to
        # wait for decays with a 60 s time constant and still reads 20-60 minutes after the load -- this run's own start-up -- is gone)
        t_wait = time.perf_counter()
        busy = host_busy_fraction(1.0)
        while busy >= 0.03 and time.perf_counter() - t_wait < 45.0:
            busy = host_busy_fraction(1.0)
        waited = time.perf_counter() - t_wait
        before = machine_state()
        before['waited_for_quiet_s'] = round(waited, 1)
        before['host_busy_fraction_last_second'] = round(busy, 4)
        phys = physical_cores()
        env = dict(os.environ, OMP_PLACES='cores', OMP_PROC_BIND='close', OMP_NUM_THREADS=str(phys), OMP_DYNAMIC='false')
        env.pop('KNS_ORACLE_JITTER', None)
        # (a child that crashes, times out or prints something else must not cost the whole line: the GPU legs are done by now)
        try:
            child = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-leg-child', '--frames', str(T), '--streams', str(distinct)],
                                   env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            if child.returncode != 0:
                raise RuntimeError('cpu leg exited with status %d: %s' % (child.returncode, child.stderr.decode(errors='replace')[-400:]))
            leg = json.loads(child.stdout.decode().strip().splitlines()[-1])
            cpu = {'value': leg['frames_per_s'], 'unit': 'frames/s', 'cores': leg['threads'], 'kind': 'port',
                   'host_logical_cpus': ncores, 'binding': 'OMP_PLACES=cores OMP_PROC_BIND=close, one thread per physical core, own process',
                   'noisy': bool(busy >= 0.03),
                   'one_thread_frames_per_s': leg['one_thread_frames_per_s'], 'scaling_vs_1_thread': round(leg['frames_per_s'] / leg['one_thread_frames_per_s'], 1),
                   'build': leg['build'],
                   'sample': '%d streams x %d frames of the same synthetic workload, oracle/kns_oracle.c fp32 (register-blocked '
                             'k-ascending fmaf GEMMs, OpenMP over stream blocks of %d), %.1f s'
                             % (leg['streams'], T, leg['block'], leg['seconds']),
                   'machine_state_before': before, 'machine_state_after': machine_state()}
        except Exception as e:  # noqa: BLE001
            cpu = {'value': None, 'unit': 'frames/s', 'cores': phys, 'kind': 'port', 'error': repr(e)[:600], 'machine_state_before': before}
        # parity of the timed engine's first call against the oracle run with the same rounding points, over every
        # distinct stream of the batch
        want = oracle.Oracle(model, distinct, oracle.PREC_BF16 if args.precision == 'bf16' else oracle.PREC_FP32).process(
            np.ascontiguousarray(x[:distinct])).astype(np.int64)
        d = np.abs(first_call.astype(np.int64) - want)
        hist = np.bincount(np.minimum(d.ravel(), 8), minlength=9)
        parity = {'streams': int(distinct), 'frames': T, 'oracle': 'kns_oracle.c, %s rounding points' % args.precision,
                  'max_lsb': int(d.max()), 'abs_diff_histogram_0_to_8plus': hist.tolist(),
                  'within_1_lsb': round(float((d <= 1).mean()), 6)}
        cpu['gpu_vs_oracle_max_lsb'] = parity['max_lsb']
        # north_star's criterion for the floating-point mask path: the timed engine's masks against the UNROUNDED (fp32)
        # oracle, RMS over every distinct stream x frame x bin of the batch
        _, want_mask = oracle.Oracle(model, distinct, oracle.PREC_FP32).process_with_mask(np.ascontiguousarray(x[:distinct]))
        dm = first_mask.astype(np.float64) - want_mask
        parity['mask_rms_vs_fp32_oracle'] = float('%.3e' % np.sqrt(np.mean(dm * dm)))
        parity['mask_max_abs_vs_fp32_oracle'] = float('%.3e' % np.abs(dm).max())
        parity['mask_rms_bar'] = 1e-3
        # the bars of the timed configuration: bf16 -- the spec's tolerance: at most 5 LSB against the
        # oracle with the same rounding points, >= 99.9 % of the samples within 1 LSB, mask within 1e-3 RMS of the fp32 path;
        # fp32 -- the oracle's samples.  A run that misses one prints its line and FAILS (exit status 3).
        bars = {'max_lsb': 5, 'within_1_lsb': 0.999, 'mask_rms': 1e-3} if args.precision == 'bf16' else \
               {'max_lsb': 0, 'within_1_lsb': 1.0, 'mask_rms': 1e-3}
        parity['bars'] = bars
        parity['pass'] = bool(parity['max_lsb'] <= bars['max_lsb'] and parity['within_1_lsb'] >= bars['within_1_lsb'] and
                              parity['mask_rms_vs_fp32_oracle'] < bars['mask_rms'])
    dominant = max(stages, key=lambda k: stages[k]['avg_launch_ms'] * stages[k]['launches_per_step'])
    # HBM bytes per launch come from separate rocprofv3 --pmc passes of this same command (tools/pmc_run.sh ->
    # profiles/*_pmc.json; FETCH_SIZE doubled per the gfx950 correction); they cannot be collected inside a timed run.
    # `traffic_commit` is the source revision the counters were collected on: compare it with HEAD to see staleness.
    traffic_meta = {}
    live = None
    if rank == 0 and world == 1 and not args.no_extra and not args.no_live_traffic:
        live = live_traffic(args)

    def quote(name, nbytes):
        stages[name]['traffic'] = nbytes
        if stages[name]['bound'] == 'hbm':  # against the bytes the kernel was measured to move
            gbs = nbytes / (stages[name]['avg_launch_ms'] * 1e-3) / 1e9
            stages[name]['traffic_GBps'] = round(gbs, 1)
            stages[name]['frac_of_peak_by_traffic'] = round(gbs / HBM_PEAK_GBS, 4)
    if live:
        traffic_meta = {'traffic_source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, two passes of this command (3 steps) '
                                          'started by this run; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch',
                        'traffic_measured_in_this_run': True}
        for name in stages:
            if name in live:
                quote(name, live[name])
    else:
        try:
            import glob
            pmc_files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc.json')))
            pmc = json.load(open(pmc_files[-1])) if pmc_files else {}
            wl = pmc.pop('_workload', None)
            commit = pmc.pop('_commit', None)
            if wl == {'streams_per_gpu': B, 'frames_per_call': T, 'dtype': args.precision}:
                traffic_meta = {'traffic_source': os.path.basename(pmc_files[-1]), 'traffic_commit': commit,
                                'traffic_measured_in_this_run': False}
                for name in stages:
                    for entry in pmc.values():
                        if isinstance(entry, dict) and entry.get('class') == name and 'hbm_bytes' in entry:
                            quote(name, entry['hbm_bytes'])
        except Exception:
            pass
    roofline = dict(stages[dominant])
    roofline['kernel'] = dominant
    # `frac` divides by the launch duration of the EVENT-INSTRUMENTED pass, whose launches are serialised by their event pairs and
    # a few per cent slower than in the timed pass (`stages_pass`); the same fraction with every class time scaled by
    # timed step / sum of class times is what the timed run reached
    timed_ms = elapsed_max / args.steps * 1e3
    if dev_ms > 0:
        roofline['frac_scaled_to_timed_pass'] = round(roofline['frac'] * dev_ms / timed_ms, 5)
    roofline.setdefault('traffic', None)
    roofline.update(traffic_meta)

    extra = None
    if rank == 0 and world == 1 and not args.no_extra:
        extra = extra_points(args, torch, np, koala_amd, model, kb, x, dx, dy, base, local_rank)
    # the constants the fractions are divided by, next to what a plain device-to-device copy reaches on this box
    peaks = {'hbm_GBps': HBM_PEAK_GBS, 'mfma_TFLOPs': MFMA_PEAK_TFLOPS[args.precision], 'board_sustained': sustained}
    if rank == 0:
        src = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
        dst = torch.empty_like(src)
        src.fill_(1)
        for _ in range(3):
            dst.copy_(src)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(20):
            dst.copy_(src)
        torch.cuda.synchronize()
        peaks['measured_d2d_copy_GBps'] = round(2.0 * src.numel() * 20 / (time.perf_counter() - c0) / 1e9, 1)
        del src, dst

    kb.delete()
    if collective:
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
            'throughput_all_reduce': collective,  # 'nccl' (= RCCL) under a launcher, None for a plain single-process run
            'host_cores_per_rank': cores_per_rank,
        },
        'real_time_factor': round(elapsed_max / (args.steps * T * 256 / 16000.0) / B, 9),
        'frames_per_sec_per_gpu': round(value / world, 1),
        'roofline': roofline,
        'peaks': peaks,
        'stages': stages,
        # the per-class times above come from a SECOND pass with a HIP event pair around every launch; that pass runs a few per
        # cent slower than the timed one (the events serialise the launches): both step times, so the shares can be scaled
        'stages_pass': {'ms_per_step': round(instrumented_ms_per_step, 4), 'sum_of_class_ms': round(dev_ms, 4),
                        'timed_ms_per_step': round(elapsed_max / args.steps * 1e3, 4)},
        # HIP-event time of every timed step, what priming did, and the board's clock / power right before the timed region
        'timing': {'per_step_ms': per_step_ms, 'prime': prime, 'board_before_t0': board_before},
        'cpu_baseline': cpu,
        'parity': parity,
        'extra': extra,
    }
    print(json.dumps(line))
    sys.stdout.flush()
    if parity is not None and not parity['pass']:
        print('bench.py: the timed engine misses its parity bars: %r' % (parity,), file=sys.stderr)
        sys.exit(3)


if __name__ == '__main__':
    if '--cpu-leg-child' in sys.argv:  # (before anything touches torch or the GPU)
        a = parse_args()
        cpu_leg_child(a.frames, a.streams)
        sys.exit(0)
    main()
