"""
Multi-rank path on real hardware, as far as a one-GPU box can show it (BASELINE configs[3]): two ENGINE
ranks (separate processes, separate HIP contexts) share GPU 0, rendezvous over gloo on 127.0.0.1, each runs its own
contiguous shard of the streams through libpv_koala.so; the union must equal the unsharded engine bit for bit, and
`bench.py --gpus 2` must launch its own two ranks and report n_gpus = 2.  (On an 8-GPU node the same code runs one rank
per GPU with backend nccl = RCCL; nothing on the data path changes.)
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import koala_amd
from conftest import ROOT, model_file, synth_streams

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
import torch
import torch.distributed as dist
import koala_amd
from conftest import synth_streams
from koala_amd.sharding import aggregate_throughput, shard_range
dist.init_process_group('gloo', init_method='env://')
rank, world = dist.get_rank(), dist.get_world_size()
N, T, precision = int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
b, e = shard_range(N, rank, world)
x = synth_streams(N, 2 * T, seed=77)[b:e]
kb = koala_amd.create_batch('key', e - b, T, precision, model_path=sys.argv[2], device='gpu:0')
y = np.concatenate([kb.process(np.ascontiguousarray(x[:, c * T * 256:(c + 1) * T * 256])) for c in range(2)], axis=1)
kb.delete()
np.save(os.path.join(sys.argv[3], 'shard%d.npy' % rank), y)
frames, elapsed = aggregate_throughput((e - b) * 2 * T, 1.0 + rank)
if rank == 0:
    open(os.path.join(sys.argv[3], 'agg.txt'), 'w').write('%d %f' % (frames, elapsed))
dist.barrier()
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize('precision,N,T', [('bf16', 600, 4), ('fp32', 75, 3)])
def test_two_engine_ranks_equal_the_unsharded_engine(tmp_path, precision, N, T):
    model = model_file('random', 1234)
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, model, str(tmp_path), str(N), str(T), precision],
                                      env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    x = synth_streams(N, 2 * T, seed=77)
    kb = koala_amd.create_batch('key', N, T, precision, model_path=model, device='gpu:0')
    whole = np.concatenate([kb.process(np.ascontiguousarray(x[:, c * T * 256:(c + 1) * T * 256])) for c in range(2)], axis=1)
    kb.delete()
    got = np.concatenate([np.load(tmp_path / ('shard%d.npy' % r)) for r in range(2)])
    assert np.array_equal(got, whole)
    frames, elapsed = (tmp_path / 'agg.txt').read_text().split()
    assert int(frames) == N * 2 * T and abs(float(elapsed) - 2.0) < 1e-9


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: two ranks appear (sharing GPU 0 here, gloo in place of RCCL),
    the line says n_gpus = 2 and the aggregate counts both ranks' streams."""
    env = dict(os.environ, KOALA_BENCH_SHARE_GPU='1')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dist-backend', 'gloo', '--steps', '3',
                          '--warmup', '1', '--prime-seconds', '0', '--streams', '512', '--frames', '8', '--no-cpu-baseline',
                          '--no-extra'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['global_streams'] == 1024
    assert abs(line['value'] - 1024 * 8 * 3 / (line['ms_per_step'] * 3e-3)) / line['value'] < 1e-3


def test_eight_ranks_like_configs3():
    """BASELINE configs[3]'s shape on the one GPU this box has: EIGHT ranks (separate processes and HIP contexts, all on GPU 0,
    gloo in place of RCCL) x 512 streams each -- an 8-way rendezvous, `shard_range(8 * 512, r, 8)` per rank, per-rank host-core
    sets, max-over-ranks timing, and the aggregate of all eight on the line.  (On an 8-GPU node the driver runs the same command
    without KOALA_BENCH_SHARE_GPU and with the nccl backend: one engine and one GPU per rank.)"""
    env = dict(os.environ, KOALA_BENCH_SHARE_GPU='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--dist-backend', 'gloo', '--steps', '3',
                          '--warmup', '1', '--prime-seconds', '0', '--sustain-seconds', '0', '--streams', '512', '--frames', '8',
                          '--no-cpu-baseline', '--no-extra'], env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 8 and line['config']['global_streams'] == 8 * 512 and line['scaling'] == 'weak'
    assert abs(line['value'] - 8 * 512 * 8 * 3 / (line['ms_per_step'] * 3e-3)) / line['value'] < 1e-3
    assert abs(line['frames_per_sec_per_gpu'] * 8 - line['value']) / line['value'] < 1e-3
    aff = line['config']['host_cores_per_rank']
    assert aff is None or aff >= 1


def test_rccl_branch_runs_with_one_rank_under_torchrun():
    """The driver starts multi-GPU runs as `python -m torch.distributed.run ... bench.py --gpus N` with backend nccl (= RCCL).
    A one-GPU box cannot host two RCCL ranks, but it can host ONE: communicator creation on the device
    (`init_process_group('nccl', device_id=...)`), the barriers and the all-reduce of device tensors all execute."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                          '--master-port', str(free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
                          '--prime-seconds', '0', '--sustain-seconds', '0', '--streams', '512', '--frames', '8', '--no-cpu-baseline',
                          '--no-extra'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['config']['throughput_all_reduce'] == 'nccl'
    assert abs(line['value'] - 512 * 8 * 3 / (line['ms_per_step'] * 3e-3)) / line['value'] < 1e-3
