"""
Multi-process (world_size 2, gloo, CPU) test of the N > 1 host logic: contiguous stream sharding with no data-path
collective and the final {sum frames, max elapsed} all-reduce.  Each rank runs its shard through the
CPU oracle standing in for its GPU; the union must equal the unsharded result bit for bit.
"""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT, model_file, synth_streams
from koala_amd.sharding import aggregate_throughput, shard_range

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
import torch.distributed as dist
from conftest import synth_streams
from koala_amd.sharding import aggregate_throughput, shard_range
from oracle import oracle
dist.init_process_group('gloo', init_method='env://')
rank, world = dist.get_rank(), dist.get_world_size()
N, T = 10, 5
b, e = shard_range(N, rank, world)
x = synth_streams(N, T, seed=21)[b:e]
y = oracle.Oracle(sys.argv[2], e - b).process(x)
np.save(os.path.join(sys.argv[3], 'shard%d.npy' % rank), y)
frames, elapsed = aggregate_throughput((e - b) * T, 1.0 + rank)
if rank == 0:
    open(os.path.join(sys.argv[3], 'agg.txt'), 'w').write('%d %f' % (frames, elapsed))
dist.destroy_process_group()
'''


def test_shard_ranges_partition_the_streams():
    for n in (1, 7, 8, 4096, 32768, 10):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(32768, 3, 8) == (12288, 16384)  # BASELINE configs[3]: 4096 streams per GPU


def test_single_process_aggregate_is_identity():
    assert aggregate_throughput(123, 0.5) == (123, 0.5)


def test_two_ranks_gloo(tmp_path):
    model = model_file('random', 1234)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   OMP_NUM_THREADS='2')
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, model, str(tmp_path)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    from oracle import oracle
    whole = oracle.Oracle(model, 10).process(synth_streams(10, 5, seed=21))
    got = np.concatenate([np.load(tmp_path / ('shard%d.npy' % r)) for r in range(2)])
    assert np.array_equal(got, whole)
    frames, elapsed = (tmp_path / 'agg.txt').read_text().split()
    assert int(frames) == 50 and abs(float(elapsed) - 2.0) < 1e-9


def test_bench_refuses_a_world_size_it_was_not_asked_for():
    """`--gpus N` must describe the job that ran: a launcher-provided WORLD_SIZE that differs is an error, not a line."""
    env = dict(os.environ, WORLD_SIZE='4', RANK='0', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 2 and 'WORLD_SIZE=4' in out.stderr and not out.stdout.strip()
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 2
