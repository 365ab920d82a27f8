"""The RCCL branch itself, on the one GPU a test box has: a fresh process initialises a `nccl` (= RCCL on ROCm) process group of
world size 1 bound to cuda:0, runs keynet_amd.dist.sharded_forward on the REAL key-nets (HIP path), and the device
all_gather_into_tensor of keynet_amd/dist.py must have run on a cuda tensor and returned logits bit-identical to the plain forward.
(Two ranks cannot share one GPU under RCCL -- "duplicate GPU" -- so world sizes > 1 are covered with gloo in test_dist_gpu.py and
test_dist_gloo.py; what is proved here is communicator set-up, the device collective and its stream ordering against the HIP kernels.)
Also `bench.py --gpus 1 --dist`: the same through the bench's own init path, with the collective's cost in the JSON line."""
import json
import os
import subprocess
import sys

import pytest

import dist_harness

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('kind,n', [('lenet', 8), ('tiled', 9)])
def test_rccl_world_size_one_device_collective(kind, n):
    res = dist_harness.run(kind, n, world_size=1, backend='nccl')
    assert len(res) == 1
    (rank, equal, bounds, shape, device, backend, calls) = res[0]
    assert equal, 'gathered logits differ from the plain forward'
    assert backend == 'nccl' and device.startswith('cuda') and bounds == (0, n) and shape[0] == n
    assert len(calls) == 1 and calls[0][0].startswith('cuda') and calls[0][1][0] == n, calls      # the DEVICE all_gather_into_tensor ran


def test_bench_dist_flag_runs_the_rccl_path_on_one_gpu():
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'KN_BENCH_SHARE_GPU'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--dist', '--workload', 'lenet', '--steps', '3', '--warmup', '1',
                        '--layer-iters', '1', '--no-cpu-baseline', '--no-secondary'], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    c = r['collective']
    assert r['n_gpus'] == 1 and c['backend'] == 'nccl' and c['ranks_seen'] == 1 and c['gather_ms'] > 0 and c['bytes_per_rank'] == 4 * 1024 * 10 and c['rank_images_per_s']['min'] > 0
    assert c['every_rank_shard_bit_equal_to_its_local_forward'] is True and c['peer_shard_recomputed_on_rank0']['bit_equal'] is True and r['parity']['ok']
    assert c['ranks'] == [[0, 0]] and len(lines[0]) < 4096                 # the compact line: [rank, device_index] pairs, details in bench_detail.json
    full = json.load(open(os.path.join(ROOT, r['detail'])))
    assert full['collective']['rank0_shard_bit_equal'] is True and full['collective']['ranks'][0]['device_name']


def test_keyed_net_handover_protocol_runs_on_the_rccl_backend():
    """benchlegs/shared.py's hand-over (broadcast_object_list of the archive's (pid, descriptor), the polled asynchronous barrier) has only ever run on gloo with more than one
    rank: there is no multi-GPU box in the builder's pool.  Forced with ONE rank on the real RCCL backend (KN_BENCH_TEST_FORCE_SHARED: test-only), the same calls must work --
    rank 0 keys, writes the anonymous archive, broadcasts to itself, passes the barrier -- and the bench line must come out as usual."""
    env = dict(os.environ, KN_BENCH_TEST_FORCE_SHARED='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'KN_BENCH_SHARE_GPU'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--dist', '--workload', 'lenet', '--steps', '3', '--warmup', '1',
                        '--layer-iters', '1', '--no-cpu-baseline', '--no-secondary'], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert 'keyed net handed to the other 0 ranks through an anonymous file' in p.stderr, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r['collective']['backend'] == 'nccl' and r['parity']['ok'] and r['parity']['oracle_bit_equal'] is True
