"""N>1 path with the REAL kernels: two processes share cuda:0 (gloo rendezvous; the collective bounces through the host, which
is what keynet_amd.dist does for gloo + device tensors), each runs its shard of one encrypted batch through KeyedModel on the
HIP path, and the gathered logits must equal the single-process logits bit for bit -- even and ragged batch sizes, a bit-exact
CSR key-net and a tiled (MFMA) key-net.  Also launches `bench.py --gpus 2` exactly as a user would (no launcher, no WORLD_SIZE):
it must start its own ranks and print ONE JSON line with n_gpus == 2."""
import json
import os
import subprocess
import sys

import pytest

import dist_harness

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('kind,n', [('lenet', 8), ('lenet', 7), ('tiled', 9)])
def test_real_keynet_sharded_over_two_ranks(kind, n):
    res = dist_harness.run(kind, n)
    assert [r[1] for r in res] == [True, True], res
    assert res[0][2][0] == 0 and res[0][2][1] == res[1][2][0] and res[1][2][1] == n
    assert all(r[3][0] == n and r[4].startswith('cuda') for r in res)


def test_one_ranks_data_trips_the_contract_for_all():
    """Float-key mini-net under the 'auto' contract on two ranks; rank 1's images are 300x larger.  After sharded_forward both ranks run
    every layer under the same contract, each rank's shard of the gathered logits equals its own forward under those contracts, and the
    PEER's shard recomputed locally is bit-equal too (replicas stay bit-identical: what bench.py's collective record asserts)."""
    res = dist_harness.run_contract('device')
    ((_, st0, own0, peer0, sw0), (_, st1, own1, peer1, sw1)) = res
    assert st0 == st1, (st0, st1)
    assert all(v in (True, False) for v in st0.values())
    assert own0 and own1 and peer0 and peer1, res


def test_bench_starts_its_own_ranks():
    env = dict(os.environ, KN_BENCH_SHARE_GPU='1')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', 'lenet', '--steps', '3', '--warmup', '1', '--layer-iters', '1'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['config']['global_batch'] == 2 * r['config']['images_per_gpu'] and r['value'] > 0 and r['parity']['ok']
