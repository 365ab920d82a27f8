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


@pytest.mark.parametrize('kind', ['lenet', 'tiled'])
def test_real_keynet_sharded_over_eight_ranks_ragged(kind):
    """BASELINE configs[4] is an 8-rank job: the first 8-GPU run must not be the first 8-rank run.  Eight processes share cuda:0 (gloo), a
    ragged total of 250 images (shards of 32 and 31), the real key-nets: every rank's gathered block equals the single-process forward
    of the whole batch bit for bit, the shard bounds tile [0, 250) in rank order, one collective per forward."""
    n = 250
    res = dist_harness.run(kind, n, world_size=8, timeout=600)
    assert [r[0] for r in res] == list(range(8)) and all(r[1] for r in res), res
    bounds = [r[2] for r in res]
    assert bounds[0][0] == 0 and bounds[-1][1] == n and all(bounds[k][1] == bounds[k + 1][0] for k in range(7))
    assert sorted(set(hi - lo for (lo, hi) in bounds)) == [31, 32]
    assert all(r[3][0] == n and r[4].startswith('cuda') and r[5] == 'gloo' for r in res)
    assert all(len(r[6]) == 1 for r in res), [r[6] for r in res]          # ONE all_gather_into_tensor per sharded forward, nothing else on the data path


def test_one_ranks_data_trips_the_contract_for_all_eight():
    """The 'auto' contract on eight ranks: only the last rank's images (300x larger) can trip a tolerance screen; after sharded_forward ALL
    eight run every layer under the same contract, own and peer shards are bit-equal on recompute."""
    res = dist_harness.run_contract('device', world_size=8, timeout=600)
    states = [r[1] for r in res]
    assert all(st == states[0] for st in states), states
    assert all(v in (True, False) for v in states[0].values())
    assert all(r[2] and r[3] for r in res), res


def test_bench_eight_ranks_on_one_gpu():
    """`python bench.py --gpus 8` as the driver's multi-GPU tier starts it (its own ranks through torch.distributed.run), on a shared GPU with
    gloo: the control flow of cfg5 -- 8 keyings, 8 uploads, the 8-way gather, max-over-ranks timing, the 8-rank compact line -- on a real box."""
    import time
    env = dict(os.environ, KN_BENCH_SHARE_GPU='1')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--workload', 'lenet', '--steps', '3', '--warmup', '1', '--layer-iters', '1'],
                       env=env, capture_output=True, text=True, timeout=900)
    wall = time.time() - t0
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and len(lines[0]) < 4096, (len(lines), [len(l) for l in lines])
    r = json.loads(lines[0])
    c = r['collective']
    assert r['n_gpus'] == 8 and r['config']['global_batch'] == 8 * r['config']['images_per_gpu'] and r['value'] > 0 and r['parity']['ok']
    assert c['ranks_seen'] == 8 and sorted(x[0] for x in c['ranks']) == list(range(8))
    assert c['every_rank_shard_bit_equal_to_its_local_forward'] is True and c['peer_shard_recomputed_on_rank0'] == {'peer_rank': 7, 'bit_equal': True}
    assert r['scaling'] == 'weak' and 'N=1' in json.dumps(r.get('cpu_baseline'))
    assert wall < 120, wall


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


def _bench(args, extra_env=None, timeout=900):
    env = dict(os.environ, KN_BENCH_SHARE_GPU='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=timeout)


def _shm_archives():
    return sorted(n for n in os.listdir('/dev/shm') if n.startswith('keynet_bench')) if os.path.isdir('/dev/shm') else []


def test_bench_eight_ranks_share_one_keying_of_a_tiled_vgg():
    """cfg5's control flow with cfg5's KIND of key-net (21 keyed layers of VGG-16, tiled conv operators; reduced width so that the tier stays short -- the full-size
    8-rank run is profiles/r06_vgg16_bench_8ranks_one_gpu.json): rank 0 keys ONCE, seven ranks load its anonymous archive (benchlegs/shared.py), every rank uploads and
    runs its shard on the matrix cores under the calibrated contract, one all-gather per step.  The line validates itself: backend, ranks seen, bytes gathered per rank,
    the gather's own time, slowest / fastest rank, every shard bit-equal to its rank's local forward and the LAST rank's shard recomputed on rank 0 bit-equal
    (= loaded operators == keyed operators)."""
    before = _shm_archives()
    p = _bench(['--gpus', '8', '--workload', 'vgg16-slice', '--batch', '16', '--steps', '2', '--warmup', '1', '--layer-iters', '1', '--no-cpu-baseline'])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and len(lines[0]) < 4096
    r = json.loads(lines[0])
    c = r['collective']
    assert r['n_gpus'] == 8 and r['config']['global_batch'] == 128 and r['value'] > 0 and r['parity']['ok'] and r['scaling'] == 'weak'
    assert c['backend'] == 'gloo' and c['ranks_seen'] == 8 and sorted(x[0] for x in c['ranks']) == list(range(8))
    assert c['bytes_per_rank'] == 16 * 10 * 4 and c['gather_ms'] > 0
    assert 0 < c['rank_images_per_s']['min'] <= c['rank_images_per_s']['max']
    assert c['every_rank_shard_bit_equal_to_its_local_forward'] is True and c['peer_shard_recomputed_on_rank0'] == {'peer_rank': 7, 'bit_equal': True}
    assert p.stderr.count('loaded rank 0\'s keyed net') == 7 and 'keying locally' not in p.stderr      # seven loaders, nobody fell back to its own keying
    assert _shm_archives() == before


def test_bench_rank_dying_before_the_barrier_fails_the_job_and_leaves_no_archive():
    """A loader dies holding rank 0's archive, before the barrier (KN_BENCH_TEST_DIE_BEFORE_BARRIER: test-only).  The job must end non-zero -- no line, no hang, no rank
    re-executed -- and nothing may be left in /dev/shm: the archive is an anonymous file (it never had a name), so there is nothing a killed rank 0 would have to clean up."""
    import time
    before = _shm_archives()
    t0 = time.time()
    p = _bench(['--gpus', '4', '--workload', 'vgg16-slice', '--batch', '8', '--steps', '1', '--warmup', '0', '--layer-iters', '1', '--no-cpu-baseline'],
               extra_env={'KN_BENCH_TEST_DIE_BEFORE_BARRIER': '2'}, timeout=600)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert time.time() - t0 < 300
    assert _shm_archives() == before
    assert p.stderr.count('keyed vgg16-slice on the host') == 1                # one keying: nobody restarted
