"""bench.py's one-keying-per-node hand-over (benchlegs/shared.py) on CPU: gloo ranks, the real LeNet key-net.  Rank 0 keys and writes an ANONYMOUS archive, the others
re-open it through /proc/<pid>/fd/<fd>: every rank ends with the same operators, nothing with a name ever appears in /dev/shm, a rank 0 that cannot key fails
EVERY rank, and a loader that dies before the barrier leaves nothing behind however the others are stopped."""
import hashlib
import os
import signal
import sys
import time

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dist_harness import free_port   # noqa: E402


def _shm_names():
    # (sem.mp-*: the semaphores of this test's own multiprocessing queue)
    return sorted(n for n in os.listdir('/dev/shm') if not n.startswith('sem.')) if os.path.isdir('/dev/shm') else []


def _digest(knet, sensor):
    from keynet_amd import sparse as ksp
    from keynet_amd.layer import KeyedLayer
    h = hashlib.sha256()
    for (name, c) in knet._keynet.named_children():
        h.update(name.encode())
        if isinstance(c, KeyedLayer):
            for a in ksp._stored_order_csr(c.W._matrix):
                h.update(np.ascontiguousarray(a).tobytes())
    for a in ksp._stored_order_csr(sensor._encryptkey.tocsr()):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _worker(rank, world, port, mode, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from benchlegs import shared
    if mode == 'rank0_fails' and rank == 0:
        def boom(*a, **k):
            raise MemoryError('no room to key')
        shared.build_workload = boom
    if mode == 'loader_dies' and rank == world - 1:
        shared._barrier = lambda what: os._exit(17)             # dies holding the archive open, before the barrier
    if mode == 'cannot_open' and rank == 1:
        real_open = open

        def deny(path, *a, **k):
            if str(path).startswith('/proc/'):
                raise PermissionError(path)
            return real_open(path, *a, **k)
        shared.open = deny                                       # module-level name shadows the builtin inside benchlegs.shared only
    try:
        (sensor, knet, inshape, batch, desc, net) = shared.build_workload_shared('lenet', rank, world)
        q.put((rank, 'ok', _digest(knet, sensor), tuple(inshape), batch, net is not None, [n for n in _shm_names() if n.startswith('keynet_bench')]))
    except RuntimeError as e:
        q.put((rank, 'raised', str(e)))
        raise SystemExit(3)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, mode):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    return (procs, q)


@pytest.mark.parametrize('mode', ['normal', 'cannot_open'])
def test_every_rank_gets_rank0s_keynet(mode):
    before = _shm_names()
    (procs, q) = _run(3, mode)
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
    assert [p.exitcode for p in procs] == [0, 0, 0]
    assert [r[1] for r in res] == ['ok'] * 3
    assert len(set(r[2] for r in res)) == 1, 'replicas do not hold the same keys'       # (the rank that keyed for itself too: the seeds make keying deterministic)
    assert all(r[3] == (1, 28, 28) and r[4] == 1024 for r in res)
    assert [r[5] for r in res] == [True, False, False]                                    # only rank 0 keeps the source network
    assert all(r[6] == [] for r in res) and _shm_names() == before                        # the archive never had a name


def test_rank0_failure_is_collective():
    (procs, q) = _run(3, 'rank0_fails')
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
    assert [r[1] for r in res] == ['raised'] * 3 and all('no room to key' in r[2] for r in res)
    assert all(p.exitcode not in (0, None) for p in procs)


def test_a_loader_dying_before_the_barrier_leaves_nothing_behind():
    """The survivors are waiting in the barrier holding the anonymous file.  gloo notices the lost peer and fails the barrier (they exit non-zero by themselves);
    a backend that keeps waiting gets what torch.distributed.run sends when a worker dies -- SIGTERM, then SIGKILL to the keying rank, which cannot run any cleanup.
    Either way no archive is left: it never had a name (the round-5 scheme left /dev/shm/keynet_bench_*.npz behind in exactly this case)."""
    before = _shm_names()
    (procs, q) = _run(3, 'loader_dies')
    t0 = time.time()
    while procs[2].exitcode is None and time.time() - t0 < 300:
        time.sleep(0.1)
    assert procs[2].exitcode == 17
    assert _shm_names() == before
    time.sleep(2.0)
    if procs[1].is_alive():
        os.kill(procs[1].pid, signal.SIGTERM)
    if procs[0].is_alive():
        os.kill(procs[0].pid, signal.SIGKILL)
    for p in procs[:2]:
        p.join(60)
    assert all(p.exitcode not in (0, None) for p in procs), [p.exitcode for p in procs]
    assert _shm_names() == before
