"""One keying per node: the ranks of `bench.py --gpus N` share ONE keyed net (measured on the GPU box, tools/time_startup.py, VGG-16: one keying 26 s, eight
concurrent keyings 58-62 s each -- they compete for memory bandwidth).

Rank 0 keys and writes the neutral archive (keynet_amd.io, uncompressed) into an ANONYMOUS file -- O_TMPFILE in /dev/shm, never linked into the directory -- and
broadcasts (pid, fd) over the process group; the other ranks re-open the inode through /proc/<pid>/fd/<fd>, everyone meets at a barrier, rank 0 closes its
descriptor and the loaders read.  What this buys over the round-5 scheme (a predictable /dev/shm/keynet_bench_<name>_<port>_<uid>.npz, removed by atexit):
  * nothing to clean up and nothing to find: a crashed, SIGTERMed or SIGKILLed run leaves no archive (the memory goes when the last descriptor closes), so a
    later run can never load a stale net while its rank 0 keys a fresh one (round-5 advisor finding), and no other user can read or pre-create it (mode 0600,
    no name);
  * failures are collective: a rank 0 that cannot key broadcasts the error and EVERY rank raises; a loader that cannot open or parse the archive keys for itself
    and says so -- the `collective` record's peer-shard recompute then proves bit for bit that loaded == keyed.
The process group must exist before this is called (bench.py initialises it first at N > 1: nothing forks there, the scipy baseline is an N = 1 leg)."""
import os
import time

import torch.distributed as dist

from keynet_amd import io as kio
from .common import log
from .workloads import build_workload


def _anonymous_file():
    """(fd, where) of a read-write file with no name: O_TMPFILE in /dev/shm (tmpfs: held in memory) or /tmp; where O_TMPFILE is unsupported, mkstemp + unlink."""
    import tempfile
    for d in ('/dev/shm', tempfile.gettempdir()):
        if not os.path.isdir(d):
            continue
        try:
            return (os.open(d, getattr(os, 'O_TMPFILE', 0o20200000) | os.O_RDWR, 0o600), d + ' (O_TMPFILE)')
        except OSError:
            try:
                (fd, p) = tempfile.mkstemp(prefix='keynet_bench_', dir=d)
                os.unlink(p)
                return (fd, d + ' (unlinked)')
            except OSError:
                continue
    raise OSError('no directory for the shared archive')


def _barrier(what):
    """dist.barrier that a Python signal handler can interrupt (the work is polled, the main thread keeps returning to the interpreter)."""
    w = dist.barrier(async_op=True)
    t0 = time.time()
    while not w.is_completed():
        time.sleep(0.02)
        if time.time() - t0 > 1800:
            raise RuntimeError('barrier "%s" did not complete in 30 min' % what)
    w.wait()


def build_workload_shared(name, rank, world, exact=None):
    """build_workload for the ranks of ONE node (bench contract: --nnodes=1, so rank 0 is the node's first rank).  Ranks other than 0 get net = None (only rank 0
    evaluates the plain network for the parity gate).  Returns the tuple of build_workload."""
    forced = os.environ.get('KN_BENCH_TEST_FORCE_SHARED') == '1'            # test-only: run the protocol with one rank (its collectives on the RCCL backend of a single-GPU box)
    if (world <= 1 and not forced) or not dist.is_available() or not dist.is_initialized():
        return build_workload(name, rank, exact=exact)
    out = None
    fd = -1
    msg = None
    if rank == 0:
        try:
            out = build_workload(name, rank, exact=exact)
            (sensor, knet, inshape, batch, desc, net) = out
            t0 = time.time()
            (fd, where) = _anonymous_file()
            with os.fdopen(os.dup(fd), 'wb') as f:
                kio.save_keynet(knet, f, sensor=sensor, compress=False)
            size = os.fstat(fd).st_size
            msg = {'pid': os.getpid(), 'fd': fd, 'bytes': size, 'inshape': list(inshape), 'batch': batch, 'desc': desc}
            log('[bench rank 0] keyed net handed to the other %d ranks through an anonymous file in %s (%.1f s, %.0f MB)' % (world - 1, where, time.time() - t0, size / 1e6))
        except Exception as e:                               # every rank must learn of it: the others are waiting in the broadcast
            msg = {'error': '%s: %s' % (type(e).__name__, e)}
    box = [msg]
    dist.broadcast_object_list(box, src=0)
    msg = box[0]
    if 'error' in msg:
        raise RuntimeError('rank 0 could not key "%s": %s' % (name, msg['error']))
    f = None
    t_wait = time.time()
    if rank != 0:
        try:
            f = open('/proc/%d/fd/%d' % (msg['pid'], msg['fd']), 'rb', buffering=1 << 24)        # the same inode through rank 0's descriptor table; our own open file description
            assert os.fstat(f.fileno()).st_size == msg['bytes'], 'archive size differs from what rank 0 wrote'
        except (OSError, AssertionError) as e:
            log('[bench rank %d] cannot open rank 0\'s archive (%s): keying locally' % (rank, e))
            f = None
    if os.environ.get('KN_BENCH_TEST_DIE_BEFORE_BARRIER') == str(rank):
        os._exit(17)                                         # test-only (tests/test_dist_gpu.py): a loader dies holding the archive; never set by the driver
    _barrier('archive opened')                               # every loader holds the inode now
    if rank == 0:
        os.close(fd)                                         # the archive lives exactly as long as somebody is reading it
        return out
    if f is not None:
        try:
            t1 = time.time()
            with f:
                (sensor, knet) = kio.load_keynet(f, with_sensor=True)
            log('[bench rank %d] loaded rank 0\'s keyed net (%.0f MB) in %.1f s' % (rank, msg['bytes'] / 1e6, time.time() - t1))
            return (sensor, knet, tuple(msg['inshape']), msg['batch'], msg['desc'], None)
        except Exception as e:
            log('[bench rank %d] could not load the archive (%s: %s): keying locally' % (rank, type(e).__name__, e))
    (sensor, knet, inshape, batch, desc, _) = build_workload(name, rank, exact=exact)
    log('[bench rank %d] keyed locally, %.1f s after rank 0\'s message' % (rank, time.time() - t_wait))
    return (sensor, knet, inshape, batch, desc, None)
