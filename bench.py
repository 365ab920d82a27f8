#!/usr/bin/env python3
"""bench.py -- keyed forward throughput on MI355X (BASELINE.json metric: encrypted images/sec + roofline, keyed VGG-16 224x224).

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts N ranks itself, one per GPU, over RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path (KeyedModel.forward_linear: all keyed layers, ReLU fused) over one batch of
synthetic encrypted images already resident in HBM.  Default workload = BASELINE.json configs[3]:
    TiledPermutationKeynet(VGG16(num_classes=2622), (3,224,224), tile 64 -> effective 56/28/14/7), 256 images per GPU.
Random-init weights (torch.manual_seed(0)), keys from np.random.seed(0), images ~ N(0,1) (there is no network for
checkpoints or datasets).  With N GPUs every rank runs its own 256-image shard (weak scaling, no collective inside the
forward) and the step ends with ONE RCCL all-gather of the logits.

Rank 0 prints ONE JSON line.
  roofline      the dominant kernel of the mode that ran: convtaps_mfma_kernel against the f32-MFMA peak (tolerance mode of the
                tiled key-nets), the order-preserving kernels against the no-FMA VALU peak (--exact, AllConvNet), the CSR kernels
                against HBM (LeNet).  `traffic` (HBM bytes per forward of that kernel from a separate rocprofv3 --pmc pass) is
                quoted only when the committed pass was taken on THESE kernel sources (sha256 of keynet_amd/csrc recorded with it).
  exact         (default vgg16 run, N=1) the SAME key-net switched to the bit-exact contract (KeyedModel.exact_mode(True): every
                layer in the reference's accumulation order, no MFMA): images/s, ms/step, its own roofline (VALU without FMA,
                39.3 T MAC/s) and a parity record -- bit-equality with the CPU oracle on sampled rows of real conv layers.
  cpu_baseline  the reference's own arithmetic -- scipy.sparse.csr_matrix.dot (keynet/sparse.py:488-492) -- timed on this node's
                host cores BEFORE the GPU is touched (the worker pool is forked from a GPU-free process): conv1_1, conv5_1, every
                pool, fc6-8 measured directly on 1 thread and on all physical cores; the other layers extrapolated at the measured
                ns/(nz*column) and labelled so; plus the reference-faithful tocsr()+dot of one tiled layer (keynet/sparse.py:603-612).
  parity        this run's own gate: logits of the timed batch's first images against the source network in plain torch f32
                (the reference's criterion, atol 1e-3); a run that fails it raises instead of printing a number.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from keynet_amd import system as ksys          # noqa: E402
from keynet_amd import sparse as ksp           # noqa: E402
from keynet_amd import dist as kdist           # noqa: E402
from keynet_amd import io as kio               # noqa: E402
from keynet_amd.layer import KeyedLayer        # noqa: E402
from keynet_amd.models import VGG16, LeNet_AvgPool, AllConvNet   # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md:42 (dense f32-input MFMA = f32 vector peak)
PEAK_VALU_NOFMA_TMACS = 39.3     # the same vector peak with separate multiply and add (bit-exact contract): 157.3 / 4 T MAC/s
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md:36 (spec; 6.29 TB/s measured copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ----------------------------------------------------------------------------------------------------------------------------
# workload (host only: no GPU call in this section)
def build_workload(name, rank, exact=None):
    """(sensor, knet, inshape, per_gpu_batch, description, source network).  Deterministic under the seeds, identical on every rank."""
    t0 = time.time()
    if name == 'vgg16':
        torch.manual_seed(0)
        net = VGG16(num_classes=2622).eval()
        np.random.seed(0)
        (sensor, knet) = ksys.TiledPermutationKeynet((3, 224, 224), net, 64, exact=exact)
        (inshape, batch, desc) = ((3, 224, 224), 256, 'TiledPermutationKeynet VGG16(2622) 3x224x224 tile=64 (effective 56/28/14/7)')
    elif name == 'vgg16-gain':
        # the float-key variant of the same config that is constructible at full size: block permutation + block-local photometric gain
        # (every keyed entry carries the coefficient a_out[o] / a_in[i]; 1e-5 contract); orthogonal tile keys fill every tile in
        torch.manual_seed(0)
        net = VGG16(num_classes=2622).eval()
        np.random.seed(0)
        (sensor, knet) = ksys.Keynet((3, 224, 224), net, local_geometric='permutation', local_photometric='uniform_random_gain', beta=0.5,
                                     tileshape=(64, 64), blocksize=64, exact=exact)
        (inshape, batch, desc) = ((3, 224, 224), 256, 'Keynet(permutation + uniform_random_gain, tile=64) VGG16(2622) 3x224x224: float keys')
    elif name == 'vgg16-givens':
        # the reference's OWN float-key VGG-16 configuration (test/test_keynet.py:133-151, test_vgg16_orthogonal): block-local Givens rotations
        # (alpha = 2) + block-local affine photometric keys (beta = gamma = 1), tile = blocksize = 224 // 16 = 14, channel memory order.
        # Keyed directly in factored form (the reference route cannot build it: 15 G non-zeros); fill-in: ~9.0-9.3 slots per output pixel
        # on average, up to 19, every entry carries a coefficient.
        torch.manual_seed(0)
        net = VGG16(num_classes=2622).eval()
        np.random.seed(0)
        (sensor, knet) = ksys.Keynet((3, 224, 224), net, tileshape=(224 // 16, 224 // 16), global_geometric='identity', hierarchical_blockshape=(2, 2),
                                     hierarchical_permute_at_level=(0, 1, 2), local_geometric='givens_orthogonal', alpha=2.0, blocksize=224 // 16,
                                     local_photometric='uniform_random_affine', beta=1.0, gamma=1.0, memoryorder='channel', exact=exact)
        (inshape, batch, desc) = ((3, 224, 224), 256, 'Keynet(givens_orthogonal alpha=2 + uniform_random_affine beta=gamma=1, tile=blocksize=14) VGG16(2622) 3x224x224: '
                                                       'the float-key configuration of test/test_keynet.py:133-151')
    elif name == 'vgg16-givens28':
        # test/test_keynet.py:155-173 (test_vgg16_orthogonal_8): the same float-key family with tile = blocksize = 224 // 8 = 28
        torch.manual_seed(0)
        net = VGG16(num_classes=2622).eval()
        np.random.seed(0)
        (sensor, knet) = ksys.Keynet((3, 224, 224), net, tileshape=(224 // 8, 224 // 8), global_geometric='identity', hierarchical_blockshape=(2, 2),
                                     hierarchical_permute_at_level=(0, 1, 2), local_geometric='givens_orthogonal', alpha=2.0, blocksize=224 // 8,
                                     local_photometric='uniform_random_affine', beta=1.0, gamma=1.0, memoryorder='channel', exact=exact)
        (inshape, batch, desc) = ((3, 224, 224), 256, 'Keynet(givens_orthogonal alpha=2 + uniform_random_affine beta=gamma=1, tile=blocksize=28) VGG16(2622) 3x224x224: '
                                                       'the float-key configuration of test/test_keynet.py:155-173')
    elif name == 'vgg16-stochastic':
        # test/test_keynet.py:116-129 (test_vgg16_stochastic; the reference asserts 1e-5 there): hierarchical block permutation at levels 0, 1, 2 +
        # block-local doubly-stochastic keys (alpha = 2) + affine photometric keys, tile = blocksize = 14.  The INVERSE of a doubly-stochastic block
        # is dense, so every 14 x 14 block of a keyed operator fills in: ~490-560 (first layer of a stage: 1 700-5 400) slots per output pixel instead
        # of 9 -- 60x the multiply-adds of the permutation key-net (0.9 T per image), which is why this workload runs 16 images per step.
        torch.manual_seed(0)
        net = VGG16(num_classes=2622).eval()
        np.random.seed(0)
        (sensor, knet) = ksys.Keynet((3, 224, 224), net, tileshape=(224 // 16, 224 // 16), global_geometric='hierarchical_permutation', hierarchical_blockshape=(2, 2),
                                     hierarchical_permute_at_level=(0, 1, 2), local_geometric='doubly_stochastic', alpha=2.0, blocksize=224 // 16,
                                     local_photometric='uniform_random_affine', beta=1.0, gamma=1.0, memoryorder='channel', exact=exact)
        (inshape, batch, desc) = ((3, 224, 224), 16, 'Keynet(hierarchical_permutation levels 0-2 + doubly_stochastic alpha=2 + uniform_random_affine, tile=blocksize=14) VGG16(2622) '
                                                      '3x224x224: test/test_keynet.py:116-129')
    elif name == 'lenet':
        torch.manual_seed(0)
        net = LeNet_AvgPool().eval()
        np.random.seed(0)
        (sensor, knet) = ksys.PermutationKeynet((1, 28, 28), net)
        (inshape, batch, desc) = ((1, 28, 28), 1024, 'PermutationKeynet LeNet_AvgPool 1x28x28')
    elif name == 'allconv':
        torch.manual_seed(0)
        net = AllConvNet(batchnorm=False).eval()
        np.random.seed(0)
        (sensor, knet) = ksys.PermutationKeynet((3, 32, 32), net)
        (inshape, batch, desc) = ((3, 32, 32), 4096, 'PermutationKeynet AllConvNet 3x32x32 (BASELINE configs[2])')
    else:
        raise ValueError('unknown workload "%s"' % name)
    log('[bench rank %d] keyed %s on the host in %.1f s' % (rank, name, time.time() - t0))
    return (sensor, knet, inshape, batch, desc, net)


def build_workload_shared(name, rank, world, exact=None, wait_s=900.0):
    """build_workload for the ranks of ONE node: local rank 0 keys the net and hands the arrays to the others through the neutral archive
    (keynet_amd.io, uncompressed, in /dev/shm) instead of every rank keying for itself.  Measured on the GPU box (tools/time_startup.py,
    VGG-16): one keying 26 s, eight concurrent keyings 58-62 s each (they compete for memory bandwidth).  Falls back to keying locally when
    the archive does not appear.  Ranks other than 0 get net = None (only rank 0 evaluates the plain network for the parity gate)."""
    if world <= 1:
        return build_workload(name, rank, exact=exact)
    tag = '%s_%s_%d' % (name, os.environ.get('MASTER_PORT', '0'), os.getuid())
    path = os.path.join('/dev/shm' if os.path.isdir('/dev/shm') else '/tmp', 'keynet_bench_%s.npz' % tag)
    meta = path + '.json'
    if int(os.environ.get('LOCAL_RANK', rank)) == 0:
        out = build_workload(name, rank, exact=exact)
        (sensor, knet, inshape, batch, desc, net) = out
        t0 = time.time()
        try:
            kio.save_keynet(knet, path + '.tmp.npz', sensor=sensor, compress=False)
            json.dump({'inshape': list(inshape), 'batch': batch, 'desc': desc}, open(meta + '.tmp', 'w'))
            os.replace(meta + '.tmp', meta)
            os.replace(path + '.tmp.npz', path)                     # (the archive appears last, whole)
            import atexit
            atexit.register(lambda: [os.path.exists(f) and os.remove(f) for f in (path, meta)])
            log('[bench rank %d] keyed net handed to the other ranks through %s (%.1f s, %.0f MB)' % (rank, path, time.time() - t0, os.path.getsize(path) / 1e6))
        except OSError as e:
            log('[bench rank %d] could not write %s (%s): the other ranks key for themselves' % (rank, path, e))
        return out
    t0 = time.time()
    while not os.path.exists(path) and time.time() - t0 < wait_s:
        time.sleep(0.2)
    if not os.path.exists(path):
        log('[bench rank %d] no archive after %.0f s: keying locally' % (rank, wait_s))
        return build_workload(name, rank, exact=exact)
    t1 = time.time()
    (sensor, knet) = kio.load_keynet(path, with_sensor=True)
    m = json.load(open(meta))
    log('[bench rank %d] waited %.1f s for rank 0\'s keying, loaded the archive in %.1f s' % (rank, t1 - t0, time.time() - t1))
    return (sensor, knet, tuple(m['inshape']), m['batch'], m['desc'], None)


def keyed_layers(knet):
    return [(n, c) for (n, c) in knet._keynet.named_children() if isinstance(c, KeyedLayer)]


def host_nnz(c):
    """nnz of the operator the reference would apply (= algorithmic MACs per image), from the host description alone."""
    W = c.W
    if isinstance(W, ksp.Conv2dTiledMatrix) and W._taps is not None:
        t = W._taps
        return int(len(t['ent_out'])) * W._outshape[0] * W._inshape[0] + (int(np.count_nonzero(t['lastcol'])) if t['lastcol'] is not None else 0)
    if isinstance(W, ksp.TiledMatrix):
        return int(W.tocsr().nnz)
    return int(W.nnz())


# ----------------------------------------------------------------------------------------------------------------------------
# CPU baseline: scipy on the host cores, measured before any GPU call (fork-safe)
def host_cores():
    """(logical CPUs this process may run on, physical cores among them) from sched_getaffinity + lscpu."""
    aff = sorted(os.sched_getaffinity(0))
    phys = None
    try:
        out = subprocess.run(['lscpu', '-p=CPU,CORE,SOCKET'], capture_output=True, text=True, timeout=10).stdout
        seen = set()
        for line in out.splitlines():
            if line.startswith('#') or not line.strip():
                continue
            (cpu, core, sock) = [int(v) for v in line.split(',')[:3]]
            if cpu in aff:
                seen.add((sock, core))
        phys = len(seen) or None
    except Exception:
        phys = None
    return (len(aff), phys if phys else len(aff))


_CPU_JOBS = None    # [(name, [row band k of the operator], X)] inherited by the forked workers (copy-on-write)


def _cpu_worker(arg):
    (k, warm) = arg
    for (name, bands, X) in _CPU_JOBS:
        if bands[k].shape[0]:
            bands[k].dot(X[:, :1] if warm else X)
    return 0


def cpu_baseline(knet, workload, budget_s=24.0):
    """scipy.sparse.csr_matrix.dot (the call the reference makes: keynet/sparse.py:492; single-threaded _sparsetools.csr_matvecs)
    on this node's host cores.  VGG-16: conv1_1 (0.7 GB CSR), conv5_1 (3.4 GB), all pools and fc6-8 are expanded to the CSR the
    reference would hold and MEASURED; the remaining conv layers (up to 14.7 GB each) are extrapolated at the measured conv
    ns/(nz*column).  The number of batch columns per layer is sized to the time budget (csr_matvecs is linear in them)."""
    import multiprocessing as mp
    import scipy
    global _CPU_JOBS
    layers = keyed_layers(knet)
    nnz = {n: host_nnz(c) for (n, c) in layers}
    total_nnz = float(sum(nnz.values()))
    (logical, physical) = host_cores()
    rng = np.random.RandomState(0)
    measured = []
    sampled = {}
    for (n, c) in layers:
        W = c.W
        if isinstance(W, ksp.Conv2dTiledMatrix) and W._taps is not None:
            if workload == 'vgg16' and n not in ('conv1_1', 'conv5_1'):
                continue
            t0 = time.time()
            # rows (co, o) in the reference's order; conv5_1 is bounded to its first 128 of 512 output channels (0.84 of 3.4 GB:
            # every row walks the same columns, so ns/(nz*column) is that of the whole operator and first-touch of the CSR stays cheap)
            ch = 128 if (n == 'conv5_1' and W._outshape[0] > 128) else None
            M = W.rows_csr(None, channels=ch)
            sampled[n] = 'first %d of %d output channels, every pixel' % (ch, W._outshape[0]) if ch else 'whole operator'
            log('[bench cpu] expanded %s (%s) to CSR: %d nnz in %.1f s' % (n, sampled[n], M.nnz, time.time() - t0))
        elif isinstance(W, ksp.TiledMatrix):
            M = W.tocsr()
        else:
            M = W._matrix.tocsr()
        measured.append((n, M))
    def random_block(rows, cols):
        """[rows, cols] f32 activations: 8 independent N(0,1) columns tiled across (csr_matvecs has no data-dependent cost; drawing
        800 M normals for the pool layers would take longer than everything that is measured)."""
        base = rng.standard_normal((rows, min(cols, 8))).astype(np.float32)
        return np.ascontiguousarray(np.tile(base, (1, (cols + base.shape[1] - 1) // base.shape[1]))[:, :cols])

    # size the batch columns per layer to the time budget at this host's measured rate (calibrated on the smallest operator)
    (n0, M0) = min(measured, key=lambda nm: nm[1].nnz)
    X0 = random_block(M0.shape[1], 64)
    M0.dot(X0)
    t0 = time.perf_counter()
    for _ in range(3):
        M0.dot(X0)
    est_ns = max(1e9 * (time.perf_counter() - t0) / 3 / (M0.nnz * 64), 0.02) * 1.5
    share = budget_s / 2.0 / max(len(measured), 1)
    jobs = []
    for (n, M) in measured:
        cols = int(min(256, max(32, share / (est_ns * 1e-9 * max(M.nnz, 1)))))
        cols = max(32, (cols // 8) * 8)                      # >= 32 columns: below that the (col,val) stream, not the arithmetic, is what is timed
        jobs.append((n, M, random_block(M.shape[1], cols)))
    # (i) one thread: the reference's real behaviour
    rows = []
    for (n, M, X) in jobs:
        M.dot(X[:, :1])                                      # page in
        t0 = time.perf_counter()
        Y = M.dot(X)
        dt = time.perf_counter() - t0
        assert Y.dtype == np.float32
        rows.append(dict(layer=n, nnz=int(M.nnz), columns=int(X.shape[1]), seconds=dt, ns_per_nz_col=1e9 * dt / (M.nnz * X.shape[1])))
        log('[bench cpu] %-10s nnz=%10d x %3d columns  %.3f s  %.3f ns/(nz*col)  [scipy, 1 thread]' % (n, M.nnz, X.shape[1], dt, rows[-1]['ns_per_nz_col']))
    by = {r['layer']: r for r in rows}
    conv_rate = [r['ns_per_nz_col'] for r in rows if r['layer'].startswith('conv')]
    conv_big = by['conv5_1']['ns_per_nz_col'] if 'conv5_1' in by else (float(np.mean(conv_rate)) if conv_rate else float(np.mean([r['ns_per_nz_col'] for r in rows])))
    sec_per_image = 0.0
    extrapolated = []
    for (n, _) in layers:
        if n in by:
            sec_per_image += by[n]['ns_per_nz_col'] * 1e-9 * nnz[n]
        else:
            sec_per_image += conv_big * 1e-9 * nnz[n]
            extrapolated.append(n)
    res = dict(value=1.0 / sec_per_image, unit='images/s', cores=1, kind='port', engine='scipy.sparse.csr_matrix.dot (scipy %s), float32' % scipy.__version__,
               host={'logical_cpus': logical, 'physical_cores': physical},
               sample='measured directly on 1 thread: {%s}; extrapolated at the measured conv5_1 rate (%.3f ns per nz*column): {%s}; %.4g nnz per image'
                      % (', '.join('%s x%d cols%s' % (r['layer'], r['columns'], (' [%s]' % sampled[r['layer']]) if sampled.get(r['layer'], 'whole operator') != 'whole operator' else '')
                                   for r in rows), conv_big, ', '.join(extrapolated) or 'none', total_nnz),
               layers=rows)
    res['sample_short'] = ('%d of %d layers timed on 1 thread, %d-%d batch columns each, %.1f s of scipy work; %s; %.4g nnz/image'
                           % (len(rows), len(layers), min(r['columns'] for r in rows), max(r['columns'] for r in rows), sum(r['seconds'] for r in rows),
                              ('%d conv layers extrapolated at the conv5_1 rate %.3f ns/(nz*col)' % (len(extrapolated), conv_big)) if extrapolated else 'none extrapolated', total_nnz))
    # (ii) every physical core: one process per core, each owning a contiguous band of the operator's ROWS for all batch columns
    # (scipy's kernel is serial; rows are independent, so this is what a whole-host deployment of the same arithmetic would do)
    bands = None
    try:
        P = max(1, physical)
        bands = []
        for (n, M, X) in jobs:
            cut = np.searchsorted(M.indptr, np.linspace(0, M.nnz, P + 1)).clip(0, M.shape[0])      # equal non-zeros per band
            cut[0] = 0
            cut[-1] = M.shape[0]
            bands.append((n, [M[int(cut[k]):int(cut[k + 1])] for k in range(P)], X))
        _CPU_JOBS = bands
        ctx = mp.get_context('fork')                         # safe: nothing in this process has touched the GPU yet
        with ctx.Pool(P) as pool:
            pool.map(_cpu_worker, [(k, 1) for k in range(P)], chunksize=1)
            t0 = time.perf_counter()
            pool.map(_cpu_worker, [(k, 0) for k in range(P)], chunksize=1)
            par = time.perf_counter() - t0
        macs = float(sum(M.nnz * X.shape[1] for (_, M, X) in jobs))
        serial = float(sum(r['seconds'] for r in rows))
        res['all_cores'] = dict(value=res['value'] * serial / par, unit='images/s', cores=P,
                                sample='the same measured layers and columns, operator rows banded over %d processes (one per physical core): %.2f s wall vs %.2f s on one '
                                       'thread (%.4f ns per nz*column aggregate); whole-net figure scaled by that ratio' % (P, par, serial, 1e9 * par / macs))
    except Exception as e:       # a reported-only baseline must never break the bench line
        res['all_cores'] = dict(value=None, error=str(e))
    finally:
        _CPU_JOBS = None
    # (iii) the reference's TiledMatrix.torchdot rebuilds the CSR on EVERY call (keynet/sparse.py:610): tocsr() + dot of one tiled layer
    tiled = [(n, c) for (n, c) in layers if type(c.W) is ksp.TiledMatrix]
    if tiled:
        (n, c) = tiled[len(tiled) // 2]
        X = rng.randn(c.W.shape[1], 32).astype(np.float32)
        t0 = time.perf_counter()
        M = c.W.tocsr()
        t1 = time.perf_counter()
        M.dot(X)
        t2 = time.perf_counter()
        res['tocsr_per_call'] = dict(layer=n, nnz=int(M.nnz), tocsr_seconds=t1 - t0, dot_seconds=t2 - t1, columns=32,
                                     note='tile expansion here is this build\'s vectorised host restatement; the reference walks the blocks in Python (slower)')
    del jobs, measured, bands
    return res


# ----------------------------------------------------------------------------------------------------------------------------
# GPU side
def _takes_small_k_kernel(W, batch):
    """Mirror of the dispatch in kn_conv.hip (convtaps_spmm): one output pixel's whole contraction fits 28 rows (VGG conv1_1),
    which runs in the write-bound convtaps_smallk_kernel and is therefore not part of the MFMA roofline aggregate."""
    t = getattr(W, '_taps', None)
    if t is None or batch % 256:
        return False
    return int(np.bincount(t['ent_out']).max()) * W._inshape[0] + (1 if t['lastcol'] is not None else 0) <= 28


def layer_table(knet, batch):
    """Per keyed layer: the kernel family that runs it in the key-net's CURRENT mode, algorithmic MACs (= nnz of the expanded
    operator the reference applies) and bytes."""
    rows = []
    children = list(knet._keynet.named_children())
    for (i, (name, c)) in enumerate(children):
        if not isinstance(c, KeyedLayer):
            continue
        contract = getattr(c, '_exact', True)
        exact = contract is True or contract == 'auto'
        if type(c.W) is ksp.SparseMatrix and not exact and c.W._dense_device_op() is not None:
            kind = 'dense'                     # keyed nn.Linear on the split-K MFMA path (tolerance mode)
            (r, cdim) = c.W.shape
            nnz_exp = int(c.W.nnz())
            wbytes = 4 * r * cdim
        elif isinstance(c.W, ksp.Conv2dTiledMatrix):
            op = c.W._device_op()
            (r, cdim) = op.shape()
            nnz_exp = op.nnz_expanded()
            kind = 'convexact' if exact else ('convsplit' if contract == 'split' else ('smallk' if _takes_small_k_kernel(c.W, batch) else 'convtaps'))
            wbytes = 4 * c.W.nnz()             # taps + entries + last column actually read
        elif isinstance(c.W, ksp.FactoredSparseMatrix):
            # an untiled keyed conv whose stored CSR is provably the expansion of its factored form: runs the order-preserving conv pipeline from
            # the taps (sparse.py: FactoredSparseMatrix); algorithmic MACs = the stored non-zeros of the reference's CSR
            op = c.W._device_op()
            (r, cdim) = op.shape()
            nnz_exp = int(c.W.nnz())
            kind = 'convexact'
            wbytes = 4 * c.W._factored.nnz()   # taps + entries + last column actually read
        else:
            op = c.W._device_op()
            (r, cdim) = op.shape()
            nnz_exp = op.nnz_expanded()
            kind = 'csr'
            wbytes = 8 * nnz_exp               # (col,val) per non-zero
        flags = (1 if ((i + 1 < len(children)) and isinstance(children[i + 1][1], torch.nn.ReLU)) or c.iskeyedrelu() else 0) | (2 if (exact and kind != 'dense') else 0) | (4 if contract == 'bf16x3' else 0)
        plan = (c.W._dense_device_op() if kind == 'dense' else op).plan(batch, flags)
        if 'bf16x3' in plan:
            kind = 'convbf16x3'
        flops_exec = None
        if kind == 'convsplit':                # the split application (Conv2dTiledMatrix._split_ops): spatial CSR per input channel, then an ntaps-slot conv-taps operator
            (opK, op2) = c.W._split_ops()
            t = c.W._taps
            plan = '%d x [%s]; %s' % (c.W._inshape[0], opK.plan(batch, 2), op2.plan(batch, flags & 1))
            flops_exec = 2.0 * batch * (len(t['ent_out']) * c.W._inshape[0] + len(t['taps']) * c.W._outshape[1] * c.W._outshape[2] * c.W._inshape[0] * c.W._outshape[0])
        rows.append(dict(name=name, kind=kind, rows=r, cols=cdim, nnz=nnz_exp, flops=2.0 * nnz_exp * batch, flops_executed=flops_exec,
                         bytes=float(wbytes) + 4.0 * batch * (r + cdim), layer=c, plan=plan,
                         fuse=(i + 1 < len(children)) and isinstance(children[i + 1][1], torch.nn.ReLU)))
    return rows


def time_layers(x_cipher, table, iters, reps=3):
    """Per-layer kernel time with HIP events on the launch stream (torch's current stream is the one kn_spmm launches on).
    In the forward the kernels run back to back, so each timing is over `reps` back-to-back launches behind one untimed launch:
    a launch that follows an idle period runs 1-20 % slower while the clock ramps back up (measured with per-workgroup time
    stamps, profiles/r02_workgroup_timeline_conv_layers.txt), which is not what happens inside the timed step.  The MEDIAN over
    `iters` such timings is kept: a multi-GB output allocation can occasionally fall out of the caching allocator and cost tens
    of ms, which must not leak into a kernel's average."""
    y = x_cipher
    for row in table:
        c = row['layer']
        xin = y
        y = None
        out = c.forward(xin, fuse_relu=row['fuse'])       # warm
        torch.cuda.synchronize()
        times = []
        for _ in range(max(iters, 1)):
            del out
            out = c.forward(xin, fuse_relu=row['fuse'])
            (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            e0.record()
            for _ in range(reps):
                del out
                out = c.forward(xin, fuse_relu=row['fuse'])
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / reps)
        row['ms'] = float(np.median(times))
        row['ms_min'] = float(np.min(times))
        y = out
    return table


def kernel_sources_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'keynet_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()


def committed_traffic(workload, mode):
    """HBM bytes per forward of the dominant kernel from the committed PMC passes (profiles/rNN_<workload>_*traffic.json:
    separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same bench, FETCH doubled per the guide's gfx950
    note).  bench.py cannot collect PMC counters on itself, so the figure is quoted only when that pass was taken in the same
    mode on byte-identical kernel sources (`csrc_sha256` recorded by tools/make_profiles.py); otherwise (None, reason)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_%s_*traffic.json' % workload)))
    if not files:
        return (None, 'no committed PMC pass')
    t = json.load(open(files[-1]))
    rel = os.path.relpath(files[-1], ROOT)
    if t.get('mode', 'tolerance') != mode:
        return (None, '%s is a %s-mode pass' % (rel, t.get('mode', 'tolerance')))
    if t.get('csrc_sha256') != kernel_sources_sha():
        return (None, '%s was taken on other kernel sources' % rel)
    return (t.get('convtaps_hbm_bytes_per_forward', t.get('dominant_hbm_bytes_per_forward')), rel)


def roofline_of(table, workload, batch, mode):
    """Roofline record of the dominant kernel family of `table` (the layers as they ran in this mode)."""
    kinds = {}
    for r in table:
        kinds.setdefault(r['kind'], []).append(r)
    by_ms = sorted(kinds.items(), key=lambda kv: -sum(r['ms'] for r in kv[1]))
    (kind, dom) = by_ms[0]
    dom_ms = sum(r['ms'] for r in dom)
    if kind == 'convbf16x3':
        peak = PEAK_F32_MFMA_TFLOPS * 16.0 / 6.0
        ach = sum(r['flops'] for r in dom) / dom_ms / 1e9
        return dict(bound='mfma', kernel='convtaps_bf16x3_kernel (%d launches/forward)' % len(dom), achieved=ach, peak=peak, unit='TFLOP/s (f32-equivalent)', frac=ach / peak, traffic=None,
                    algorithmic_flops=sum(r['flops'] for r in dom), algorithmic_bytes=sum(r['bytes'] for r in dom), ms_per_forward=dom_ms,
                    note='peak = six v_mfma_f32_32x32x16_bf16 per f32 product block at 16x the f32-input MFMA rate: 157.3 * 16 / 6')
    if kind == 'convsplit':
        ach = sum(r['flops_executed'] for r in dom) / dom_ms / 1e9
        return dict(bound='mfma', kernel='split application of filled-in conv layers: spatial CSR kernels per input channel + convtaps_mfma_kernel (%d layers/forward)' % len(dom), achieved=ach,
                    peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s (executed)', frac=ach / PEAK_F32_MFMA_TFLOPS, traffic=None, algorithmic_flops=sum(r['flops'] for r in dom),
                    executed_flops=sum(r['flops_executed'] for r in dom), algorithmic_bytes=sum(r['bytes'] for r in dom), ms_per_forward=dom_ms,
                    note='algorithmic_flops = the stored entries of the fused operator the reference applies; the split application executes executed_flops for the same product')
    if kind in ('convtaps', 'dense'):
        dom = kinds.get('convtaps', []) or dom
        dom_ms = sum(r['ms'] for r in dom)
        ach = sum(r['flops'] for r in dom) / dom_ms / 1e9
        (traffic, tsrc) = committed_traffic(workload, mode) if batch == 256 else (None, 'PMC pass is for 256 images')
        return dict(bound='mfma', kernel='convtaps_mfma_kernel (%d launches/forward)' % len(dom), achieved=ach, peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s',
                    frac=ach / PEAK_F32_MFMA_TFLOPS, traffic=traffic, traffic_unit='bytes/forward (PMC, offline pass)', traffic_source=tsrc,
                    algorithmic_bytes=sum(r['bytes'] for r in dom), algorithmic_flops=sum(r['flops'] for r in dom), ms_per_forward=dom_ms)
    macs = sum(r['nnz'] for r in dom) * float(batch)
    intensity = 2.0 * macs / sum(r['bytes'] for r in dom)
    mains = []                                            # the first kernel of each layer's plan (kn_spmm_plan lists the main launch first, guards / last-row helpers behind it)
    for r in dom:
        k = [w.split('<')[0] for w in str(r.get('plan', '')).replace(',', ' ').split() if w.split('<')[0].endswith('_kernel')]
        if k and k[0] not in mains:
            mains.append(k[0])
    names = ' / '.join(mains) or \
        {'convexact': 'convtaps_exact_pipe_kernel / convtaps_exact_kernel', 'csr': 'csr_group_kernel / csr_rows_kernel', 'smallk': 'convtaps_smallk_kernel'}[kind]
    if kind == 'smallk' or intensity < 2.0 * PEAK_VALU_NOFMA_TMACS * 1e3 / PEAK_HBM_GBS:      # below the balance point of the no-FMA VALU roof: HBM-bound
        ach = sum(r['bytes'] for r in dom) / dom_ms / 1e6
        return dict(bound='hbm', kernel='%s (%d launches/forward)' % (names, len(dom)), achieved=ach, peak=PEAK_HBM_GBS, unit='GB/s', frac=ach / PEAK_HBM_GBS,
                    traffic=None, algorithmic_bytes=sum(r['bytes'] for r in dom), ms_per_forward=dom_ms)
    ach = macs / dom_ms / 1e9
    (traffic, tsrc) = committed_traffic(workload, mode)
    return dict(bound='valu-nofma', kernel='%s (%d launches/forward)' % (names, len(dom)), achieved=ach, peak=PEAK_VALU_NOFMA_TMACS, unit='T MAC/s',
                frac=ach / PEAK_VALU_NOFMA_TMACS, traffic=traffic, traffic_unit='bytes/forward (PMC, offline pass)', traffic_source=tsrc, algorithmic_macs=macs,
                algorithmic_bytes=sum(r['bytes'] for r in dom), ms_per_forward=dom_ms,
                note='bit-exact contract: a separately rounded f32 product and an f32 add per stored value, in the reference\'s order -- no fused multiply-add, no '
                     'accumulating matrix instruction; roof = one product + one add per lane per 2 cycles = 157.3 TFLOP/s / 4 (kernels that take their products '
                     'from K = 1 matrix instructions with a zero accumulator still pay the adds on the same lanes: DESIGN.md section 8)')


def exact_parity(knet, x_cipher, n_img=8, n_pix=4, layers=('conv1_1', 'conv1_2', 'pool3_3', 'conv4_2', 'conv5_2', 'fc6')):
    """Checker for the exact leg: the order-preserving kernels AS TIMED -- launched on the whole batch -- on one real operator of each kernel family
    (first-layer conv, 64- and 512-channel conv pipelines, a keyed pooling layer = loose CSR rows, a keyed Linear = one big pattern group) against
    the CPU oracle (oracle/: scipy csr_matvecs restated) on sampled output rows, the first `n_img` batch columns, bit for bit, chained layer to
    layer with the key-net's own activations as input."""
    import oracle
    import scipy.sparse
    rng = np.random.RandomState(1)
    y = x_cipher
    checked = []
    children = list(knet._keynet.named_children())
    for (i, (name, c)) in enumerate(children):
        if not isinstance(c, KeyedLayer):
            continue
        fuse = (i + 1 < len(children)) and isinstance(children[i + 1][1], torch.nn.ReLU)
        out = c.forward(y, fuse_relu=fuse)
        if name in layers:
            W = c.W
            xh = y.t()[:, :n_img].contiguous().cpu().numpy()
            if isinstance(W, ksp.Conv2dTiledMatrix) and W._taps is not None:
                (Cout, Hout, Wout) = W._outshape
                pix = np.sort(rng.choice(Hout * Wout, size=n_pix, replace=False))
                M = W.rows_csr(pix)
                rows = (np.arange(Cout)[:, None] * Hout * Wout + pix[None, :]).ravel()
            else:
                full = W.tocsr() if isinstance(W, ksp.TiledMatrix) else W._matrix.tocsr()
                rows = np.unique(np.concatenate((rng.choice(full.shape[0] - 1, size=min(300, full.shape[0] - 1), replace=False), [full.shape[0] - 1])))
                if isinstance(W, ksp.TiledMatrix):
                    M = full[rows]
                else:                                             # stored (unsorted) order of the keyed Linear's rows, untouched
                    (ip, ix, dt) = (full.indptr, full.indices, full.data)
                    sel = np.concatenate([np.arange(ip[r], ip[r + 1]) for r in rows])
                    M = scipy.sparse.csr_matrix((dt[sel], ix[sel], np.concatenate(([0], np.cumsum(ip[rows + 1] - ip[rows])))), shape=(len(rows), full.shape[1]))
            ref = oracle.csr_matvecs(M.shape, M.indptr, M.indices, M.data.astype(np.float32), xh)
            if fuse:
                ref = np.maximum(ref, 0)
            got = out.t()[torch.as_tensor(rows, device=out.device)][:, :n_img].cpu().numpy()
            with torch.cuda.device(out.device):
                plan = W._device_op(out.device).plan(int(y.shape[0]), 2 | (1 if fuse else 0)).split(' grid=')[0] if hasattr(W, '_device_op') else ''
            checked.append({'layer': name, 'rows': int(len(rows)), 'images': n_img, 'batch_columns_launched': int(y.shape[0]), 'kernel': plan[:80],
                            'bit_equal': bool(np.array_equal(got, ref))})
        y = out
        if name == layers[-1]:
            break
    return {'check': 'exact-mode kernels, launched on the whole batch, vs the CPU oracle (scipy csr_matvecs restated) on sampled output rows of real layers', 'layers': checked,
            'ok': bool(checked) and all(r['bit_equal'] for r in checked)}


def float_key_parity(dev, batch=256):
    """Float-key family on a VGG-16 slice (the same 21-layer topology at width 8 on 32x32 inputs, keyed by TiledOrthogonalKeynet:
    hierarchical permutation + block Givens rotations + affine photometric keys, gamma = 100).  The order-preserving path is bit-exact
    with the reference's scipy arithmetic (tests/test_parity_gpu.py), so it stands in for the reference here.  Two records:
      contract   the key-net under its DEFAULT contract ('auto'): per conv layer, the shipped forward's output against the exact path on
                 the same input -- `ok` = every layer within 1e-5 * max(1, |y|), unconditioned; `layers_switched_to_exact` = the layers the
                 calibration moved off the matrix cores to get there;
      forced_mfma  the same layers forced onto the matrix cores (exact_mode(False)): how far a re-ordered f32 evaluation lands."""
    import warnings
    t0 = time.time()
    torch.manual_seed(0)
    net = VGG16(num_classes=10, width=8, fc_width=64, insize=32).eval()
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        (sensor, knet) = ksys.TiledOrthogonalKeynet((3, 32, 32), net, 8)
    g = torch.Generator(device=dev).manual_seed(77)
    x = torch.randn((batch, 3, 32, 32), generator=g, device=dev)
    xc = sensor.fromtensor(x).encrypt().astensor()
    import logging
    logging.getLogger('keynet_amd').setLevel(logging.ERROR)          # the switches are reported below, not as log lines
    la = knet.forward_linear(xc)[:, :-1]                              # calibrates every layer
    logging.getLogger('keynet_amd').setLevel(logging.WARNING)
    rep = knet.contract_report()

    def per_layer(force_mfma):
        rows = []
        y = xc
        children = list(knet._keynet.named_children())
        for (i, (name, c)) in enumerate(children):
            if not isinstance(c, KeyedLayer):
                continue
            fuse = (i + 1 < len(children)) and isinstance(children[i + 1][1], torch.nn.ReLU)
            if isinstance(c.W, ksp.Conv2dTiledMatrix):
                xt = y.t()
                ye = c.W.torchdot(xt, relu=fuse, exact=True)
                ys = c.W.torchdot(xt, relu=fuse, exact=False) if force_mfma else c.forward(y, fuse_relu=fuse).t()
                (d, m) = (float((ye - ys).abs().max()), float(ye.abs().max()))
                rows.append({'layer': name, 'max_abs_diff': d, 'max_abs_out': m, 'within_1e-5': bool(d <= 1e-5 * max(1.0, m)), 'ran': 'mfma' if (force_mfma or c._exact is False) else 'exact'})
                y = ye.t()
            else:
                y = c.forward(y, fuse_relu=fuse)
        return rows
    rows_auto = per_layer(False)
    rows_mfma = per_layer(True)
    knet.exact_mode(True)
    le = knet.forward_linear(xc)[:, :-1]
    knet.exact_mode(False)
    lm = knet.forward_linear(xc)[:, :-1]
    with torch.no_grad():
        lp = net(x.cpu()).reshape(batch, -1)
    return {'net': 'TiledOrthogonalKeynet VGG16 slice (width 8, 3x32x32, tile 8), %d images' % batch,
            'contract': {'tolerance': 1e-5, 'layers': rows_auto, 'ok': bool(all(r['within_1e-5'] for r in rows_auto)), 'layers_switched_to_exact': rep['switched'],
                         'worst_layer_abs_diff': max(r['max_abs_diff'] for r in rows_auto),
                         'logits_max_abs_diff_vs_exact': float((le - la).abs().max())},
            'ok': bool(all(r['within_1e-5'] for r in rows_auto)), 'layers_switched_to_exact': rep['switched'],
            'forced_mfma': {'layers': rows_mfma, 'worst_layer_abs_diff_mfma_vs_exact': max(r['max_abs_diff'] for r in rows_mfma),
                            'logits_max_abs_diff_mfma_vs_exact': float((le - lm).abs().max())},
            'logits_max_abs': float(le.abs().max()),
            'logits_max_abs_err_exact_vs_source_network': float((le.cpu() - lp).abs().max()),
            'logits_max_abs_err_mfma_vs_source_network': float((lm.cpu() - lp).abs().max()), 'seconds': time.time() - t0}


def oracle_parity_csr(knet, x_cipher, logits, n_img=8):
    """Checker for the untiled (permutation) key-nets: the CPU oracle (oracle/: scipy csr_matvecs restated) recomputes the first images
    through EVERY layer of the same stored-order operators; the device logits of the timed batch must equal them bit for bit."""
    import oracle
    t0 = time.time()
    yo = np.ascontiguousarray(x_cipher[:n_img].cpu().numpy().T)                     # [D0+1, n] feature-major
    children = list(knet._keynet.named_children())
    i = 0
    while i < len(children):
        (name, c) = children[i]
        if not isinstance(c, KeyedLayer) or not isinstance(c.W, ksp.SparseMatrix) or isinstance(c.W, ksp.TiledMatrix):
            return {'check': 'CPU oracle on every layer', 'ok': None, 'skipped': 'layer %s is not a plain stored-order CSR operator' % name}
        fuse = (i + 1 < len(children)) and isinstance(children[i + 1][1], torch.nn.ReLU)
        (ip, ix, dt) = ksp._stored_order_csr(c.W._matrix)
        yo = oracle.csr_matvecs(c.W.shape, ip, ix, dt, yo)
        if fuse or c.iskeyedrelu():
            yo = np.maximum(yo, 0)
        i += 2 if fuse else 1
    got = logits[:n_img].contiguous().cpu().numpy()
    eq = bool(np.array_equal(got, yo.T[:, :-1]))
    return {'check': 'logits of the timed batch vs the CPU oracle (scipy csr_matvecs restated) run through every layer on the first %d images' % n_img,
            'bit_equal': eq, 'ok': eq, 'images': n_img, 'seconds': time.time() - t0}


def run_secondary(args):
    """BASELINE configs[1] and [2] in the driver's line: LeNet_AvgPool B=1024 and AllConvNet B=4096 each run as a CHILD process of this
    bench (its own host phase, scipy baseline, device phase, oracle parity) BEFORE this process touches the GPU; the child's JSON line is
    condensed into `secondary`.  (A child process, not an exec: the parent goes on to the VGG legs.)"""
    out = {}
    # (LeNet: a forward is 37 us -- three warm-up steps are 0.1 ms, not enough for the GPU to leave its idle clock: 2 000 warm-up steps = 75 ms)
    for (wl, steps, warm, extra) in (('lenet', max(args.steps, 200), max(args.warmup, 2000), ['--graph-leg']), ('allconv', max(args.steps, 10), max(args.warmup, 3), [])):
        t0 = time.time()
        child_detail = os.path.join(ROOT, 'bench_detail_%s.json' % wl)
        cmd = [sys.executable, os.path.abspath(__file__), '--workload', wl, '--steps', str(steps), '--warmup', str(warm), '--layer-iters', '3',
               '--no-secondary', '--cpu-budget', '8', '--detail', child_detail] + extra
        env = dict(os.environ)
        for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
            env.pop(k, None)
        try:
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
            lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
            for l in p.stderr.splitlines():
                if l.startswith('[bench'):
                    log('  [%s] %s' % (wl, l))
            if p.returncode != 0 or len(lines) != 1:
                out[wl] = {'error': 'child exited with %d' % p.returncode, 'stderr_tail': p.stderr[-800:]}
                continue
            r = json.loads(lines[0])                      # the child's compact line; its full record is in its own detail file
            cpu = r.get('cpu_baseline') or {}
            out[wl] = {'workload': r['config']['workload'], 'images_per_gpu': r['config']['images_per_gpu'], 'images_per_s': r['value'], 'ms_per_step': r['ms_per_step'],
                       'steps': r['steps'], 'warmup': r['warmup'], 'roofline': r['roofline'],
                       'parity': {'bit_equal': (r.get('parity') or {}).get('oracle_bit_equal'), 'check': 'logits of the timed batch vs the CPU oracle run through every layer on the first 8 images'},
                       'parity_vs_source_network': _pick(r.get('parity') or {}, ('ok', 'max_abs_err', 'atol')),
                       'cpu_baseline': cpu, 'detail': r.get('detail'), 'child_wall_s': time.time() - t0}
            try:
                out[wl]['full'] = json.load(open(child_detail))
            except Exception:
                pass
        except Exception as e:      # a reported-only section must never break the headline
            out[wl] = {'error': str(e)}
    return out


def end_to_end(sensor, knet, x_plain, steps, warmup):
    """Plaintext -> logits (keynet/system.py:250-255 + 130-133): sensor.fromtensor(x).encrypt() -- homogenise on the device
    (kn_affine_to_linear) and apply the image key (the same SpMM primitive) -- inside the timed loop, then the keyed forward."""
    def step():
        return knet.forward_linear(sensor.fromtensor(x_plain).encrypt().astensor())
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        y = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    e0.record()
    for _ in range(5):
        xc = sensor.fromtensor(x_plain).encrypt().astensor()
    e1.record()
    torch.cuda.synchronize()
    del y, xc
    n = x_plain.shape[0]
    return {'images_per_s': n * steps / el, 'ms_per_step': 1e3 * el / steps, 'steps': steps, 'encrypt_ms': e0.elapsed_time(e1) / 5,
            'what': 'sensor.fromtensor(x_plain).encrypt() + forward_linear per step, plaintext batch resident in HBM'}


def collective_record(knet, sensor, x_cipher, gathered, batch, world, rank, local_rank, dev, inshape, share):
    """What the N>1 line says about itself (every rank takes part; rank 0 keeps the record): the ranks and devices that were really
    there, the cost of the logits all-gather alone (HIP events on the launch stream), and two bit-level checks of the gathered block --
    every rank's own shard against its local forward, and the LAST rank's shard recomputed on rank 0 from that rank's input seed
    (weights are replicated and batch columns independent, so a single process must reproduce any shard bit for bit)."""
    info = {'rank': rank, 'local_rank': local_rank, 'device_index': dev.index, 'device_name': torch.cuda.get_device_name(dev), 'pid': os.getpid()}
    infos = [None] * world
    dist.all_gather_object(infos, info)
    yl = knet.forward_linear(x_cipher)[:, :-1].contiguous()
    for _ in range(3):
        kdist.gather_logits(yl, total=batch * world)
    torch.cuda.synchronize()
    dist.barrier()
    n_calls = 20
    (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n_calls):
        g = kdist.gather_logits(yl, total=batch * world)
    e1.record()
    torch.cuda.synchronize()
    wall_ms = 1e3 * (time.perf_counter() - t0) / n_calls
    ev_ms = e0.elapsed_time(e1) / n_calls
    own = bool(torch.equal(g[rank * batch:(rank + 1) * batch], yl)) and bool(torch.equal(gathered[rank * batch:(rank + 1) * batch], yl))
    flag = torch.tensor([1 if own else 0], dtype=torch.int32, device=torch.device('cpu') if share else dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    peer = world - 1
    peer_equal = None
    if rank == 0:
        gp = torch.Generator(device=dev).manual_seed(1234 + peer)
        xp = torch.randn((batch,) + tuple(inshape), generator=gp, device=dev)
        yp = knet.forward_linear(sensor.fromtensor(xp).encrypt().astensor())[:, :-1]
        peer_equal = bool(torch.equal(g[peer * batch:(peer + 1) * batch], yp))
        del xp, yp
    return {'backend': dist.get_backend(), 'ranks_seen': dist.get_world_size(), 'ranks': infos,
            'op': 'all_gather_into_tensor of [%d, %d] f32 logits per rank' % (batch, yl.shape[1]), 'bytes_per_rank': int(yl.numel() * 4),
            'ms_per_call': ev_ms, 'ms_per_call_wall': wall_ms, 'calls_timed': n_calls,
            'every_rank_shard_bit_equal_to_its_local_forward': bool(flag.item() == 1), 'rank0_shard_bit_equal': own if rank == 0 else None,
            'rank0_shard_sha256': hashlib.sha256(yl.cpu().numpy().tobytes()).hexdigest() if rank == 0 else None,
            'peer_shard_recomputed_on_rank0': {'peer_rank': peer, 'bit_equal': peer_equal}}


# ----------------------------------------------------------------------------------------------------------------------------
# the ONE stdout line: compact (< 4 KB) so that a driver keeping a bounded stdout tail always sees the whole record; everything else
# (per-layer tables, plans, contract evidence, child lines, experimental legs) goes to bench_detail.json and to stderr
LINE_LIMIT = 4096
DETAIL_FILE = 'bench_detail.json'


def _num(v, sig=6):
    """Floats to `sig` significant digits (the line is a summary; bench_detail.json keeps full precision)."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        return float('%.*g' % (sig, v)) if np.isfinite(v) else None
    if isinstance(v, dict):
        return {k: _num(x, sig) for (k, x) in v.items()}
    if isinstance(v, (list, tuple)):
        return [_num(x, sig) for x in v]
    return v


def _pick(d, keys):
    return {k: d.get(k) for k in keys if isinstance(d, dict) and k in d}


def _clip(s, n):
    return s if (not isinstance(s, str) or len(s) <= n) else s[:n - 3] + '...'


def _compact_roofline(r):
    if not isinstance(r, dict):
        return r
    out = _pick(r, ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_bytes', 'algorithmic_flops', 'algorithmic_macs', 'ms_per_forward'))
    out['kernel'] = _clip(out.get('kernel'), 96)
    return out


def _compact_cpu(c):
    if not isinstance(c, dict):
        return c
    out = _pick(c, ('value', 'unit', 'cores', 'kind', 'engine', 'host'))
    out['engine'] = _clip(out.get('engine'), 72)
    if isinstance(c.get('all_cores'), dict):
        out['all_cores'] = _pick(c['all_cores'], ('value', 'cores'))
    out['sample'] = _clip(c.get('sample_short') or c.get('sample'), 200)
    return out


def _compact_secondary(s):
    if not isinstance(s, dict):
        return s
    if 'error' in s:
        return {'error': _clip(str(s['error']), 120)}
    roof = s.get('roofline') or {}
    par = s.get('parity') or {}
    cpu = s.get('cpu_baseline') or {}
    return {'images_per_gpu': s.get('images_per_gpu'), 'images_per_s': s.get('images_per_s'), 'ms_per_step': s.get('ms_per_step'), 'bound': roof.get('bound'), 'achieved': roof.get('achieved'),
            'peak': roof.get('peak'), 'unit': roof.get('unit'), 'frac': roof.get('frac'), 'kernel_ms': roof.get('ms_per_forward'), 'bit_equal': par.get('bit_equal'),
            'cpu_images_per_s': cpu.get('value'), 'cpu_cores': cpu.get('cores')}


def _compact_collective(c):
    if not isinstance(c, dict):
        return c
    out = _pick(c, ('backend', 'ranks_seen', 'bytes_per_rank', 'ms_per_call', 'every_rank_shard_bit_equal_to_its_local_forward'))
    out['ranks'] = [[r.get('rank'), r.get('device_index')] for r in (c.get('ranks') or []) if isinstance(r, dict)]      # [rank, device_index] per rank
    out['peer_shard_recomputed_on_rank0'] = c.get('peer_shard_recomputed_on_rank0')
    return out


def compact_record(res, detail_path=DETAIL_FILE):
    """The driver's line from the full record `res` (which is written to `detail_path`).  Keys and order follow the bench contract; every
    nested object is cut to the fields a reader needs to check the number (the rest is in the detail file, whose path the line carries)."""
    cfg = res.get('config') or {}
    line = {k: res.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')}
    line['metric'] = _clip(line['metric'], 120)
    line['config'] = _pick(cfg, ('workload', 'mode', 'headline_contract', 'default_contract_value', 'images_per_gpu', 'global_batch', 'nnz_per_image', 'parallelism'))
    line['config']['workload'] = _clip(line['config'].get('workload'), 160)
    line['config']['mode'] = _clip(line['config'].get('mode'), 160)
    line['roofline'] = _compact_roofline(res.get('roofline'))
    line['cpu_baseline'] = _compact_cpu(res.get('cpu_baseline')) if isinstance(res.get('cpu_baseline'), dict) else _clip(res.get('cpu_baseline'), 120)
    par = res.get('parity') or {}
    line['parity'] = _pick(par, ('ok', 'max_abs_err', 'atol'))
    if isinstance(res.get('oracle_parity'), dict):
        line['parity']['oracle_bit_equal'] = res['oracle_parity'].get('bit_equal')
    ex = res.get('exact')
    if isinstance(ex, dict):
        if 'error' in ex:
            line['exact'] = {'error': _clip(str(ex['error']), 120)}
        else:
            roof = ex.get('roofline') or {}
            epar = ex.get('parity') or {}
            line['exact'] = {'images_per_s': ex.get('images_per_s'), 'ms_per_step': ex.get('ms_per_step'), 'frac': roof.get('frac'), 'peak': roof.get('peak'), 'unit': roof.get('unit'),
                             'bit_equal': epar.get('ok'), 'oracle_checked_layers': [r.get('layer') for r in (epar.get('layers') or [])]}
    if isinstance(res.get('secondary'), dict):
        line['secondary'] = {k: _compact_secondary(v) for (k, v) in res['secondary'].items()}
    if isinstance(res.get('contract'), dict):
        line['contract'] = {'tolerance': res['contract'].get('tolerance'), 'layers_switched_to_exact': res['contract'].get('layers_switched_to_exact'),
                            'rescreened_every_forward': res['contract'].get('rescreened_every_forward')}
    if isinstance(res.get('end_to_end'), dict):
        line['end_to_end'] = _pick(res['end_to_end'], ('images_per_s', 'ms_per_step', 'encrypt_ms', 'error'))
    if isinstance(res.get('exact_layers_parity'), dict):
        line['exact_layers_parity'] = {'bit_equal': res['exact_layers_parity'].get('ok'), 'oracle_checked_layers': [r.get('layer') for r in (res['exact_layers_parity'].get('layers') or [])]}
    if res.get('collective') is not None:
        line['collective'] = _compact_collective(res['collective'])
    if res.get('errors'):
        line['errors'] = {k: _clip(str(v), 100) for (k, v) in list(res['errors'].items())[:6]}
    line['detail'] = detail_path
    line = _num(line)
    s = json.dumps(line, separators=(',', ':'))
    # belt and braces: if a pathological string still pushes the line over the limit, drop optional sections until it fits
    for k in ('end_to_end', 'exact_layers_parity', 'contract', 'secondary', 'exact', 'errors'):
        if len(s) < LINE_LIMIT:
            break
        line.pop(k, None)
        s = json.dumps(line, separators=(',', ':'))
    assert len(s) < LINE_LIMIT, 'bench line is %d chars' % len(s)
    return s


def write_detail(res, path=None):
    """Full record next to bench.py (and under gpurun_out/ when that scratch directory exists, so that a GPU-box run brings it home)."""
    paths = [path or os.path.join(ROOT, DETAIL_FILE)]
    if path is None and os.path.isdir(os.path.join(ROOT, 'gpurun_out')):
        paths.append(os.path.join(ROOT, 'gpurun_out', DETAIL_FILE))
    for p in paths:
        try:
            with open(p, 'w') as f:
                json.dump(res, f, indent=1, default=str)
        except OSError as e:
            log('[bench] could not write %s: %s' % (p, e))



def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (torch.distributed.run, one per GPU) from THIS
    process, which has not touched the GPU (no torch.cuda call above this point), and exit with their code."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    log('[bench] starting %d ranks: %s' % (args.gpus, ' '.join(cmd)))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--exact-layers-parity', action='store_true',
                    help='float-key workloads (vgg16-*): check the layers the contract keeps in the reference\'s order against the CPU oracle on one sampled output pixel each (OPT-IN: the host '
                         'expansion of a filled-in pixel is 40-90 M stored entries per layer -- minutes of host time at full size; the same check runs on reduced nets in tests/test_vgg16_families_gpu.py)')
    ap.add_argument('--workload', default='vgg16', choices=['vgg16', 'vgg16-gain', 'vgg16-givens', 'vgg16-givens28', 'vgg16-stochastic', 'lenet', 'allconv'])
    ap.add_argument('--batch', type=int, default=None, help='images per GPU (default: the BASELINE config)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--layer-iters', type=int, default=5)
    ap.add_argument('--exact', action='store_true', help='bit-exact mode for the tiled key-nets as the MAIN measurement (order-preserving kernels everywhere)')
    ap.add_argument('--no-exact-leg', action='store_true', help='skip the additional bit-exact-mode measurement of the default vgg16 run')
    ap.add_argument('--graph', action='store_true', help='replay the forward from a captured HIP graph (launch-bound small nets)')
    ap.add_argument('--dist', action='store_true', help='initialise the RCCL process group and all-gather the logits every step even with ONE rank '
                                                        '(exercises the multi-GPU code path on a single-GPU box)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the LeNet / AllConvNet legs of the default run')
    ap.add_argument('--graph-leg', action='store_true', help='additionally time the forward replayed from a captured HIP graph (reported as `graph`)')
    ap.add_argument('--cpu-budget', type=float, default=24.0, help='seconds of scipy work for the CPU baseline sample')
    ap.add_argument('--experimental', action='store_true', help='additionally run the EXPERIMENTAL bf16x3 leg (never the headline; detail file only)')
    ap.add_argument('--detail', default=None, metavar='FILE', help='where the full record goes (default: bench_detail.json next to bench.py); stdout carries the compact line')
    ap.add_argument('--trace-layers', default=None, metavar='FILE', help='profiling aid (run under rocprofv3 --kernel-trace): after the first forward, launch every layer '
                                                                          '8 times back to back with a marker kernel between layers, write the layer list (name, kind, flops, bytes) to FILE and exit; '
                                                                          'tools/trace_layers.py joins it with the kernel trace into a per-layer table')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))                       # before ANY GPU call in this process
    # stdout carries exactly ONE line, the JSON record: native libraries (RCCL prints a version banner to fd 1) go to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    assert args.gpus == world, '--gpus %d but WORLD_SIZE=%d' % (args.gpus, world)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))

    secondary = None
    if args.workload == 'vgg16' and world == 1 and not args.no_secondary and not args.exact and not args.dist and args.batch is None:
        secondary = run_secondary(args)                   # children own the GPU one after the other; this process has not touched it yet

    # ---- host phase: keying and the CPU baseline, nothing below touches the GPU until "device phase" ----------------------
    # Arithmetic contract of the headline: BASELINE configs[3] names "MFMA dense sub-tiles", so the tiled VGG-16 key-nets are built with the
    # 'auto' contract EXPLICITLY (matrix cores wherever the 1e-5 tolerance holds, screened on every forward); a permutation-only tiled key-net's
    # own default is the bit-exact contract, which the `exact` leg of the same line measures on the same key-net.
    (sensor, knet, inshape, batch, desc, net) = build_workload_shared(args.workload, rank, world, exact=True if args.exact else ('auto' if args.workload.startswith('vgg16') else None))
    mode = 'exact' if (args.exact or not args.workload.startswith('vgg16')) else 'tolerance'
    mode_desc = {'exact': 'exact: order-preserving kernels, bit-exact with the reference (the default of permutation-only key-nets)',
                 'tolerance': "tolerance, explicit opt-in exact='auto': f32 MFMA within 1e-5 max(1,|y|) of the reference, re-screened every forward; bit-exact contract = `exact` leg"}[mode]
    if args.exact:
        desc += ' [exact mode: order-preserving kernels, bit-exact with the reference algorithm]'
    batch = args.batch if args.batch is not None else batch
    cpu = None
    if world == 1 and not args.no_cpu_baseline and args.workload in ('vgg16', 'lenet', 'allconv'):      # (the scipy baseline is measured on the configs BASELINE.json quotes)
        t0 = time.time()
        cpu = cpu_baseline(knet, args.workload, budget_s=args.cpu_budget)
        log('[bench cpu] baseline section took %.1f s' % (time.time() - t0))

    # ---- device phase --------------------------------------------------------------------------------------------------------
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    # Test-only override (single-GPU boxes): KN_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo, so the N>1
    # control flow (sharding, barriers, gather, max-over-ranks timing) can be exercised without 8 GPUs.  Never set by
    # the driver; the real path is one rank per GPU over RCCL.
    share = os.environ.get('KN_BENCH_SHARE_GPU') == '1'
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or args.dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if 'MASTER_PORT' not in os.environ:                  # --dist without a launcher: a rendezvous of one
            import socket
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
        if share:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm

    # synthetic encrypted batch, resident in HBM before the timed region
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn((batch,) + tuple(inshape), generator=g, device=dev)
    x_cipher = sensor.fromtensor(x).encrypt().astensor()        # [B, D0+1] view of a feature-major block
    # parity gate of this run: the reference's own integration criterion (test/test_keynet.py:94,112) -- keyed logits
    # equal the source network's on the same plain images -- evaluated on the first images of the timed batch, plain torch f32 on the host
    n_gate = min(4, batch)
    with torch.no_grad():
        y_plain = net(x[:n_gate].cpu()).reshape(n_gate, -1) if rank == 0 else None
    x_plain = x if (rank == 0 and world == 1) else None
    del x
    t0 = time.time()
    y = knet.forward_linear(x_cipher)                            # first call uploads the operators
    torch.cuda.synchronize()
    log('[bench rank %d] operators resident + first forward in %.1f s; logits %s' % (rank, time.time() - t0, tuple(y.shape)))

    if args.trace_layers:
        table = layer_table(knet, batch)
        marker = torch.zeros(1031, device=dev)
        yin = x_cipher
        for row in table:
            c = row['layer']
            marker.add_(1.0)                                   # one torch elementwise kernel = the separator tools/trace_layers.py splits on
            for _ in range(8):
                out = c.forward(yin, fuse_relu=row['fuse'])
            yin = out
        marker.add_(1.0)
        torch.cuda.synchronize()
        json.dump({'workload': desc, 'batch': batch, 'launches_per_layer': 8, 'csrc_sha256': kernel_sources_sha(),
                   'layers': [{k: r[k] for k in ('name', 'kind', 'rows', 'cols', 'nnz', 'flops', 'bytes', 'plan')} for r in table]}, open(args.trace_layers, 'w'), indent=1)
        log('[bench] layer list written to %s' % args.trace_layers)
        return

    replay = knet.capture(x_cipher) if args.graph else None

    def step():
        if not use_dist:
            return (replay(x_cipher) if replay is not None else knet.forward_linear(x_cipher))[:, :-1]
        # N ranks: this rank's shard with the replicas' calibrated contracts kept in agreement (one tiny all-reduce per step for key-nets
        # under the 'auto' contract, none otherwise: keynet_amd.dist.replicated_forward), then ONE RCCL all-gather of the logits over xGMI
        # (gloo rigs bounce through the host)
        yl = replay(x_cipher)[:, :-1] if replay is not None else kdist.replicated_forward(knet, x_cipher)
        return kdist.gather_logits(yl, total=batch * world)

    def timed(n_warm, n_steps):
        for _ in range(n_warm):
            step()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            out = step()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device('cpu') if share else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return (elapsed, out)

    (elapsed, out) = timed(args.warmup, args.steps)
    assert out.shape[0] == batch * world and bool(torch.isfinite(out).all())
    parity = None
    if rank == 0:
        err = float((out[:n_gate].contiguous().cpu() - y_plain).abs().max())   # contiguous first: a strided D2H copy is ~1500 tiny copies
        parity = {'check': 'keyed logits of the timed batch vs the source network (torch f32, host) on %d images' % n_gate, 'max_abs_err': err,
                  'max_abs_logit': float(y_plain.abs().max()), 'err_over_max_logit': err / max(float(y_plain.abs().max()), 1e-30), 'atol': 1e-3,
                  'ok': bool(err <= 1e-3)}
        if not parity['ok']:
            raise AssertionError('parity gate failed: %s' % json.dumps(parity))
    collective = None
    if use_dist:
        collective = collective_record(knet, sensor, x_cipher, out, batch, world, rank, local_rank, dev, inshape, share)
    oracle_par = None
    if rank == 0 and not args.workload.startswith('vgg16'):
        oracle_par = oracle_parity_csr(knet, x_cipher, out[:batch])
        if oracle_par.get('ok') is False:
            raise AssertionError('oracle parity failed: %s' % json.dumps(oracle_par))
    del out

    if rank == 0:
        table = time_layers(x_cipher, layer_table(knet, batch), args.layer_iters)
        nnz_img = float(sum(r['nnz'] for r in table))
        for r in table:
            log('[bench layer] %-8s %-9s rows=%8d nnz=%12d  %8.3f ms  %7.2f TFLOP/s  %8.1f GB/s(alg)%s' %
                (r['name'], r['kind'], r['rows'], r['nnz'], r['ms'], r['flops'] / r['ms'] / 1e9, r['bytes'] / r['ms'] / 1e6,
                 '' if not r.get('flops_executed') else '  (on the stored entries of the fused operator; the split application executes %.2f TFLOP/s)' % (r['flops_executed'] / r['ms'] / 1e9)))
        roof = roofline_of(table, args.workload, batch, mode)
        total_bytes = sum(r['bytes'] for r in table)
        chain = knet._chain_op(dev) if hasattr(knet, '_chain_op') else None
        if chain is not None:
            # the forward of this key-net is ONE launch of the whole-net kernel (csrc/kn_chain.hip): that launch is the dominant kernel.
            # Algorithmic bytes: every operator once (8 B per stored non-zero) + activations in and out of every layer.
            # (a launch is 37 us: 2 000 untimed launches = 75 ms bring the GPU off its idle clock, as in the timed loop; 5 + 50 launches -- 2 ms -- read 41 us)
            for _ in range(2000):
                knet.forward_linear(x_cipher)
            (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            torch.cuda.synchronize()
            e0.record()
            for _ in range(500):
                knet.forward_linear(x_cipher)
            e1.record()
            torch.cuda.synchronize()
            ch_ms = e0.elapsed_time(e1) / 500
            ach = total_bytes / ch_ms / 1e6
            # What really bounds it: bit-exactness with scipy forbids the FMA, so a stored non-zero costs one packed multiply and one
            # packed add per two batch columns, 4 cycles each on one of the CU's four SIMDs; a workgroup owns 4 columns, 256 CUs run a round.
            nnz_net = float(sum(r['nnz'] for r in table))
            rounds = -(-((batch + 3) // 4) // 256)
            valu_floor_ms = rounds * (nnz_net / 64.0) * 4 * 4 / 4 / 2.4e9 * 1e3
            roof = dict(bound='hbm', kernel=chain.plan(batch), achieved=ach, peak=PEAK_HBM_GBS, unit='GB/s', frac=ach / PEAK_HBM_GBS, traffic=None,
                        valu_floor_ms=valu_floor_ms, frac_of_valu_floor=valu_floor_ms / ch_ms,
                        algorithmic_bytes=total_bytes, algorithmic_macs=float(sum(r['nnz'] for r in table)) * batch, ms_per_forward=ch_ms,
                        t_mac_per_s=float(sum(r['nnz'] for r in table)) * batch / ch_ms / 1e9,
                        launch_per_layer_ms={r['name']: round(r['ms'], 4) for r in table},
                        note='one launch for the whole key-net, activations in LDS; launch_per_layer_ms = the seven separate kernels it replaces (KN_NO_CHAIN=1)')
        ms_per_step = 1e3 * elapsed / args.steps
        res = {
            'metric': 'encrypted images/sec (whole node), keyed %s' % {'vgg16': 'VGG-16 224x224', 'vgg16-gain': 'VGG-16 224x224 (float keys: permutation + photometric gain)', 'vgg16-givens': 'VGG-16 224x224 (float keys: Givens rotations + affine photometric, the reference\'s test_vgg16_orthogonal)', 'vgg16-givens28': 'VGG-16 224x224 (float keys: Givens rotations + affine photometric, tile 28: the reference\'s test_vgg16_orthogonal_8)', 'vgg16-stochastic': 'VGG-16 224x224 (float keys: hierarchical permutation + doubly-stochastic blocks + affine photometric: the reference\'s test_vgg16_stochastic)', 'lenet': 'LeNet_AvgPool 28x28', 'allconv': 'AllConvNet 32x32'}[args.workload],
            'value': batch * world * args.steps / elapsed, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': desc, 'mode': mode_desc,
                       # which arithmetic contract the headline `value` was measured under, next to the library default's name (advisor, round 4): a
                       # permutation-only tiled key-net is bit-exact BY DEFAULT; the matrix-core number needs exact='auto' / False at construction
                       'headline_contract': ("opt-in exact='auto' (matrix cores within 1e-5, re-screened)" if mode == 'tolerance' else 'library default (bit-exact)'),
                       'default_contract_value': 'see `exact.images_per_s`' if mode == 'tolerance' else 'this value',
                       'images_per_gpu': batch, 'global_batch': batch * world, 'nnz_per_image': nnz_img,
                       'parallelism': 'batch shards x%d, all_gather(logits)' % world if world > 1 else 'single GPU'},
            'achieved_hbm_gbs_algorithmic': total_bytes / (ms_per_step * 1e6), 'achieved_tflops_algorithmic': 2.0 * nnz_img * batch / (ms_per_step * 1e9),
            'roofline': roof, 'parity': parity,
            'cpu_baseline': cpu if (cpu is not None or world == 1) else 'measured at N=1 only (the scipy baseline runs on rank 0 of a single-GPU run; see BENCH / profiles)',
            'collective': collective,
            'layers_ms': {r['name']: round(r['ms'], 4) for r in table},
            'plans': {r['name']: r['plan'] for r in table},        # kn_spmm_plan: the kernels (tile shape, loader) each layer really takes at this batch
        }
        if oracle_par is not None:
            res['oracle_parity'] = oracle_par
        if secondary is not None:
            res['secondary'] = secondary
        # ---- the headline record is complete here.  Every leg below is reported-only: it runs inside leg(), which turns a failure into
        # an entry of res['errors'], and the line is printed from the `finally` -- a late failure (an out-of-memory in a side leg, say)
        # can no longer lose the headline.
        res['errors'] = {}
        built_mode = None          # exact_mode(None) = back to the contract this key-net was built with (the headline's)

        def leg(name, fn):
            t0 = time.time()
            try:
                fn()
            except Exception as e:
                res['errors'][name] = '%s: %s' % (type(e).__name__, e)
                log('[bench] leg "%s" failed: %s: %s' % (name, type(e).__name__, e))
            log('[bench] leg "%s" took %.1f s' % (name, time.time() - t0))

        def leg_contract():
            rep = knet.contract_report()
            if any(r['calibration'] is not None for r in rep['layers']):
                # float-key contract (KeyedLayer._calibrate): which layers the first forward left on the matrix cores, which it moved to the
                # order-preserving kernels so that |y - y_reference| <= 1e-5 max(1, |y|) holds, and the evidence per layer
                res['contract'] = {'tolerance': 1e-5, 'layers_switched_to_exact': rep['switched'], 'rescreened_every_forward': bool(rep.get('rescreen', False)),
                                   'layers': {r['name']: ({k: r['calibration'].get(k) for k in ('decided', 'bound', 'measured_mfma_vs_exact', 'tol', 'max_abs_rowsum', 'max_abs_x', 'max_abs_y')}
                                                          if r['calibration'] is not None else {'decided': 'exact' if r['exact'] else 'mfma', 'declared': True}) for r in rep['layers']}}

        def leg_slots():
            convs = [(r['name'], r['layer'].W) for r in table if isinstance(r['layer'].W, ksp.Conv2dTiledMatrix) and r['layer'].W._taps is not None]
            res['config']['slots_per_output_pixel'] = {n: {'mean': round(float(len(W._taps['ent_out'])) / (W._outshape[1] * W._outshape[2]), 3),
                                                           'max': int(np.bincount(W._taps['ent_out']).max())} for (n, W) in convs}
            res['config']['entries_carry_coefficients'] = bool(any(W._taps['ent_coef'] is not None for (n, W) in convs))

        def leg_end_to_end():
            res['end_to_end'] = end_to_end(sensor, knet, x_plain, args.steps, 1)

        def leg_graph():
            rp = knet.capture(x_cipher)
            for _ in range(3):
                rp(x_cipher)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                og = rp(x_cipher)
            torch.cuda.synchronize()
            el_g = time.perf_counter() - t0
            res['graph'] = {'images_per_s': batch * args.steps / el_g, 'ms_per_step': 1e3 * el_g / args.steps,
                            'bit_equal_to_eager': bool(torch.equal(og, knet.forward_linear(x_cipher))), 'what': 'the same forward replayed from ONE captured HIP graph (KeyedModel.capture)'}

        def leg_bf16x3():
            # EXPERIMENTAL (--experimental only; never the headline): the same key-net with the bf16x3 kernel as the first candidate of the
            # float-key contract.  Each conv layer keeps it only if its result, measured against the order-preserving kernel on the
            # calibration batch, has 4x headroom under 1e-5 * max(1, |y|).
            try:
                knet.exact_mode('auto-bf16x3')
                knet.forward_linear(x_cipher)
                torch.cuda.synchronize()
                (el_b, out_b) = timed(1, args.steps)
                err_b = float((out_b[:n_gate].contiguous().cpu() - y_plain).abs().max())
                del out_b
                table_b = time_layers(x_cipher, layer_table(knet, batch), max(1, min(args.layer_iters, 3)))
                rep_b = knet.contract_report()
                res['experimental'] = {'bf16x3': {
                    'what': 'KeyedModel.exact_mode(\'auto-bf16x3\'): f32 products emulated on the bf16 matrix pipe (three-way split of both operands, six of the nine '
                            'cross products, f32 accumulate: KN_FLAG_BF16X3) in every conv layer whose calibration measured 4x headroom under the 1e-5 tolerance',
                    'dtype': 'f32 emulated (3 x bf16 split, 6 of 9 cross products, f32 accumulate)',
                    'images_per_s': batch * args.steps / el_b, 'ms_per_step': 1e3 * el_b / args.steps, 'steps': args.steps,
                    'parity': {'vs_source_network_max_abs_err': err_b, 'ok': bool(err_b <= 1e-3),
                               'per_layer_vs_order_preserving_kernel': {r['name']: {k: r['calibration'].get(k) for k in ('decided', 'measured_bf16x3_vs_exact', 'measured_mfma_vs_exact', 'tol')}
                                                                        for r in rep_b['layers'] if r['calibration'] is not None and 'tol' in r['calibration']}},
                    'layers_on_bf16x3': [r['name'] for r in rep_b['layers'] if r['exact'] == 'bf16x3'],
                    'roofline': roofline_of(table_b, args.workload, batch, 'tolerance'),
                    'layers_ms': {r['name']: round(r['ms'], 4) for r in table_b}}}
            finally:
                knet.exact_mode(built_mode)

        def leg_exact():
            # the same key-net under the bit-exact contract (north_star: "bit-exact for the permutation-only key"; the DEFAULT of a
            # permutation-only tiled key-net -- the headline above opted into the matrix cores explicitly, config.mode says so)
            try:
                knet.exact_mode(True)
                t0 = time.time()
                knet.forward_linear(x_cipher)                    # uploads the CSR twins of fc6-8
                torch.cuda.synchronize()
                log('[bench exact] exact-mode operators resident + first forward in %.1f s' % (time.time() - t0))
                (el_x, out_x) = timed(1, args.steps)
                err_x = float((out_x[:n_gate].contiguous().cpu() - y_plain).abs().max())
                del out_x
                table_x = time_layers(x_cipher, layer_table(knet, batch), max(1, min(args.layer_iters, 3)))
                for r in table_x:
                    log('[bench exact] %-8s %-9s %8.3f ms  %7.2f T MAC/s' % (r['name'], r['kind'], r['ms'], r['nnz'] * batch / r['ms'] / 1e9))
                par_x = exact_parity(knet, x_cipher)
                par_x['vs_source_network_max_abs_err'] = err_x
                par_x['ok'] = bool(par_x['ok'] and err_x <= 1e-3)
                par_x['note'] = 'this leg samples conv1_2 and conv4_2; all 21 layers are checked the same way by tests/test_vgg16_full_gpu.py'
                res['exact'] = {'images_per_s': batch * args.steps / el_x, 'ms_per_step': 1e3 * el_x / args.steps, 'steps': args.steps,
                                'mode': 'KeyedModel.exact_mode(True): every layer in the reference\'s accumulation order and rounding (no FMA, no MFMA)',
                                'roofline': roofline_of(table_x, args.workload, batch, 'exact'), 'parity': par_x,
                                'layers_ms': {r['name']: round(r['ms'], 4) for r in table_x}}
                if not par_x['ok']:
                    raise AssertionError('exact-mode parity failed: %s' % json.dumps(par_x))
            finally:
                knet.exact_mode(built_mode)

        def leg_float_key_parity():
            res['float_key_parity'] = float_key_parity(dev)

        def leg_exact_layers():
            # float-key workloads: the layers the contract keeps in the reference's order, AS TIMED (whole batch, the key-net's own activations as input), against the CPU oracle
            # on the canonical CSR of sampled output pixels (Conv2dTiledMatrix.rows_csr: a pixel pair hit by several taps is one stored entry, its terms summed in entry order)
            names = tuple(r['name'] for r in knet.contract_report()['layers'] if r['exact'] is True and r['calibration'] is not None and r['calibration'].get('decided') == 'exact')
            if names:
                res['exact_layers_parity'] = exact_parity(knet, x_cipher, n_img=8, n_pix=1, layers=names)

        try:
            single = world == 1 and replay is None
            leg('contract', leg_contract)
            if args.workload.startswith('vgg16'):
                leg('slots', leg_slots)
            if single and x_plain is not None:
                leg('end_to_end', leg_end_to_end)
            if args.graph_leg and single:
                leg('graph', leg_graph)
            if args.workload == 'vgg16' and single and not args.exact and not args.no_exact_leg:
                leg('exact', leg_exact)
                leg('float_key_parity', leg_float_key_parity)
            if args.exact_layers_parity and args.workload.startswith('vgg16-') and single and not args.exact:
                leg('exact_layers_parity', leg_exact_layers)
            if args.experimental and args.workload.startswith('vgg16') and single and not args.exact:
                leg('experimental_bf16x3', leg_bf16x3)
        finally:
            if not res['errors']:
                del res['errors']
            detail = args.detail or os.path.join(ROOT, DETAIL_FILE)
            write_detail(res, args.detail)
            log('[bench detail] ' + json.dumps(res, default=str))
            os.write(json_fd, (compact_record(res, os.path.relpath(detail, ROOT) if detail.startswith(ROOT) else detail) + '\n').encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
