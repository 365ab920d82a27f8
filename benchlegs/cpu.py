"""cpu_baseline leg: the reference's own arithmetic -- scipy.sparse.csr_matrix.dot (keynet/sparse.py:488-492) -- timed on this node's host cores BEFORE the GPU is
touched (the worker pool is forked from a GPU-free process)."""
import os
import subprocess
import time

import numpy as np

from keynet_amd import sparse as ksp
from .common import log, keyed_layers, host_nnz


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
