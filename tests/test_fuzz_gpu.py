"""Seeded fuzzers of the whole-net kernel and of the factored untiled conv route against the CPU oracle, as collected `-m gpu` tests with a case budget
(round-5 review: they used to be scripts outside pytest).  Larger runs:  python3 tests/test_fuzz_gpu.py chain 150 | factored 60 | csr 200 | convtaps 200 | tiled 200 | models 100 | floatmodels 60 | dense 100"""
import os
import sys
import numpy as np
import pytest
import scipy.sparse
import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle                                    # noqa: E402  (checker)
from keynet_amd import _capi, sparse as ksp      # noqa: E402
from keynet_amd.layer import KeyedLayer          # noqa: E402

pytestmark = pytest.mark.gpu


def fuzz_chain(n_cases, seed=12345, verbose=False):
    """Random chains of 1 - 6 CSR operators (five row-pattern kinds incl. keyed-Linear layers on the sequential thin walk, >= 1024-row layouts, ReLU flags, Inf activations, batch widths 1 .. 257) through
    kn_chain_create / kn_spmm against the oracle layer by layer: bit-equal incl. NaN positions.  Returns (cases run, refused, mismatches)."""
    dev = torch.device('cuda:0')
    S = torch.cuda.current_stream().cuda_stream
    (bad, refused) = (0, 0)
    fuzz_chain.sequential_layers = 0              # (how many layers of the run took the sequential thin walk: the test asserts the fuzzer reaches it)
    rng = np.random.RandomState(seed)
    for case in range(n_cases):
        n_ops = rng.randint(1, 7) if rng.rand() < 0.9 else rng.randint(7, 13)                    # (up to the twelve operators a chain holds)
        dims = [int(rng.randint(1, 400)) if rng.rand() < 0.8 else int(rng.randint(1024, 2600)) for _ in range(n_ops + 1)]      # (>= 1024 rows: the two-rows-per-lane layout)
        mats = []
        for l in range(n_ops):
            (rows, cols) = (dims[l + 1], dims[l])
            kind = rng.randint(0, 5)
            (ip, ix, dt) = ([0], [], [])
            shared = rng.randint(0, cols, size=rng.randint(0, min(cols, 60) + 1))
            if kind == 4:                       # a keyed nn.Linear: (nearly) all rows carry ONE sequence of distinct columns, any length (behind another layer and >= 64 long: the
                shared = rng.permutation(cols)[:rng.randint(max(cols - 7, 1), cols + 1)]      # sequential thin walk, its tail of 1-3 entries, its odd rows on a wavefront of their own)
            for r in range(rows):
                if kind == 4:
                    c = shared if rng.rand() < 0.97 else rng.randint(0, cols, size=rng.randint(0, 4))
                elif kind == 0:
                    c = rng.randint(0, cols, size=rng.randint(0, 12))
                elif kind == 1:
                    if r % int(rng.randint(2, 20)) == 0:
                        shared = rng.randint(0, cols, size=rng.randint(0, min(cols, 60) + 1))
                    c = shared
                elif kind == 2:
                    c = rng.permutation(cols)[:rng.randint(max(cols - 2, 0), cols + 1)]
                else:
                    c = shared if rng.rand() < 0.8 else rng.randint(0, cols, size=rng.randint(0, 5))
                ix.extend(int(v) for v in c)
                dt.extend(rng.randn(len(c)).astype(np.float32))
                ip.append(len(ix))
            mats.append(((rows, cols), np.array(ip, np.int32), np.array(ix, np.int32), np.array(dt, np.float32), int(rng.randint(0, 2))))
        if verbose:
            print('case', case, 'dims', dims, 'relu', [m[4] for m in mats], flush=True)
        ops = [_capi.Operator.csr(m[0], m[1], m[2], m[3]) for m in mats]
        try:
            chain = _capi.Operator.chain(ops, [m[4] for m in mats])
        except _capi.KeynetHipError as e:
            refused += 1
            if verbose:
                print('case', case, 'refused:', str(e)[:80])
            continue
        n = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 33, 64, 130, 257]))
        import re
        m_seq = re.search(r'-- (\d+) of them sequentially', chain.plan(n))
        fuzz_chain.sequential_layers += int(m_seq.group(1)) if m_seq else 0
        X = rng.randn(dims[0], n).astype(np.float32)
        if rng.rand() < 0.2:
            X[rng.randint(0, dims[0]), rng.randint(0, n)] = np.inf          # non-finite activations must reach exactly the reference's outputs
        xd = torch.as_tensor(X).to(dev)
        yd = torch.empty((dims[-1], n), device=dev)
        chain.spmm(xd.data_ptr(), n, n, yd.data_ptr(), n, 2, S)
        ref = X
        with np.errstate(all='ignore'):
            for m in mats:
                ref = oracle.csr_matvecs(m[0], m[1], m[2], m[3], ref)
                if m[4]:
                    ref = np.where(ref < 0, np.float32(0), ref)               # torch relu: NaN stays NaN
        got = yd.cpu().numpy()
        if not np.array_equal(got, ref, equal_nan=True):
            bad += 1
            print('case', case, 'MISMATCH dims', dims, 'n', n, 'max', np.nanmax(np.abs(got - ref)))
        if rng.rand() < 0.3:
            # the same forward on a column window of wider arrays (leading dimensions > n, a start column of any alignment): same bits inside, nothing written outside
            (w0, pad) = (int(rng.choice([0, 1, 2, 3, 4, 5])), int(rng.choice([0, 1, 3, 4, 9])))
            wide = w0 + n + pad
            Xw = rng.randn(dims[0], wide).astype(np.float32)
            Xw[:, w0:w0 + n] = X
            xw = torch.as_tensor(Xw).to(dev)
            yw = torch.full((dims[-1], wide), -7.0, dtype=torch.float32, device=dev)
            chain.spmm(xw.data_ptr() + 4 * w0, wide, n, yw.data_ptr() + 4 * w0, wide, 2, S)
            yw = yw.cpu().numpy()
            if not (np.array_equal(yw[:, w0:w0 + n], ref, equal_nan=True) and np.all(yw[:, :w0] == -7.0) and np.all(yw[:, w0 + n:] == -7.0)):
                bad += 1
                print('case', case, 'WINDOW MISMATCH dims', dims, 'n', n, 'w0', w0, 'wide', wide)
    return (n_cases, refused, bad)


def fuzz_csr(n_cases, seed=4242, verbose=False):
    """Random structured CSR operators through SparseMatrix.torchdot (kn_spmm: loose rows, pattern groups of 2 .. 600 members incl. the pipelined / matrix-pipe grouped
    kernels' thresholds, rows that are their group's pattern minus a few entries, a Linear-like big group, one very long row, empty rows, duplicate and unsorted columns,
    explicit zeros) against the oracle: bit-equal incl. NaN positions, with and without ReLU, contiguous and transposed activations, batch widths 1 .. 1024.
    Returns (cases run, mismatches)."""
    dev = torch.device('cuda:0')
    bad = 0
    rng = np.random.RandomState(seed)
    fuzz_csr.kernels = set()
    for case in range(n_cases):
        n = int(rng.choice([1, 5, 33, 157, 900, 2100, 3000]))
        n_vecs = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 31, 64, 65, 128, 130, 255, 256, 300, 512, 640, 1024]))
        budget = int(1.5e9 // n_vecs)                                  # stored entries the oracle gets through in well under a second (~7 G multiply-adds per second)
        rows = []
        nnz = 0
        while nnz < budget and len(rows) < 60000 and (len(rows) == 0 or rng.rand() < 0.93):
            kind = rng.randint(0, 8)
            if kind == 6 and n >= 33:                                  # very many small groups: the software-pipelined grouped kernel's regime on wide batches
                members = int(rng.choice([4, 8, 16, 24]))
                for _ in range(int(rng.randint(200, 3000))):
                    pattern = rng.randint(0, n, rng.randint(1, 71)).astype(np.int32)
                    rows.extend([pattern] * members)
                    if rng.rand() < 0.2:
                        rows.append(rng.randint(0, n, rng.randint(0, 12)).astype(np.int32))
                    if sum(len(r) for r in rows[-members:]) * len(rows) // members > budget:
                        break
                nnz = sum(len(r) for r in rows)
                continue
            if kind == 7 and n >= 900 and n_vecs >= 128:               # groups of >= 96 members with long stored sequences: products on the matrix pipe
                for _ in range(int(rng.randint(20, 140))):
                    pattern = rng.randint(0, n, rng.randint(256, 600)).astype(np.int32)
                    rows.extend([pattern] * int(rng.choice([96, 100, 128, 200])))
                    if sum(len(r) for r in rows) > budget:
                        break
                nnz = sum(len(r) for r in rows)
                continue
            if kind == 0:                                              # loose rows
                for _ in range(rng.randint(1, 80)):
                    rows.append(rng.randint(0, n, rng.randint(0, 41)).astype(np.int32))
            elif kind in (1, 2):                                       # a pattern group (kind 2: some members lost a few entries)
                pattern = rng.randint(0, n, rng.randint(0, 121)).astype(np.int32)
                for _ in range(int(rng.choice([2, 3, 7, 8, 15, 16, 17, 31, 32, 33, 64, 100, 257, 600]))):
                    if kind == 2 and len(pattern) > 3 and rng.rand() < 0.1:
                        rows.append(np.delete(pattern, rng.choice(len(pattern), size=rng.randint(1, 4), replace=False)))
                    else:
                        rows.append(pattern)
            elif kind == 3 and n >= 2048 and nnz + 300 * n < budget:   # a keyed Linear: one big group over (nearly) all columns
                pattern = rng.permutation(n)[:rng.randint(n - 3, n + 1)].astype(np.int32)
                for _ in range(int(rng.choice([256, 300]))):
                    rows.append(pattern)
            elif kind == 4:                                            # one long row with duplicates
                rows.append(rng.randint(0, n, rng.randint(500, 4000)).astype(np.int32))
            else:                                                      # empty rows
                for _ in range(rng.randint(1, 5)):
                    rows.append(np.zeros(0, np.int32))
            nnz = sum(len(r) for r in rows)
        if rng.rand() < 0.5:
            order = rng.permutation(len(rows))                         # groups scattered over the operator
            rows = [rows[i] for i in order]
        m = len(rows)
        indptr = np.concatenate(([0], np.cumsum([len(r) for r in rows]))).astype(np.int32)
        indices = np.concatenate(rows).astype(np.int32) if nnz else np.zeros(0, np.int32)
        data = rng.randn(len(indices)).astype(np.float32)
        f64 = bool(rng.rand() < 0.15)                                              # a float64 operator (the challenge notebook's): float64 sums of float32 activations
        if f64:
            data = rng.randn(len(indices)) * np.exp(rng.uniform(-20, 20, len(indices)))
        if len(data) and rng.rand() < 0.3:
            data[rng.randint(0, len(data), size=min(len(data), 5))] = 0.0          # explicit zeros are stored entries
        X = rng.randn(n, n_vecs).astype(np.float32)
        if rng.rand() < 0.3:
            for _ in range(3):
                X[rng.randint(n), rng.randint(n_vecs)] = rng.choice([np.inf, -np.inf, np.nan])
        relu = bool(rng.rand() < 0.5)
        if verbose:
            print('case', case, 'shape', (m, n), 'nnz', len(indices), 'n_vecs', n_vecs, 'relu', relu, 'f64' if f64 else '', flush=True)
        W = ksp.SparseMatrix(scipy.sparse.csr_matrix((data, indices, indptr), shape=(m, n)))
        with np.errstate(all='ignore'):
            ref = oracle.csr_matvecs((m, n), indptr, indices, data, X)
            if relu:
                ref = np.where(ref < 0, ref.dtype.type(0), ref)        # torch relu: NaN stays NaN
        if rng.rand() < 0.5:
            xd = torch.as_tensor(np.ascontiguousarray(X.T)).to(dev).t()           # what x_affine.t() is for a row-major batch
        else:
            xd = torch.as_tensor(X).to(dev)
        got = W.torchdot(xd, relu=relu).cpu().numpy()
        if not f64 and rng.rand() < 0.4:
            # the same product on a column WINDOW of a wider block through the C ABI (leading dimensions > n_vecs, a start column of any alignment): same bits inside the
            # window, nothing written outside it
            (w0, pad) = (int(rng.choice([0, 1, 2, 3, 4, 5, 8, 64])), int(rng.choice([0, 1, 3, 4, 37, 128])))
            wide = w0 + n_vecs + pad
            Xw = rng.randn(n, wide).astype(np.float32)
            Xw[:, w0:w0 + n_vecs] = X
            xw = torch.as_tensor(Xw).to(dev)
            yw = torch.full((m, wide), -7.0, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                W._device_op(dev).spmm(xw.data_ptr() + 4 * w0, wide, n_vecs, yw.data_ptr() + 4 * w0, wide, _capi.KN_FLAG_EXACT | (_capi.KN_FLAG_RELU if relu else 0),
                                       torch.cuda.current_stream().cuda_stream)
            yw = yw.cpu().numpy()
            if not (np.array_equal(yw[:, w0:w0 + n_vecs], ref, equal_nan=True) and np.all(yw[:, :w0] == -7.0) and np.all(yw[:, w0 + n_vecs:] == -7.0)):
                bad += 1
                print('case', case, 'WINDOW MISMATCH shape', (m, n), 'n_vecs', n_vecs, 'w0', w0, 'wide', wide)
        if not f64 and m * n_vecs < 4e6 and rng.rand() < 0.3:
            # kn_spmm_planes: the operator on several activation blocks in one launch == one kn_spmm per block (or refused: the caller loops), surroundings untouched
            n_pl = int(rng.randint(2, 6))
            ld = n_vecs + int(rng.choice([0, 3, 32]))
            (xs, ys) = (n * ld + int(rng.choice([0, 4, 4 * ld])), m * ld + int(rng.choice([0, 8, 8 * ld])))
            Xp = rng.randn(n_pl * xs + 8).astype(np.float32)
            xp = torch.as_tensor(Xp).to(dev)
            st = torch.cuda.current_stream().cuda_stream
            fl = _capi.KN_FLAG_EXACT | (_capi.KN_FLAG_RELU if relu else 0)
            (ya, yb) = (torch.full((n_pl * ys + 8,), -7.0, dtype=torch.float32, device=dev), torch.full((n_pl * ys + 8,), -7.0, dtype=torch.float32, device=dev))
            with torch.cuda.device(dev):
                op = W._device_op(dev)
                took = op.spmm_planes(xp.data_ptr(), ld, xs, n_pl, n_vecs, ya.data_ptr(), ld, ys, fl, st)
                for q in range(n_pl):
                    op.spmm(xp.data_ptr() + 4 * q * xs, ld, n_vecs, yb.data_ptr() + 4 * q * ys, ld, fl, st)
            if took and not bool(torch.equal(ya, yb)):
                bad += 1
                print('case', case, 'PLANES MISMATCH shape', (m, n), 'n_vecs', n_vecs, 'planes', n_pl, 'ld', ld)
            if not took and not bool((ya == -7.0).all()):
                bad += 1
                print('case', case, 'PLANES refused but wrote', (m, n))
            with np.errstate(all='ignore'):
                r0 = oracle.csr_matvecs((m, n), indptr, indices, data, np.ascontiguousarray(Xp[:n * ld].reshape(n, ld)[:, :n_vecs]))
                r0 = np.where(r0 < 0, np.float32(0), r0) if relu else r0
            if not np.array_equal(yb.cpu().numpy()[:m * ld].reshape(m, ld)[:, :n_vecs], r0, equal_nan=True):
                bad += 1
                print('case', case, 'PLANE 0 differs from the oracle', (m, n), n_vecs)
        with torch.cuda.device(dev):
            import re
            fuzz_csr.kernels |= set(re.findall(r'(csr_\w+_kernel)', W._device_op(dev).plan(n_vecs, _capi.KN_FLAG_EXACT)))
        if not np.array_equal(got, ref, equal_nan=True):
            bad += 1
            print('case', case, 'MISMATCH shape', (m, n), 'n_vecs', n_vecs, 'relu', relu, 'max', np.nanmax(np.abs(got - ref)))
    return (n_cases, bad)


def fuzz_convtaps(n_cases, seed=2024, verbose=False, only=None, hook=None):
    """Random FACTORED conv operators (Conv2dTiledMatrix.fromtaps: 1 .. 25 taps, 0 .. 40 input pixels per output pixel, one or several taps per pixel pair -- the
    filled-in key families --, unit or float coefficients, channel counts that are and are not multiples of the kernels' bundles, with and without the homogeneous
    column) under KN_FLAG_EXACT against the oracle on the operator's canonical CSR (a pair's terms summed in entry order): bit-equal incl. NaN positions, ReLU on
    and off, batch widths 1 .. 640; the matrix-core path of the same operator inside the float-key bound.  Returns (cases run, mismatches)."""
    import re
    dev = torch.device('cuda:0')
    bad = 0
    rng = np.random.RandomState(seed)
    fuzz_convtaps.kernels = set()
    for case in range(n_cases):
        Cin = int(rng.choice([1, 2, 3, 4, 8, 16, 17, 32, 48, 64]))
        Cout = int(rng.choice([1, 5, 8, 16, 24, 32, 33, 64, 96, 128, 160]))
        side = int(rng.choice([10, 10, 17, 29]))                    # (mostly small; sometimes hundreds of pixels: the wide forms of the kernels want thousands of work items)
        (hi, wi, ho, wo) = (int(rng.randint(1, side)), int(rng.randint(1, side)), int(rng.randint(1, side)), int(rng.randint(1, side)))
        (Pin, Pout) = (hi * wi, ho * wo)
        ntaps = int(rng.choice([1, 2, 4, 9, 16, 17, 25]))
        n_vecs = int(rng.choice([1, 7, 64, 65, 96, 128, 130, 200, 256, 300, 384, 512, 640]))
        taps = (rng.randn(ntaps, Cout, Cin) / np.sqrt(max(Cin, 1))).astype(np.float32)
        filled = rng.rand() < 0.5                                   # several taps per (output, input) pixel pair
        max_in = int(rng.choice([1, 3, 9, 40]))
        coef = rng.rand() < 0.5
        (eo, ei, et, ec) = ([], [], [], [])
        for o in range(Pout):
            for i in rng.choice(Pin, size=rng.randint(0 if rng.rand() < 0.1 else 1, min(Pin, max_in) + 1), replace=False):
                k = int(rng.randint(1, min(ntaps, 6) + 1)) if filled else 1
                for t in rng.choice(ntaps, size=k, replace=False):  # entry order, not tap order
                    eo.append(o); ei.append(int(i)); et.append(int(t)); ec.append(np.float32(1.0) if rng.rand() < 0.2 else np.float32(rng.randn()))
        if not eo:
            (eo, ei, et, ec) = ([0], [0], [0], [np.float32(1.0)])
        if Cout * Cin * len(eo) * max(n_vecs, 64) > 2e9:            # keep the oracle's work and the host expansion small
            n_vecs = 64
        if Cout * Cin * len(eo) > 6e7:
            continue
        has_last = bool(rng.rand() < 0.7)
        lastcol = None
        if has_last:
            lastcol = np.concatenate((rng.randn(Cout * Pout), [1.0])).astype(np.float32)
            lastcol[rng.randint(0, Cout * Pout, size=3)] = 0.0
        W = ksp.Conv2dTiledMatrix.fromtaps((Cin, hi, wi), (Cout, ho, wo), taps, np.array(eo, np.int32), np.array(ei, np.int32), np.array(et, np.int32),
                                           np.array(ec, np.float32) if coef else None, lastcol)
        X = rng.randn(W.shape[1], n_vecs).astype(np.float32)
        if rng.rand() < 0.2:
            X[rng.randint(0, Cin * Pin), rng.randint(n_vecs)] = rng.choice([np.inf, -np.inf, np.nan])
        if has_last:
            X[-1] = 1.0
        relu = bool(rng.rand() < 0.5)
        if only is not None and case != only:                       # (re-running ONE case of a longer run: the generator has consumed what it would have)
            continue
        if hook is not None:
            hook(W, X, relu)
        with torch.cuda.device(dev):
            plan = W._device_op(dev).plan(n_vecs, _capi.KN_FLAG_EXACT)
        fuzz_convtaps.kernels |= set(re.findall(r'(convtaps_\w+_kernel(?:<[^>]*>)?)', plan))
        if verbose:
            print('case', case, 'Cin', Cin, 'Cout', Cout, 'pixels', (Pin, Pout), 'taps', ntaps, 'entries', len(eo), 'filled', filled, 'coef', coef, 'last', has_last, 'n_vecs', n_vecs,
                  'relu', relu, '|', plan[:110], flush=True)
        M = W.tosparse('csr')
        M.sort_indices()
        with np.errstate(all='ignore'):
            ref = oracle.csr_matvecs(M.shape, M.indptr, M.indices, M.data.astype(np.float32), X)
            if relu:
                ref = np.where(ref < 0, np.float32(0), ref)
        got = W.torchdot(torch.as_tensor(X).to(dev), relu=relu, exact=True).cpu().numpy()
        if np.all(np.isfinite(X)):
            # the matrix-core path of the same operator: another order of the same f32 sum -- inside 1e-5 max(1, |ref|) + 8 eps sum |w x|, element-wise
            ym = W.torchdot(torch.as_tensor(X).to(dev), relu=relu, exact=False).cpu().numpy()
            # (sum over the TERMS: the matrix cores add a pair's terms c_t w_t x one by one, the stored order adds the pair's rounded sum once -- with cancelling terms the
            # merged |entry| understates what either evaluation rounds)
            Wabs = ksp.Conv2dTiledMatrix.fromtaps((Cin, hi, wi), (Cout, ho, wo), np.abs(taps), np.array(eo, np.int32), np.array(ei, np.int32), np.array(et, np.int32),
                                                  np.abs(np.array(ec, np.float32)) if coef else None, None if lastcol is None else np.abs(lastcol))
            S = Wabs.tosparse('csr').astype(np.float64).dot(np.abs(X.astype(np.float64)))
            # (8 eps: rows of this generator hold up to ~10 000 terms; the stored-order f32 sum itself sits ~5 eps sum |w x| from the f64 value on such rows, the matrix cores' as far
            # on the other side at worst -- measured on case 194 of seed 2024, where 2 eps was exceeded by 1.7 % at one element of 369 k)
            if not np.all(np.abs(ym.astype(np.float64) - ref) <= 1e-5 * np.maximum(1.0, np.abs(ref)) + 8 * 2.0 ** -24 * S):
                bad += 1
                with torch.cuda.device(dev):
                    print('case', case, 'MATRIX-CORE PATH off by', float(np.abs(ym - ref).max()), '|', W._device_op(dev).plan(n_vecs, 0)[:160])
            if filled and case % 3 == 0:
                # the split application (spatial CSR on every input-channel plane, then a 9-slot conv on the result): a third association of the same sum
                ys = W.torchdot(torch.as_tensor(X).to(dev), relu=relu, exact='split').cpu().numpy()
                if not np.all(np.abs(ys.astype(np.float64) - ref) <= 1e-5 * np.maximum(1.0, np.abs(ref)) + 8 * 2.0 ** -24 * S):
                    bad += 1
                    print('case', case, 'SPLIT APPLICATION off by', float(np.abs(ys - ref).max()))
        if rng.rand() < 0.4:
            (w0, pad) = (int(rng.choice([0, 1, 2, 3, 4, 5, 8, 64])), int(rng.choice([0, 1, 3, 4, 37, 128])))
            wide = w0 + n_vecs + pad
            Xw = rng.randn(W.shape[1], wide).astype(np.float32)
            Xw[:, w0:w0 + n_vecs] = X
            xw = torch.as_tensor(Xw).to(dev)
            yw = torch.full((W.shape[0], wide), -7.0, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                W._device_op(dev).spmm(xw.data_ptr() + 4 * w0, wide, n_vecs, yw.data_ptr() + 4 * w0, wide, _capi.KN_FLAG_EXACT | (_capi.KN_FLAG_RELU if relu else 0),
                                       torch.cuda.current_stream().cuda_stream)
            yw = yw.cpu().numpy()
            if not (np.array_equal(yw[:, w0:w0 + n_vecs], ref, equal_nan=True) and np.all(yw[:, :w0] == -7.0) and np.all(yw[:, w0 + n_vecs:] == -7.0)):
                bad += 1
                print('case', case, 'WINDOW MISMATCH n_vecs', n_vecs, 'w0', w0, 'wide', wide, '|', plan[:120])
        if not np.array_equal(got, ref, equal_nan=True):
            bad += 1
            print('case', case, 'MISMATCH', 'Cin', Cin, 'Cout', Cout, 'pixels', (Pin, Pout), 'taps', ntaps, 'entries', len(eo), 'filled', filled, 'coef', coef, 'last', has_last,
                  'n_vecs', n_vecs, 'relu', relu, 'max', np.nanmax(np.abs(got - ref)), '|', plan[:160])
    return (n_cases, bad)


def fuzz_tiled(n_cases, seed=99, verbose=False):
    """Random TiledMatrix / DiagonalTiledMatrix operators (tile shapes that do and do not divide the matrix, repeated tiles, empty blocks, explicit tile structure) through
    kn_tiled_create (expanded once on the device) against the oracle on the canonical CSR of the same matrix: bit-equal, ReLU on and off.  Returns (cases run, mismatches)."""
    dev = torch.device('cuda:0')
    bad = 0
    rng = np.random.RandomState(seed)
    for case in range(n_cases):
        (m, n) = (int(rng.randint(1, 700)), int(rng.randint(1, 700)))
        (h, w) = (int(rng.randint(1, 90)), int(rng.randint(1, 90)))
        n_vecs = int(rng.choice([1, 3, 8, 64, 65, 130, 256, 300]))
        kind = rng.randint(0, 3)
        if kind == 0:                                               # unstructured
            T = scipy.sparse.random(m, n, density=float(rng.choice([0.0, 0.002, 0.02, 0.2])), format='coo', dtype=np.float32, random_state=rng)
            W = ksp.TiledMatrix(T, (h, w))
        elif kind == 1:                                             # a few distinct tiles repeated on a random block pattern, cropped to a shape the tile does not divide
            tiles = [scipy.sparse.random(h, w, density=float(rng.choice([0.05, 0.3, 1.0])), format='csr', dtype=np.float32, random_state=rng) for _ in range(rng.randint(1, 4))]
            (gb, gw) = ((m + h - 1) // h, (n + w - 1) // w)
            rows = []
            for _ in range(gb):
                rows.append([tiles[rng.randint(len(tiles))] if rng.rand() < 0.4 else None for _ in range(gw)])
            if all(t is None for r in rows for t in r):
                rows[0][0] = tiles[0]
            T = scipy.sparse.bmat([[t if t is not None else scipy.sparse.csr_matrix((h, w), dtype=np.float32) for t in r] for r in rows], format='csr')[:m, :n].tocoo()
            W = ksp.TiledMatrix(T, (h, w))
        else:                                                       # one block down the diagonal (a block key), the last position an identity corner
            b = int(rng.randint(1, 40))
            B = scipy.sparse.random(b, b, density=float(rng.choice([0.1, 0.5, 1.0])), format='csr', dtype=np.float32, random_state=rng)
            sq = int(rng.randint(1, 500))
            if rng.rand() < 0.5 and b <= sq:                        # (an oversized DENSE block is an AttributeError in the reference too -- keynet/sparse.py:660 calls .tocsr() on it)
                B = B.toarray()                                     # dense block: every entry stored, zeros included
            W = ksp.DiagonalTiledMatrix(B, (sq, sq))
            T = W.tosparse('coo')
            (m, n) = W.shape
        M = scipy.sparse.csr_matrix(T)
        M.sum_duplicates()
        M.sort_indices()
        X = rng.randn(n, n_vecs).astype(np.float32)
        relu = bool(rng.rand() < 0.5)
        if verbose:
            print('case', case, 'kind', kind, 'shape', (m, n), 'tile', W.tileshape(), 'nnz', M.nnz, 'n_vecs', n_vecs, flush=True)
        with np.errstate(all='ignore'):
            ref = oracle.csr_matvecs(M.shape, M.indptr, M.indices, M.data.astype(np.float32), X)
            if relu:
                ref = np.where(ref < 0, np.float32(0), ref)
        got = W.torchdot(torch.as_tensor(X).to(dev), relu=relu).cpu().numpy()
        if not np.array_equal(got, ref, equal_nan=True):
            bad += 1
            print('case', case, 'MISMATCH kind', kind, 'shape', (m, n), 'tile', W.tileshape(), 'n_vecs', n_vecs, 'max', np.nanmax(np.abs(got - ref)))
    return (n_cases, bad)


from fuzz_nets import random_net as _random_net      # noqa: E402


def fuzz_models(n_cases, seed=31337, verbose=False):
    """Random small source networks keyed by permutations -- untiled (the whole-net kernel when it applies) or tiled (the per-layer operators, bit-exact default of
    permutation-only keys) -- through sensor.encrypt + KeyedModel.forward on the device: the logits bit-equal to the oracle run over the key-net's exported operators
    (keynet_amd.io -> oracle.keynet_forward), and equal to the plaintext network's within 1e-4.  Returns (cases run, mismatches)."""
    import tempfile
    from keynet_amd import system as ksys, io as kio
    dev = torch.device('cuda:0')
    bad = 0
    rng = np.random.RandomState(seed)
    for case in range(n_cases):
        torch.manual_seed(int(rng.randint(1 << 30)))
        np.random.seed(int(rng.randint(1 << 30)))
        (net, inshape, names) = _random_net(rng)
        tile = int(rng.choice([0, 0, 2, 3, 4, 8]))
        n = int(rng.choice([1, 2, 5, 8, 33, 64, 130]))
        if verbose:
            print('case', case, 'inshape', inshape, 'tile', tile, 'n', n, names, flush=True)
        (sensor, knet) = ksys.PermutationKeynet(inshape, net) if tile == 0 else ksys.TiledPermutationKeynet(inshape, net, tile)
        x = torch.randn((n,) + inshape)
        with torch.no_grad():
            plain = net(x).numpy()
        xc = sensor.fromtensor(x.to(dev)).encrypt().astensor()
        y = knet.forward_linear(xc).cpu().numpy()
        with tempfile.TemporaryDirectory() as d:
            kio.save_keynet(knet, os.path.join(d, 'k.npz'))
            z = np.load(os.path.join(d, 'k.npz'), allow_pickle=False)
            with np.errstate(all='ignore'):
                ref = oracle.keynet_forward(oracle.load_golden_layers(z), xc.cpu().numpy())
        ok_bits = np.array_equal(y, ref)
        ok_plain = bool(np.allclose(y[:, :plain.shape[1]], plain, atol=1e-4, rtol=1e-4))
        if not (ok_bits and ok_plain):
            bad += 1
            print('case', case, 'MISMATCH bits', ok_bits, 'plain', ok_plain, 'inshape', inshape, 'tile', tile, 'n', n, names, 'max vs oracle', float(np.abs(y - ref).max()),
                  'max vs plain', float(np.abs(y[:, :plain.shape[1]] - plain).max()))
    return (n_cases, bad)


def fuzz_dense(n_cases, seed=555, verbose=False):
    """Random keyed nn.Linear operators (dense block + bias column + homogeneous row, columns in a permuted stored order): the order-preserving kernels bit-equal to the
    oracle, the split-K matrix-core GEMM of the tolerance contract (kn_dense_create, where the operator qualifies) inside the float-key bound.  Returns (cases run, mismatches)."""
    dev = torch.device('cuda:0')
    bad = 0
    rng = np.random.RandomState(seed)
    fuzz_dense.dense_ops = 0
    for case in range(n_cases):
        (outs, ins) = (int(rng.randint(1, 1500)), int(rng.randint(1, 3000)))
        if rng.rand() < 0.6:                                        # what the matrix-core GEMM takes: whole 256-column K chunks, enough elements
            (outs, ins) = (int(rng.randint(66, 1500)), 256 * int(rng.randint(1, 13)))
        n_vecs = int(rng.choice([1, 2, 4, 7, 64, 100, 128, 130, 200, 256, 512]))
        D = np.zeros((outs + 1, ins + 1), dtype=np.float32)
        D[:-1, :-1] = (rng.randn(outs, ins) / np.sqrt(ins)).astype(np.float32)
        D[:-1, -1] = rng.randn(outs).astype(np.float32)
        D[-1, -1] = 1.0
        perm = rng.permutation(ins)
        M = scipy.sparse.csr_matrix(D)
        M = scipy.sparse.csr_matrix((M.data, np.where(M.indices < ins, perm[np.minimum(M.indices, ins - 1)], M.indices).astype(np.int32), M.indptr), shape=M.shape)
        W = ksp.SparseMatrix(M)
        X = np.vstack((rng.randn(ins, n_vecs).astype(np.float32), np.ones((1, n_vecs), np.float32)))
        relu = bool(rng.rand() < 0.5)
        if verbose:
            print('case', case, 'outs', outs, 'ins', ins, 'n_vecs', n_vecs, 'relu', relu, flush=True)
        ref = oracle.csr_matvecs(M.shape, M.indptr, M.indices, M.data, X)
        if relu:
            ref = np.where(ref < 0, np.float32(0), ref)
        xd = torch.as_tensor(X).to(dev)
        if not np.array_equal(W.torchdot(xd, relu=relu).cpu().numpy(), ref):
            bad += 1
            print('case', case, 'MISMATCH (stored order) outs', outs, 'ins', ins, 'n_vecs', n_vecs)
        ym = W.torchdot(xd, relu=relu, exact=False).cpu().numpy()
        fuzz_dense.dense_ops += W._dense_device_op() is not None
        S = abs(M).astype(np.float64).dot(np.abs(X.astype(np.float64)))
        if not np.all(np.abs(ym.astype(np.float64) - ref) <= 1e-5 * np.maximum(1.0, np.abs(ref)) + 8 * 2.0 ** -24 * S):
            bad += 1
            print('case', case, 'MATRIX-CORE PATH off by', float(np.abs(ym - ref).max()), 'outs', outs, 'ins', ins, 'n_vecs', n_vecs)
    return (n_cases, bad)


def fuzz_float_models(n_cases, seed=161803, verbose=False):
    """Random small source networks under FLOAT key families (the reference's orthogonal family: Givens rotations + affine photometric keys; its doubly-stochastic family:
    filled-in operators) with the library's default contract for them ('auto': every layer decides on its first batch between the matrix cores, the split application and
    the stored order): logits against the oracle over the exported operators and against the plaintext network, a second forward bit-equal to the first, every layer's
    decision recorded.  Returns (cases run, mismatches)."""
    import tempfile
    import warnings
    from keynet_amd import system as ksys, io as kio
    dev = torch.device('cuda:0')
    bad = 0
    rng = np.random.RandomState(seed)
    fuzz_float_models.decisions = {}
    fuzz_float_models.skipped = 0
    stochastic = dict(tileshape=(4, 4), global_geometric='hierarchical_permutation', hierarchical_blockshape=(2, 2), hierarchical_permute_at_level=(0, 1),
                      local_geometric='doubly_stochastic', alpha=2.0, blocksize=4, local_photometric='uniform_random_affine', beta=1.0, gamma=1.0)
    for case in range(n_cases):
        torch.manual_seed(int(rng.randint(1 << 30)))
        np.random.seed(int(rng.randint(1 << 30)))
        (net, inshape, names) = _random_net(rng, sides=(8, 16))
        family = 'stochastic' if rng.rand() < 0.4 else 'orthogonal'
        n = int(rng.choice([1, 4, 33, 64, 130]))
        if verbose:
            print('case', case, family, 'inshape', inshape, 'n', n, names, flush=True)
        try:
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                (sensor, knet) = ksys.Keynet(inshape, net, **stochastic) if family == 'stochastic' else ksys.TiledOrthogonalKeynet(inshape, net, 4)
        except (AssertionError, ValueError) as e:                  # a shape this key family does not admit (the reference asserts the same way)
            fuzz_float_models.skipped += 1
            if verbose:
                print('   skipped:', type(e).__name__, str(e)[:80])
            continue
        x = torch.randn((n,) + inshape)
        with torch.no_grad():
            plain = net(x).numpy()
        xc = sensor.fromtensor(x.to(dev)).encrypt().astensor()
        y = knet.forward_linear(xc).cpu().numpy()
        y2 = knet.forward_linear(xc).cpu().numpy()
        for r in knet.contract_report()['layers']:
            d = (r['calibration'] or {}).get('decided', 'declared exact' if r['exact'] else 'declared mfma')
            fuzz_float_models.decisions[d] = fuzz_float_models.decisions.get(d, 0) + 1
        with tempfile.TemporaryDirectory() as d:
            kio.save_keynet(knet, os.path.join(d, 'k.npz'))
            z = np.load(os.path.join(d, 'k.npz'), allow_pickle=False)
            with np.errstate(all='ignore'):
                ref = oracle.keynet_forward(oracle.load_golden_layers(z), xc.cpu().numpy())
        scale = max(1.0, float(np.abs(ref).max()))
        (e_ref, e_plain) = (float(np.abs(y - ref).max()), float(np.abs(y[:, :plain.shape[1]] - plain).max()))
        ok = e_ref <= 1e-4 * scale and e_plain <= 1e-3 * scale and np.array_equal(y, y2)
        if not ok:
            bad += 1
            print('case', case, 'MISMATCH', family, 'inshape', inshape, 'n', n, names, 'vs oracle', e_ref, 'vs plain', e_plain, 'scale', scale, 'repeatable', bool(np.array_equal(y, y2)))
    return (n_cases, bad)


def fuzz_factored(n_cases, seed=777, verbose=False):
    """Random untiled convs (channel counts, image sides, strides, exact-zero weights, Inf / NaN activations, batch widths) through the factored route and the
    forced 16-row big-group kernel against the oracle on the STORED CSR: bit-equal incl. NaN positions.  Returns (cases run, mismatches)."""
    dev = torch.device('cuda:0')
    bad = 0
    total = 0
    old = KeyedLayer.FACTOR_UNTILED_MIN_NNZ
    KeyedLayer.FACTOR_UNTILED_MIN_NNZ = 0
    try:
        rng = np.random.RandomState(seed)
        for case in range(n_cases):
            cin = int(rng.randint(1, 9))
            cout = 32 * int(rng.randint(1, 7)) if rng.rand() < 0.8 else int(rng.randint(1, 40))      # (not a multiple of 32: no table, the conv pipeline)
            stride = int(rng.choice([1, 1, 2]))
            hw = int(rng.randint(3, 15)) * stride
            k = int(rng.choice([3, 3, 1]))
            n_vecs = int(rng.choice([64, 128, 130, 256, 300, 512]))
            torch.manual_seed(case)
            m = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2)
            n_zero = int(rng.randint(0, 4))
            with torch.no_grad():
                for _ in range(n_zero):
                    m.weight[rng.randint(cout), rng.randint(cin), rng.randint(k), rng.randint(k)] = 0.0
            (HW, HWo) = (hw * hw, (hw // stride) ** 2)
            eye = (lambda n: scipy.sparse.identity(n + 1, dtype=np.float32, format='csr'))
            layer = KeyedLayer(m, (cin, hw, hw), (cout, hw // stride, hw // stride), eye(cout * HWo), eye(cin * HW))
            W = layer.W
            fact = isinstance(W, ksp.FactoredSparseMatrix)
            (ip, ix, dt) = ksp._stored_order_csr(W._matrix if ksp.is_scipy_sparse(W._matrix) else scipy.sparse.csr_matrix(W._matrix))
            X = np.vstack((rng.randn(cin * HW, n_vecs).astype(np.float32), np.ones((1, n_vecs), np.float32)))
            if rng.rand() < 0.5:
                for _ in range(3):
                    X[rng.randint(cin * HW), rng.randint(n_vecs)] = rng.choice([np.inf, -np.inf, np.nan])
            with np.errstate(all='ignore'):
                ref = oracle.csr_matvecs(W.shape, ip, ix, dt, X)
            relu = bool(rng.randint(2))
            with np.errstate(all='ignore'):
                r = np.where(ref < 0, np.float32(0), ref) if relu else ref
            y = W.torchdot(torch.as_tensor(X).to(dev), relu=relu).cpu().numpy()
            ok = np.array_equal(y, r, equal_nan=True)
            with torch.cuda.device(dev):
                plan = W._device_op(dev).plan(n_vecs, 2 | (1 if relu else 0)).split(' ')[0]
            total += 1
            if verbose:
                print('case %3d cin %d cout %3d hw %2d k %d stride %d zeros %d n_vecs %3d factored %d %-28s %s' % (case, cin, cout, hw, k, stride, n_zero, n_vecs, fact, plan[:28], 'ok' if ok else 'MISMATCH'), flush=True)
            bad += not ok
        # big pattern groups: forced 16-row matrix-pipe kernel
        os.environ['KN_BIG_MFMA16'] = '1'      # read when the operator is created
        for case in range(max(4, n_cases // 6)):
            rows = int(rng.randint(256, 700))
            cols = int(rng.randint(2048, 2600))
            n_vecs = int(rng.choice([64, 100, 192, 256, 320]))
            pat = rng.permutation(cols + 3)[:cols].astype(np.int32)
            lists = [pat] * rows + [rng.randint(0, cols + 3, rng.randint(0, 9)).astype(np.int32) for _ in range(4)]
            indptr = np.concatenate(([0], np.cumsum([len(v) for v in lists]))).astype(np.int32)
            indices = np.concatenate(lists).astype(np.int32)
            data = rng.randn(len(indices)).astype(np.float32)
            M = scipy.sparse.csr_matrix((data, indices, indptr), shape=(len(lists), cols + 3))
            W = ksp.SparseMatrix(M)
            X = rng.randn(cols + 3, n_vecs).astype(np.float32)
            ref = oracle.csr_matvecs(M.shape, indptr, indices, data, X)
            y = W.torchdot(torch.as_tensor(X).to(dev), relu=False).cpu().numpy()
            ok = np.array_equal(y, ref)
            total += 1
            if verbose:
                print('big  %3d rows %d cols %d n_vecs %d %s' % (case, rows, cols, n_vecs, 'ok' if ok else 'MISMATCH'), flush=True)
            bad += not ok
    finally:
        KeyedLayer.FACTOR_UNTILED_MIN_NNZ = old
        os.environ.pop('KN_BIG_MFMA16', None)
    return (total, bad)


def test_fuzz_whole_net_kernel():
    (n, refused, bad) = fuzz_chain(100)
    assert bad == 0 and refused < n // 2, (n, refused, bad)
    assert fuzz_chain.sequential_layers >= 10, fuzz_chain.sequential_layers


def test_fuzz_csr_operators():
    (n, bad) = fuzz_csr(80)
    assert bad == 0, (n, bad)
    assert len(fuzz_csr.kernels) >= 3, fuzz_csr.kernels          # (the run reached several of the order-preserving CSR kernels)


def test_fuzz_factored_conv_operators():
    (n, bad) = fuzz_convtaps(150)
    assert bad == 0, (n, bad)
    assert len(fuzz_convtaps.kernels) >= 6, fuzz_convtaps.kernels


def test_fuzz_tiled_operators():
    (n, bad) = fuzz_tiled(150)
    assert bad == 0, (n, bad)


def test_fuzz_keyed_models():
    (n, bad) = fuzz_models(60)
    assert bad == 0, (n, bad)


def test_fuzz_float_keyed_models():
    (n, bad) = fuzz_float_models(30)
    assert bad == 0 and fuzz_float_models.skipped < n // 2, (n, bad, fuzz_float_models.skipped)
    assert len(fuzz_float_models.decisions) >= 2, fuzz_float_models.decisions


def test_fuzz_keyed_linear_operators():
    (n, bad) = fuzz_dense(40)
    assert bad == 0 and fuzz_dense.dense_ops >= 10, (n, bad, fuzz_dense.dense_ops)


def test_fuzz_factored_untiled_route():
    (n, bad) = fuzz_factored(24)
    assert bad == 0 and n >= 28, (n, bad)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'chain'
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    fn = {'chain': fuzz_chain, 'csr': fuzz_csr, 'convtaps': fuzz_convtaps, 'tiled': fuzz_tiled, 'models': fuzz_models, 'floatmodels': fuzz_float_models, 'dense': fuzz_dense, 'factored': fuzz_factored}[which]
    r = fn(cases, verbose=True, seed=int(sys.argv[3])) if len(sys.argv) > 3 else fn(cases, verbose=True)       # (a third argument: another seed than the tier's)
    if which == 'floatmodels':
        print('decisions:', fuzz_float_models.decisions, 'skipped:', fuzz_float_models.skipped)
    if which in ('csr', 'convtaps'):
        print('kernels reached:', sorted(fuzz_csr.kernels if which == 'csr' else fuzz_convtaps.kernels))
    print('%s fuzz: cases / (refused) / mismatches = %s%s' % (which, r, '; layers on the sequential thin walk: %d' % fuzz_chain.sequential_layers if which == 'chain' else ''))
    sys.exit(1 if r[-1] else 0)
