"""Seeded fuzzers of the whole-net kernel and of the factored untiled conv route against the CPU oracle, as collected `-m gpu` tests with a case budget
(round-5 review: they used to be scripts outside pytest).  Larger runs:  python3 tests/test_fuzz_gpu.py chain 150 | factored 60"""
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
        n_ops = rng.randint(1, 7)
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
    return (n_cases, refused, bad)


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
    (n, refused, bad) = fuzz_chain(40)
    assert bad == 0 and refused < n // 2, (n, refused, bad)
    assert fuzz_chain.sequential_layers >= 3, fuzz_chain.sequential_layers


def test_fuzz_factored_untiled_route():
    (n, bad) = fuzz_factored(24)
    assert bad == 0 and n >= 28, (n, bad)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'chain'
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    r = fuzz_chain(cases, verbose=True) if which == 'chain' else fuzz_factored(cases, verbose=True)
    print('%s fuzz: cases / (refused) / mismatches = %s%s' % (which, r, '; layers on the sequential thin walk: %d' % fuzz_chain.sequential_layers if which == 'chain' else ''))
    sys.exit(1 if r[-1] else 0)
