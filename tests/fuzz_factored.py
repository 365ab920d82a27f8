#!/usr/bin/env python3
"""One-off fuzz of the factored untiled conv route (tap-table kernel / conv pipeline + zero guard) and of the 16-row big-group kernel against the CPU oracle
on the STORED CSR: random channel counts, image sides, strides, exact-zero weights, Inf / NaN activations, batch widths.
    gpurun -- 'python3 tests/fuzz_factored.py 60'   (checker script: lives under tests/ because it uses the oracle)"""
import os
import sys
import numpy as np
import scipy.sparse
import torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle                                    # noqa: E402  (checker)
from keynet_amd import sparse as ksp             # noqa: E402
from keynet_amd.layer import KeyedLayer          # noqa: E402

KeyedLayer.FACTOR_UNTILED_MIN_NNZ = 0
dev = torch.device('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(777)
bad = 0
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
    print('case %3d cin %d cout %3d hw %2d k %d stride %d zeros %d n_vecs %3d factored %d %-28s %s' % (case, cin, cout, hw, k, stride, n_zero, n_vecs, fact, plan[:28], 'ok' if ok else 'MISMATCH'), flush=True)
    bad += not ok
# big pattern groups: forced 16-row matrix-pipe kernel
os.environ['KN_BIG_MFMA16'] = '1'
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
    print('big  %3d rows %d cols %d n_vecs %d %s' % (case, rows, cols, n_vecs, 'ok' if ok else 'MISMATCH'), flush=True)
    bad += not ok
print('mismatches: %d' % bad)
sys.exit(1 if bad else 0)
