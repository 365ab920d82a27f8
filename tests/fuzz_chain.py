#!/usr/bin/env python3
"""One-off fuzz of the whole-net kernel and the bf16x3 kernel against the CPU oracle (random shapes, patterns, batch widths).
    gpurun -- 'python3 tests/fuzz_chain.py 150'   (checker script: lives under tests/ because it uses the oracle)"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle                                    # noqa: E402  (checker)
from keynet_amd import _capi, sparse as ksp      # noqa: E402

dev = torch.device('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(12345)
S = torch.cuda.current_stream().cuda_stream
bad = 0
for case in range(n_cases):
    n_ops = rng.randint(1, 7)
    dims = [int(rng.randint(1, 400)) if rng.rand() < 0.8 else int(rng.randint(1024, 2600)) for _ in range(n_ops + 1)]      # (>= 1024 rows: the two-rows-per-lane layout)
    mats = []
    for l in range(n_ops):
        (rows, cols) = (dims[l + 1], dims[l])
        kind = rng.randint(0, 4)
        (ip, ix, dt) = ([0], [], [])
        shared = rng.randint(0, cols, size=rng.randint(0, min(cols, 60) + 1))
        for r in range(rows):
            if kind == 0:
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
    ops = [_capi.Operator.csr(m[0], m[1], m[2], m[3]) for m in mats]
    try:
        chain = _capi.Operator.chain(ops, [m[4] for m in mats])
    except _capi.KeynetHipError as e:
        print('case', case, 'refused:', str(e)[:80])
        continue
    n = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 33, 64, 130, 257]))
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
print('chain fuzz: %d cases, %d mismatches' % (n_cases, bad))
sys.exit(1 if bad else 0)
