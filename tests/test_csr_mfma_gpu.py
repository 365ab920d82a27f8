"""The matrix-pipe grouped kernel (csrc/kn_csr_mfma.hip: products by v_mfma_f32_32x32x1_2b_f32 with a ZERO accumulator = the IEEE-rounded
f32 product; running sums by v_pk_add_f32 in stored order) against the CPU oracle (scipy csr_matvecs restated), bit for bit, and against
the vector-ALU kernels it replaces (KN_GROUP_MFMA=0).  The KN_* options are read when an operator is CREATED and recorded in its handle
(kn_spmm_plan prints them): each formulation gets its own handle."""
import os

import numpy as np
import pytest
import scipy.sparse
import torch

import oracle
from keynet_amd import sparse as ksp
from keynet_amd import _capi

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def grouped_operator(rng, n, members, n_groups, ncol_of, loose_every=5, dtype_scale=1.0):
    rows = []
    for g in range(n_groups):
        pattern = rng.randint(0, n, ncol_of(g)).astype(np.int32)          # unsorted, duplicates allowed
        for _ in range(members):
            rows.append(pattern)
        if loose_every and g % loose_every == 0:
            rows.append(rng.randint(0, n, rng.randint(0, 12)).astype(np.int32))
    indptr = np.concatenate(([0], np.cumsum([len(r) for r in rows]))).astype(np.int32)
    indices = np.concatenate(rows).astype(np.int32)
    data = (rng.randn(len(indices)) * dtype_scale).astype(np.float32)
    return (len(rows), indptr, indices, data)


@pytest.mark.parametrize('members,n_vecs,nrb', [(96, 512, 1), (192, 256, 1), (32, 1024, 1), (24, 512, 1), (40, 384, 1), (100, 260, 1), (70, 1000, 1), (200, 128, 1),
                                                (96, 512, 3), (192, 256, 3), (100, 260, 3), (70, 1000, 2), (200, 128, 3), (160, 384, 2)])
def test_matrix_pipe_grouped_kernel_vs_oracle(members, n_vecs, nrb, monkeypatch):
    """Groups of 24 .. 200 member rows, chunks of one (the default), two and three 32-row blocks per workgroup (KN_MF_NRB, read when the operator is
    created) with partly filled last blocks, 1 .. 75 stored columns per group (fewer than the six columns in flight, every remainder modulo six),
    batches that are not a multiple of 64 or 256, loose rows in between, ReLU on and off."""
    monkeypatch.setenv('KN_GROUP_MFMA', '1')                 # forced: by default only operators with long stored sequences (mean >= 256 columns) take it
    if nrb != 1:
        monkeypatch.setenv('KN_MF_NRB', str(nrb))
    rng = np.random.RandomState(members * 1000 + n_vecs)
    n = 1100
    n_groups = max(600 * 96 // members, 40)
    (m, indptr, indices, data) = grouped_operator(rng, n, members, n_groups, lambda g: 1 + (g * 7) % 75)
    X = rng.randn(n, n_vecs).astype(np.float32)
    W = ksp.SparseMatrix(scipy.sparse.csr_matrix((data, indices, indptr), shape=(m, n)))
    with torch.cuda.device(dev()):
        plan = W._device_op(dev()).plan(n_vecs, _capi.KN_FLAG_EXACT)
    assert 'csr_group_mfma_kernel' in plan and ('row blocks=%d' % nrb) in plan, plan
    ref = oracle.csr_matvecs((m, n), indptr, indices, data, X)
    xd = torch.as_tensor(X).to(dev())
    for relu in (False, True):
        y = W.torchdot(xd, relu=relu).cpu().numpy()
        assert np.array_equal(y, np.maximum(ref, 0) if relu else ref), (members, n_vecs, relu, int(np.sum(y != (np.maximum(ref, 0) if relu else ref))))
    monkeypatch.setenv('KN_GROUP_MFMA', '0')                 # a second handle of the same operator, created with the vector-ALU kernels forced
    W0 = ksp.SparseMatrix(W._matrix)
    with torch.cuda.device(dev()):
        plan0 = W0._device_op(dev()).plan(n_vecs, _capi.KN_FLAG_EXACT)
    assert 'csr_group_mfma_kernel' not in plan0 and 'group_mfma=0' in plan0, plan0
    assert 'group_mfma=1' in plan                            # the options a handle was created with are part of its plan
    assert np.array_equal(W0.torchdot(xd).cpu().numpy(), ref)


def test_matrix_pipe_products_on_special_values(monkeypatch):
    """What could tell a matrix-pipe product from v_mul_f32: denormal operands and results, products that underflow to zero or overflow to
    Inf, signed zeros (a -0 product comes back as +0: invisible in a sum that starts at +0.0), Inf and NaN activations (NaN / Inf reach exactly
    the outputs whose rows hold that column; 0 * Inf = NaN like the reference).  Column windows (ldx = ldy > n_vecs) through the C ABI."""
    monkeypatch.setenv('KN_GROUP_MFMA', '1')
    rng = np.random.RandomState(5)
    n = 400
    (m, indptr, indices, data) = grouped_operator(rng, n, 96, 60, lambda g: 3 + (g * 5) % 40, loose_every=0)
    spec = np.array([0.0, -0.0, 1e-30, -1e-30, 1e-38, 3e-39, 1e-45, 1e30, -3e38, 1.0, -1.0, 2.5e-20, 7e19], dtype=np.float32)
    data[::7] = spec[np.arange(len(data[::7])) % len(spec)]
    n_vecs = 256
    X = rng.randn(n, n_vecs).astype(np.float32)
    X[::3, ::5] = spec[(np.arange(len(X[::3, 0]))[:, None] + np.arange(len(X[0, ::5]))[None, :]) % len(spec)]
    X[7, 3] = np.inf
    X[11, 64] = -np.inf
    X[13, 200] = np.nan
    W = ksp.SparseMatrix(scipy.sparse.csr_matrix((data, indices, indptr), shape=(m, n)))
    with np.errstate(all='ignore'):
        ref = oracle.csr_matvecs((m, n), indptr, indices, data, X)
    y = W.torchdot(torch.as_tensor(X).to(dev())).cpu().numpy()
    assert np.array_equal(y, ref, equal_nan=True), int(np.sum(~((y == ref) | (np.isnan(y) & np.isnan(ref)))))
    assert np.isnan(ref).any() and np.isinf(ref).any() and (np.abs(ref[np.isfinite(ref) & (ref != 0)]) < 1.2e-38).any()      # the cases really occur
    # a column window of a wider block
    ld = 640
    Xw = rng.randn(n, ld).astype(np.float32)
    xd = torch.as_tensor(Xw).to(dev())
    yd = torch.full((m, ld), 7.0, device=dev())
    c0 = 256
    with torch.cuda.device(dev()):
        W._device_op(dev()).spmm(xd.data_ptr() + 4 * c0, ld, 320, yd.data_ptr() + 4 * c0, ld, _capi.KN_FLAG_EXACT | _capi.KN_FLAG_RELU, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    refw = np.maximum(oracle.csr_matvecs((m, n), indptr, indices, data, np.ascontiguousarray(Xw[:, c0:c0 + 320])), 0)
    got = yd.cpu().numpy()
    assert np.array_equal(got[:, c0:c0 + 320], refw) and np.all(got[:, :c0] == 7.0) and np.all(got[:, c0 + 320:] == 7.0)


def test_matrix_pipe_with_patched_members(monkeypatch):
    """Patched group members (rows that lost an entry to an exact zero ride in the group with 0.0f; kn_csr.hip) through the matrix-pipe kernel,
    incl. Inf / NaN at a missing position (the guard kernel restores the reference's result)."""
    monkeypatch.setenv('KN_GROUP_MFMA', '1')
    rng = np.random.RandomState(3)
    (n_cols, n_groups, members, seq_len) = (900, 560, 96, 70)          # enough work items for the matrix-pipe dispatch
    (ip, ix, dt) = ([0], [], [])
    missing = []
    for g in range(n_groups):
        seq = rng.permutation(n_cols)[:seq_len]
        for mm in range(members):
            keep = np.ones(seq_len, bool)
            if mm in (3, 50, 95):
                lose = {3: [0], 50: [seq_len - 1, 5], 95: [1, 30, 31]}[mm]
                keep[lose] = False
                missing.append((len(ip) - 1, seq[lose]))
            c = seq[keep]
            ix.extend(int(v) for v in c)
            dt.extend(rng.randn(len(c)).astype(np.float32))
            ip.append(len(ix))
    shape = (len(ip) - 1, n_cols)
    (ip, ix, dt) = (np.array(ip, np.int32), np.array(ix, np.int32), np.array(dt, np.float32))
    op = _capi.Operator.csr(shape, ip, ix, dt)
    n_vecs = 256
    with torch.cuda.device(dev()):
        plan = op.plan(n_vecs, _capi.KN_FLAG_EXACT)
    assert 'csr_group_mfma_kernel' in plan and 'csr_patch_guard_kernel<%d patched rows>' % len(missing) in plan, plan
    X = rng.randn(n_cols, n_vecs).astype(np.float32)
    for poison in (False, True):
        Xp = X.copy()
        if poison:
            Xp[missing[0][1][0], 1] = np.inf
            Xp[missing[4][1][-1], 2] = np.nan
        xd = torch.as_tensor(Xp).to(dev())
        yd = torch.empty((shape[0], n_vecs), device=dev())
        with torch.cuda.device(dev()):
            op.spmm(xd.data_ptr(), n_vecs, n_vecs, yd.data_ptr(), n_vecs, _capi.KN_FLAG_EXACT, torch.cuda.current_stream().cuda_stream)
        with np.errstate(invalid='ignore', over='ignore'):
            ref = oracle.csr_matvecs(shape, ip, ix, dt, Xp)
        got = yd.cpu().numpy()
        assert np.array_equal(got, ref, equal_nan=True), poison


def test_default_dispatch_rule():
    """By default the matrix-pipe kernel serves pattern groups with long stored sequences (mean >= 256 columns per member row: the 3x3 conv
    layers of AllConvNet with 96 / 192 input channels); short sequences stay on the vector-ALU pipeline (same bits either way)."""
    rng = np.random.RandomState(9)
    for (ncol, expect) in ((300, True), (60, False)):
        (m, indptr, indices, data) = grouped_operator(rng, 2000, 96, 560, lambda g: ncol, loose_every=0)
        op = _capi.Operator.csr((m, 2000), indptr, indices, data)
        with torch.cuda.device(dev()):
            plan = op.plan(256, _capi.KN_FLAG_EXACT)
        assert ('csr_group_mfma_kernel' in plan) == expect, (ncol, plan)
        X = rng.randn(2000, 256).astype(np.float32)
        y = torch.empty((m, 256), device=dev())
        xd = torch.as_tensor(X).to(dev())
        with torch.cuda.device(dev()):
            op.spmm(xd.data_ptr(), 256, 256, y.data_ptr(), 256, _capi.KN_FLAG_EXACT, torch.cuda.current_stream().cuda_stream)
        assert np.array_equal(y.cpu().numpy(), oracle.csr_matvecs((m, 2000), indptr, indices, data, X))


@pytest.mark.parametrize('rows,cols,n_vecs,forced', [(300, 2500, 256, True), (257, 2048, 64, True), (520, 2100, 100, True), (1030, 2050, 2048, False)])
def test_big_group_16_row_matrix_pipe_kernel_vs_oracle(rows, cols, n_vecs, forced, monkeypatch):
    """A keyed nn.Linear in the reference's order (ONE pattern group: >= 256 member rows x >= 2048 stored columns, here in a scrambled stored order with
    duplicate columns): csr_group_mfma16_kernel (v_mfma_f32_16x16x1_4b_f32 with a zero accumulator + packed adds) against the oracle and against the
    LDS-staged big-group kernel (KN_BIG_MFMA16=0 at create), bit for bit; partly filled last 16-row chunk, ragged batches, loose rows beside the group, ReLU on
    and off.  By default it is taken when its 16-row x 64-column wavefronts number >= 2048 (last case); the small cases force it."""
    if forced:
        monkeypatch.setenv('KN_BIG_MFMA16', '1')
    rng = np.random.RandomState(rows + n_vecs)
    pattern = rng.permutation(cols + 7)[:cols].astype(np.int32)
    pattern[5] = pattern[900]                                              # a duplicate column: two stored entries, added in stored order
    lists = [pattern] * rows + [rng.randint(0, cols + 7, 9).astype(np.int32) for _ in range(5)] + [np.zeros(0, np.int32)]
    indptr = np.concatenate(([0], np.cumsum([len(r) for r in lists]))).astype(np.int32)
    indices = np.concatenate(lists).astype(np.int32)
    data = rng.randn(len(indices)).astype(np.float32)
    (m, n) = (len(lists), cols + 7)
    X = rng.randn(n, n_vecs).astype(np.float32)
    W = ksp.SparseMatrix(scipy.sparse.csr_matrix((data, indices, indptr), shape=(m, n)))
    with torch.cuda.device(dev()):
        plan = W._device_op(dev()).plan(n_vecs, _capi.KN_FLAG_EXACT)
    assert 'csr_group_mfma16_kernel' in plan and 'csr_big_group_kernel' not in plan, plan
    ref = oracle.csr_matvecs((m, n), indptr, indices, data, X)
    xd = torch.as_tensor(X).to(dev())
    for relu in (False, True):
        r = np.maximum(ref, 0) if relu else ref
        y = W.torchdot(xd, relu=relu).cpu().numpy()
        assert np.array_equal(y, r), (rows, cols, n_vecs, relu, int(np.sum(y != r)))
    monkeypatch.setenv('KN_BIG_MFMA16', '0')                 # a second handle with the LDS-staged kernel forced
    W0 = ksp.SparseMatrix(W._matrix)
    with torch.cuda.device(dev()):
        plan0 = W0._device_op(dev()).plan(n_vecs, _capi.KN_FLAG_EXACT)
    assert 'csr_group_mfma16_kernel' not in plan0 and 'csr_big_group_kernel' in plan0, plan0
    assert np.array_equal(W0.torchdot(xd, relu=False).cpu().numpy(), ref)
    # few wavefronts (one per SIMD or less): the LDS-staged kernel by default
    monkeypatch.delenv('KN_BIG_MFMA16')
    W1 = ksp.SparseMatrix(W._matrix)
    with torch.cuda.device(dev()):
        assert 'csr_group_mfma16_kernel' not in W1._device_op(dev()).plan(128, _capi.KN_FLAG_EXACT)
