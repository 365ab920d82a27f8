"""An untiled keyed conv layer whose stored CSR is provably the ascending-column expansion of its factored form (identity / channel-replicated
permutation keys on both sides: every conv layer of PermutationKeynet AllConvNet but the first) runs on the device from its taps
(keynet_amd.sparse.FactoredSparseMatrix): host proof (CPU) and, on the GPU, bit-equality with the CPU oracle on the STORED CSR -- incl. filter
weights that are exact zeros (absent from the reference's CSR) under Inf / NaN activations."""
import numpy as np
import pytest
import scipy.sparse
import torch
from torch import nn

import oracle
from keynet_amd import sparse as ksp
from keynet_amd.layer import KeyedLayer


def conv_layer(cin, cout, hw, stride=1, zeros=(), perm_in=None, perm_out=None, seed=0):
    """KeyedLayer for an untiled conv (tileshape None) under identity or channel-replicated permutation keys; `zeros` = (co, ci, i, j) weights set to 0."""
    torch.manual_seed(seed)
    m = nn.Conv2d(cin, cout, 3, stride=stride, padding=1)
    with torch.no_grad():
        for (co, ci, i, j) in zeros:
            m.weight[co, ci, i, j] = 0.0
    (HW, HWo) = (hw * hw, (hw // stride) ** 2)

    def key(channels, numel, perm):
        n = channels * numel
        p = np.arange(numel) if perm is None else perm
        rows = np.concatenate([c * numel + np.arange(numel) for c in range(channels)] + [[n]])
        cols = np.concatenate([c * numel + p for c in range(channels)] + [[n]])
        return scipy.sparse.csr_matrix((np.ones(n + 1, np.float32), (rows, cols)), shape=(n + 1, n + 1))
    A = key(cout, HWo, perm_out)
    Ain = key(cin, HW, perm_in)
    return KeyedLayer(m, (cin, hw, hw), (cout, hw // stride, hw // stride), A, Ain.transpose().tocsr())


@pytest.fixture()
def small_threshold(monkeypatch):
    monkeypatch.setattr(KeyedLayer, 'FACTOR_UNTILED_MIN_NNZ', 0)


def test_host_proof_accepts_and_refuses(small_threshold):
    rng = np.random.RandomState(0)
    plain = conv_layer(4, 6, 8)
    assert isinstance(plain.W, ksp.FactoredSparseMatrix) and plain.W.nnz() == plain.W._matrix.nnz
    strided = conv_layer(8, 16, 8, stride=2, zeros=[(1, 2, 0, 0), (4, 0, 2, 1)])      # 2 exact zeros among 1 152 weights
    assert isinstance(strided.W, ksp.FactoredSparseMatrix)
    assert strided.W._matrix.nnz < strided.W._factored.rows_csr().nnz                 # the exact zeros are absent from the reference's CSR
    # a spatial permutation on the INPUT side scrambles the stored column order: not the ascending expansion, the CSR it is
    permuted = conv_layer(4, 6, 8, perm_in=rng.permutation(64))
    assert type(permuted.W) is ksp.SparseMatrix
    # many exact zeros (a pruned filter): refused, the device would re-check every one of them
    pruned = conv_layer(4, 6, 8, zeros=[(co, ci, i, j) for co in range(6) for ci in range(2) for i in range(3) for j in range(3)])
    assert type(pruned.W) is ksp.SparseMatrix
    # the host API is the plain container's
    assert plain.W.tocoo().nnz == plain.W.nnz() and plain.nnz() == plain.W._matrix.nnz


def test_layers_below_the_threshold_stay_csr():
    assert type(conv_layer(4, 6, 8).W) is ksp.SparseMatrix                            # LeNet-sized operators keep the whole-net kernel's CSR form


@pytest.mark.gpu
@pytest.mark.parametrize('n_vecs', [256, 40])
def test_factored_device_form_is_bit_equal_to_the_stored_csr(small_threshold, n_vecs):
    dev = torch.device('cuda:0')
    layer = conv_layer(6, 32, 10, zeros=[(3, 2, 1, 1), (17, 0, 0, 2), (31, 5, 2, 2)], seed=3)
    W = layer.W
    assert isinstance(W, ksp.FactoredSparseMatrix)
    (ip, ix, dt) = ksp._stored_order_csr(W._matrix)
    rng = np.random.RandomState(1)
    X = np.vstack((rng.randn(6 * 100, n_vecs).astype(np.float32), np.ones((1, n_vecs), np.float32)))
    with torch.cuda.device(dev):
        plan = W._device_op(dev).plan(n_vecs, 2)
    # wide batches: the matrix-pipe grouped kernel fed from the tap table; narrow ones: the order-preserving conv kernels; the zero guard behind either
    assert ('taps> (factored operator' in plan if n_vecs >= 128 else 'convtaps_exact' in plan) and 'convtaps_zero_guard_kernel<3 zero tap entries>' in plan, plan
    for poison in (False, True):
        Xp = X.copy()
        if poison:
            Xp[2 * 100 + 45, 1] = np.inf            # channel 2, a pixel in the window of many outputs: hits the (3, 2, 1, 1) zero weight at its centre tap
            Xp[0 * 100 + 12, 2] = np.nan
            Xp[5 * 100 + 77, 3] = -np.inf
        with np.errstate(all='ignore'):
            ref = oracle.csr_matvecs(W.shape, ip, ix, dt, Xp)
        for relu in (False, True):
            with np.errstate(all='ignore'):
                r = np.where(ref < 0, np.float32(0), ref) if relu else ref
            y = W.torchdot(torch.as_tensor(Xp).to(dev), relu=relu).cpu().numpy()
            assert np.array_equal(y, r, equal_nan=True), (poison, relu, np.argwhere(~((y == r) | (np.isnan(y) & np.isnan(r))))[:5])
        if poison:
            # the zero weight (co = 3, ci = 2, centre tap): output pixel 45 of channel 3 must NOT see the Inf at input pixel 45 of channel 2 through it
            assert np.isinf(Xp[245, 1]) and not np.isnan(ref[3 * 100 + 45, 1])


@pytest.mark.gpu
@pytest.mark.parametrize('cin,cout,hw,stride,n_vecs', [(4, 64, 9, 1, 192), (8, 96, 12, 2, 300), (5, 32, 7, 1, 128), (3, 192, 6, 1, 256)])
def test_factored_table_kernel_odd_shapes(small_threshold, cin, cout, hw, stride, n_vecs):
    """Image sides that are not powers of two (ragged last strip of the pixel order), a strided conv, 1 / 2 / 3 row blocks per workgroup, ragged batches:
    the tap-table kernel against the oracle on the stored CSR, bit for bit."""
    dev = torch.device('cuda:0')
    layer = conv_layer(cin, cout, hw, stride=stride, zeros=[(1, 0, 0, 1)], seed=hw)
    W = layer.W
    assert isinstance(W, ksp.FactoredSparseMatrix)
    (ip, ix, dt) = ksp._stored_order_csr(W._matrix)
    rng = np.random.RandomState(hw)
    X = np.vstack((rng.randn(cin * hw * hw, n_vecs).astype(np.float32), np.ones((1, n_vecs), np.float32)))
    with torch.cuda.device(dev):
        assert 'taps> (factored operator' in W._device_op(dev).plan(n_vecs, 2)
    ref = oracle.csr_matvecs(W.shape, ip, ix, dt, X)
    y = W.torchdot(torch.as_tensor(X).to(dev), relu=False).cpu().numpy()
    assert np.array_equal(y, ref)


def test_factored_form_travels_with_saved_keynets(small_threshold, tmp_path):
    """save_keynet writes the factored device form next to the CSR (fact_* arrays); load_keynet PROVES it against the stored CSR again and only then hands
    it to the device -- a file whose CSR was edited loads as the plain container."""
    from keynet_amd import io as kio
    from keynet_amd import system as ksys
    layer = conv_layer(4, 32, 8, zeros=[(1, 2, 0, 0)])
    assert isinstance(layer.W, ksp.FactoredSparseMatrix)
    knet = ksys.KeyedModel.fromlayers({'conv': layer}, (32, 8, 8))
    f = str(tmp_path / 'k.npz')
    kio.save_keynet(knet, f)
    k2 = kio.load_keynet(f)
    W2 = k2.conv.W
    assert isinstance(W2, ksp.FactoredSparseMatrix) and W2.nnz() == layer.W.nnz()
    (a, b) = (ksp._stored_order_csr(layer.W._matrix), ksp._stored_order_csr(W2._matrix))
    assert all(np.array_equal(x, y) for (x, y) in zip(a, b))
    for k in ('taps', 'ent_out', 'ent_in', 'ent_tap', 'lastcol'):
        assert np.array_equal(layer.W._factored._taps[k], W2._factored._taps[k]), k
    # tamper with one stored value: the proof fails, the operator is the reference's plain CSR container
    z = dict(np.load(f, allow_pickle=False))
    key = [k for k in z if k.endswith('conv.data')][0]
    z[key] = z[key].copy()
    z[key][3] += 1.0
    f2 = str(tmp_path / 'k2.npz')
    np.savez(f2, **z)
    assert type(kio.load_keynet(f2).conv.W) is ksp.SparseMatrix
