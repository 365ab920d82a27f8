"""KN_FLAG_EXACT on a factored conv operator with the products on the matrix pipe (convtaps_exact_mfma_kernel: K = 1 matrix instruction with a zero
accumulator = the IEEE-rounded product; sums on the vector ALU in the expansion's column order) against the CPU oracle on the expanded operator,
bit for bit, and against the vector-ALU pipeline (the default: the matrix-pipe variant is opt-in, KN_EXACT_MFMA=1, because it measured 4-8 % slower on
the keyed VGG-16 layers): unit and coefficient entries, 1 / 2 / 3 channel blocks per
workgroup, contractions shorter than the six columns in flight and every remainder modulo six, batches that are not a multiple of 64 or 256,
explicit-zero bias entries, Inf / NaN activations."""
import os

import numpy as np
import pytest
import torch

import oracle
from keynet_amd import _capi
from test_splitk_gpu import conv_operator

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


@pytest.mark.parametrize('cin,cout,hw,n_vecs,gain', [(16, 64, 10, 256, False), (3, 64, 12, 512, False), (20, 96, 8, 320, True), (32, 128, 7, 260, False),
                                                      (1, 32, 9, 256, True), (64, 256, 5, 1000, False), (5, 160, 6, 256, True)])
def test_exact_conv_on_the_matrix_pipe_vs_oracle(cin, cout, hw, n_vecs, gain, monkeypatch):
    W = conv_operator(cin, cout, hw, seed=cin * 7 + cout, gain=gain)
    W._taps['lastcol'][::5] = 0.0                                    # absent bias entries (explicit zeros do not survive keying)
    rng = np.random.RandomState(2)
    X = np.vstack((rng.randn(cin * hw * hw, n_vecs).astype(np.float32), np.ones((1, n_vecs), np.float32)))
    X[3, 5] = np.inf
    X[min(7, cin * hw * hw - 1), 70 % n_vecs] = np.nan
    xd = torch.as_tensor(X).to(dev())
    with torch.cuda.device(dev()):
        assert 'convtaps_exact_mfma_kernel' not in W._device_op(dev()).plan(n_vecs, _capi.KN_FLAG_EXACT)     # never by default
    monkeypatch.setenv('KN_EXACT_MFMA', '1')
    with torch.cuda.device(dev()):
        plan = W._device_op(dev()).plan(n_vecs, _capi.KN_FLAG_EXACT)
    assert 'convtaps_exact_mfma_kernel' in plan, plan
    M = W.rows_csr()
    with np.errstate(all='ignore'):
        ref = oracle.csr_matvecs(M.shape, M.indptr, M.indices, M.data.astype(np.float32), X)
    for nrb in ('1', '2', '3'):
        monkeypatch.setenv('KN_EXACT_MFMA_NRB', nrb)
        for relu in (False, True):
            with np.errstate(all='ignore'):
                r = np.where(ref < 0, np.float32(0), ref) if relu else ref
            y = W.torchdot(xd, relu=relu, exact=True).cpu().numpy()
            assert np.array_equal(y, r, equal_nan=True), (nrb, relu, int(np.sum(~((y == r) | (np.isnan(y) & np.isnan(r))))))
    monkeypatch.delenv('KN_EXACT_MFMA_NRB')
    monkeypatch.delenv('KN_EXACT_MFMA')
    y0 = W.torchdot(xd, relu=False, exact=True).cpu().numpy()
    assert np.array_equal(y0, ref, equal_nan=True)


def test_exact_conv_matrix_pipe_on_a_column_window(monkeypatch):
    monkeypatch.setenv('KN_EXACT_MFMA', '1')
    W = conv_operator(16, 64, 8, seed=4)
    n = 16 * 64 + 1
    ld = 512
    x = torch.randn(n, ld, device=dev())
    x[-1] = 1
    y = torch.full((W.shape[0], ld), 7.0, device=dev())
    with torch.cuda.device(dev()):
        W._device_op(dev()).spmm(x.data_ptr() + 4 * 128, ld, 256, y.data_ptr() + 4 * 128, ld, _capi.KN_FLAG_EXACT | _capi.KN_FLAG_RELU, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    M = W.rows_csr()
    ref = np.maximum(oracle.csr_matvecs(M.shape, M.indptr, M.indices, M.data.astype(np.float32), np.ascontiguousarray(x[:, 128:384].cpu().numpy())), 0)
    got = y.cpu().numpy()
    assert np.array_equal(got[:, 128:384], ref) and np.all(got[:, :128] == 7.0) and np.all(got[:, 384:] == 7.0)


@pytest.mark.parametrize('groups', ['1', '2', '4', '8'])
def test_exact_pipeline_channel_bundle_groups(groups, monkeypatch):
    """convtaps_exact_pipe_kernel deals (channel-bundle group, pixel, bundle) work items so that an XCD keeps its slice of the taps in L2 (the
    default on operators with more than 1 MB of taps: VGG-16 conv3_x .. conv5_x); any grouping is the same arithmetic: bit-equal to the oracle."""
    monkeypatch.setenv('KN_EXACT_COB_GROUPS', groups)
    for (cin, cout, hw, gain) in ((16, 128, 9, False), (8, 256, 5, True)):
        W = conv_operator(cin, cout, hw, seed=cin + int(groups), gain=gain)
        rng = np.random.RandomState(3)
        n_vecs = 512
        X = np.vstack((rng.randn(cin * hw * hw, n_vecs).astype(np.float32), np.ones((1, n_vecs), np.float32)))
        xd = torch.as_tensor(X).to(dev())
        with torch.cuda.device(dev()):
            assert 'convtaps_exact_pipe_kernel' in W._device_op(dev()).plan(n_vecs, _capi.KN_FLAG_EXACT)
        M = W.rows_csr()
        ref = oracle.csr_matvecs(M.shape, M.indptr, M.indices, M.data.astype(np.float32), X)
        y = W.torchdot(xd, relu=True, exact=True).cpu().numpy()
        assert np.array_equal(y, np.maximum(ref, 0)), (groups, cin, cout)


def test_exact_pipeline_default_grouping_on_a_large_tap_matrix():
    """More than 1 MB of taps (here 256 x 256 channels x 9 taps = 2.4 MB): the default rule groups the channel bundles; bit-equal to the oracle."""
    W = conv_operator(256, 256, 4, seed=11)
    rng = np.random.RandomState(5)
    n_vecs = 256
    X = np.vstack((rng.randn(256 * 16, n_vecs).astype(np.float32), np.ones((1, n_vecs), np.float32)))
    M = W.rows_csr()
    ref = oracle.csr_matvecs(M.shape, M.indptr, M.indices, M.data.astype(np.float32), X)
    y = W.torchdot(torch.as_tensor(X).to(dev()), relu=False, exact=True).cpu().numpy()
    assert np.array_equal(y, ref)
