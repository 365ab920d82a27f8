"""Split-K twin of a conv-taps operator with few output pixels and a long contraction (VGG-16 conv5_x: 196 pixels): every pixel's slots dealt to
two pseudo-pixels, the shares summed in order + bias + ReLU by dense_reduce_kernel.  Opt-in (KN_SPLITK=1 when the operator is created AND per call): the
same-process A/B on the real conv5_x layers measured it 1-10 % SLOWER than the unsplit launch, so nothing takes it by default.  Matrix-core contract only: against the order-preserving
path within the float-key tolerance, against the unsplit launch (KN_SPLITK=0) likewise; KN_FLAG_EXACT never takes it (bit-equal to the oracle
as before); kn_spmm_screen's max |Y| and concurrent use of one handle from two streams keep working."""
import os

import numpy as np
import pytest
import torch

import oracle
from keynet_amd import sparse as ksp
from keynet_amd import direct as kdirect
from keynet_amd import _capi

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def conv_operator(cin, cout, hw, seed=0, gain=False):
    rng = np.random.RandomState(seed)
    HW = hw * hw
    w = (rng.randn(cout, cin, 3, 3) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.randn(cout).astype(np.float32)
    (pi, po) = (rng.permutation(HW), rng.permutation(HW))
    (g_out, g_in) = ((rng.rand(HW) + 0.5).astype(np.float32), (rng.rand(HW) + 0.5).astype(np.float32))
    (eo, ei, et, ec) = ([], [], [], [])
    for (t, (_, S)) in enumerate(kdirect.shift_matrices((hw, hw), 3, 1)):
        S = S.tocoo()
        eo.append(po[S.row]); ei.append(pi[S.col]); et.append(np.full(S.nnz, t)); ec.append((g_out[po[S.row]] / g_in[pi[S.col]]).astype(np.float32))
    taps = np.stack([w[:, :, i, j] for i in range(3) for j in range(3)])
    lastcol = np.concatenate((np.repeat(b, HW), [1.0])).astype(np.float32)
    return ksp.Conv2dTiledMatrix.fromtaps((cin, hw, hw), (cout, hw, hw), taps, np.concatenate(eo).astype(np.int32), np.concatenate(ei).astype(np.int32),
                                          np.concatenate(et).astype(np.int32), np.concatenate(ec) if gain else None, lastcol)


@pytest.mark.parametrize('cin,cout,hw,n_vecs,gain', [(64, 128, 14, 256, False), (128, 256, 7, 128, False), (64, 192, 10, 384, True)])
def test_splitk_twin_within_tolerance_and_exact_path_untouched(cin, cout, hw, n_vecs, gain, monkeypatch):
    monkeypatch.setenv('KN_SPLITK', '1')                                  # the twin is built only on request
    W = conv_operator(cin, cout, hw, seed=cin + hw, gain=gain)
    rng = np.random.RandomState(1)
    X = np.vstack((rng.randn(cin * hw * hw, n_vecs).astype(np.float32), np.ones((1, n_vecs), np.float32)))
    xd = torch.as_tensor(X).to(dev())
    with torch.cuda.device(dev()):
        op = W._device_op(dev())
        plan1 = op.plan(n_vecs, _capi.KN_FLAG_RELU)
        monkeypatch.setenv('KN_SPLITK', '0')
        plan0 = op.plan(n_vecs, _capi.KN_FLAG_RELU)
        monkeypatch.delenv('KN_SPLITK')
        plan_default = op.plan(n_vecs, 0)
        plan_exact = op.plan(n_vecs, _capi.KN_FLAG_EXACT)
    assert 'dense_reduce_kernel' in plan1 and 'dense_reduce_kernel' not in plan0 and 'dense_reduce_kernel' not in plan_exact, (plan1, plan0, plan_exact)
    assert 'dense_reduce_kernel' not in plan_default                    # never by default
    M = W.rows_csr()
    ref = oracle.csr_matvecs(M.shape, M.indptr, M.indices, M.data.astype(np.float32), X)
    ye = W.torchdot(xd, relu=False, exact=True).cpu().numpy()
    assert np.array_equal(ye, ref)                                       # the order-preserving path never takes the twin
    scale = max(1.0, float(np.abs(ref).max()))
    for relu in (False, True):
        r = np.maximum(ref, 0) if relu else ref
        monkeypatch.setenv('KN_SPLITK', '1')
        slot = torch.zeros(1, device=dev())
        y1 = W.torchdot(xd, relu=relu, exact=False, absmax=slot)
        y1b = W.torchdot(xd, relu=relu, exact=False)
        monkeypatch.setenv('KN_SPLITK', '0')
        y0 = W.torchdot(xd, relu=relu, exact=False)
        monkeypatch.delenv('KN_SPLITK')
        assert float(np.abs(y1.cpu().numpy() - r).max()) <= 1e-5 * scale
        assert float((y1 - y0).abs().max()) <= 1e-5 * scale
        assert float(slot.item()) == float(y1.abs().max())
        assert torch.equal(y1, y1b)                                      # deterministic (ordered reduction, no atomics)
        assert torch.equal(y0, W.torchdot(xd, relu=relu, exact=False))   # the default is the unsplit launch


def test_splitk_twin_on_two_streams_and_column_windows(monkeypatch):
    """The partial sums live in one workspace PER STREAM: two half-batch windows of one activation block on two streams at once (what the
    overlapped forward does) give the same bits as one after the other."""
    monkeypatch.setenv('KN_SPLITK', '1')
    W = conv_operator(64, 128, 14, seed=3)
    n = 256
    x = torch.randn(64 * 196 + 1, n, device=dev())
    x[-1] = 1
    y_ref = W.torchdot(x, relu=True, exact=False)
    with torch.cuda.device(dev()):
        op = W._device_op(dev())
        (s0, s1) = (torch.cuda.Stream(), torch.cuda.Stream())
        y = torch.empty_like(y_ref)
        torch.cuda.synchronize()
        for rep in range(3):
            for (h, st) in enumerate((s0, s1)):
                op.spmm(x.data_ptr() + 4 * 128 * h, n, 128, y.data_ptr() + 4 * 128 * h, n, _capi.KN_FLAG_RELU, st.cuda_stream)
        torch.cuda.synchronize()
    # a 128-column window runs other tile instantiations than the 256-column launch: equal to rounding, and the two windows are independent
    scale = max(1.0, float(y_ref.abs().max()))
    assert float((y - y_ref).abs().max()) <= 1e-5 * scale
    with torch.cuda.device(dev()):
        y2 = torch.empty_like(y_ref)
        for h in range(2):
            op.spmm(x.data_ptr() + 4 * 128 * h, n, 128, y2.data_ptr() + 4 * 128 * h, n, _capi.KN_FLAG_RELU, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    assert torch.equal(y, y2)
