"""The arithmetic contract on the device (north_star: "bit-exact for the permutation-only key and within 1e-5 for float keyed layers"),
held on EVERY batch: defaults per key family, kn_spmm_screen / kn_absmax against numpy, the per-forward re-screen that sends a layer back
to calibration when its input outgrows the calibrated magnitude (eager, overlapped and HIP-graph forwards), and persistence of decisions."""
import logging
import os

import numpy as np
import pytest
import torch

import oracle
from keynet_amd import io as kio
from keynet_amd import sparse as ksp
from keynet_amd import system as ksys
from keynet_amd import _capi
from keynet_amd.layer import KeyedLayer, FLOAT_KEY_TOL, gate
from nets import MiniNet, load_weights

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    return torch.device('cuda:0')


def keyed(knet):
    return [(n, c) for (n, c) in knet._keynet.named_children() if isinstance(c, KeyedLayer)]


def mini_net(golden):
    return load_weights(MiniNet(), golden('mini_tiled_permutation.npz'))


def gain_keynet(golden, **kw):
    """Float keys on the mini-net: block permutation + block-local photometric gain (every keyed entry carries a_out[o] / a_in[i])."""
    np.random.seed(0)
    return ksys.Keynet((2, 16, 16), mini_net(golden), local_geometric='permutation', local_photometric='uniform_random_gain', beta=0.5, tileshape=(4, 4), blocksize=4, **kw)


def layerwise_within_tolerance(knet, xc):
    """Every keyed layer's shipped output (on the layer input the shipped forward produces) against the order-preserving kernel -- which is
    bit-exact with the reference's arithmetic (test_parity_gpu.py) -- on that same input: max over layers and ELEMENTS of diff / (1e-5 + 1e-5 |y|)."""
    children = list(knet._keynet.named_children())
    y = xc
    worst = 0.0
    i = 0
    while i < len(children):
        (name, c) = children[i]
        if isinstance(c, KeyedLayer):
            fuse = (i + 1 < len(children)) and isinstance(children[i + 1][1], torch.nn.ReLU)
            out = c.forward(y, fuse_relu=fuse)
            ref = c.W.torchdot(y.t(), relu=(fuse or c.iskeyedrelu()), exact=True).t()
            worst = max(worst, gate(out, ref)[0])                     # element-wise |d| / (1e-5 + 1e-5 |ref|): <= 1 is np.allclose(atol=1e-5)
            y = out
            i += 2 if fuse else 1
        else:
            y = torch.relu(y)
            i += 1
    return worst


def test_permutation_only_tiled_keynet_is_bit_exact_by_default(golden):
    """TiledPermutationKeynet(...).forward with NO contract argument == the reference's logits, bit for bit (it used to take the matrix
    cores by default: <= 1.4e-6 off).  exact='auto' is the explicit opt-in to the matrix cores; both keying routes."""
    z = golden('mini_tiled_permutation.npz')
    for direct in (False, True):
        np.random.seed(0)
        (sensor, knet) = ksys.TiledPermutationKeynet((2, 16, 16), mini_net(golden), 4, direct=direct)
        assert all(c._exact is True for (_, c) in keyed(knet))
        xc = sensor.fromtensor(torch.as_tensor(z['x_plain']).to(dev())).encrypt().astensor()
        assert np.array_equal(xc.cpu().numpy(), z['x_cipher'])
        y = knet.forward(xc).reshape(4, 10).cpu().numpy()
        assert np.array_equal(y, z['logits_keyed']), float(np.abs(y - z['logits_keyed']).max())
        full = knet.forward_linear(xc).cpu().numpy()
        assert np.array_equal(full, z['Y.fc1'])
    np.random.seed(0)
    (sensor, kauto) = ksys.TiledPermutationKeynet((2, 16, 16), mini_net(golden), 4, exact='auto')
    ya = kauto.forward_linear(xc).cpu().numpy()
    assert float(np.abs(ya - z['Y.fc1']).max()) <= 1e-5 * max(1.0, float(np.abs(z['Y.fc1']).max()))
    rep = kauto.contract_report()
    assert not rep['undecided'] and not rep['switched'] and any(r['screened'] for r in rep['layers'])      # conv layers on the matrix cores, screened


@pytest.mark.parametrize('kind', ['csr', 'csr-pool-256', 'csr-pool-half-wave', 'csr-grouped', 'conv-mfma-256', 'conv-mfma-128-window', 'conv-mfma-narrow', 'conv-exact', 'dense'])
def test_spmm_screen_reports_max_abs_output(kind):
    """kn_spmm_screen: same Y as kn_spmm bit for bit, and the slot holds max |Y| exactly -- folded into the tile stores of the matrix-core
    kernels (whole 256-column tiles), one reduction pass behind every other kernel; the slot is only ever raised."""
    rng = np.random.RandomState(7)
    if kind == 'csr':
        import scipy.sparse
        M = scipy.sparse.random(300, 200, density=0.05, random_state=rng, format='csr', dtype=np.float32)
        (W, n, exact) = (ksp.SparseMatrix(M), 48, True)
    elif kind in ('csr-pool-256', 'csr-pool-half-wave'):
        # keyed-pooling shape: thousands of loose rows of ~9 entries (the row kernels fold max |y| into their epilogue: no extra pass)
        import scipy.sparse
        (m, nc) = (6000, 2500)
        lens = rng.randint(6, 12, m)
        lens[::97] = 0
        ip = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
        M = scipy.sparse.csr_matrix((rng.randn(int(ip[-1])).astype(np.float32), rng.randint(0, nc, int(ip[-1])).astype(np.int32), ip), shape=(m, nc))
        (W, n, exact) = (ksp.SparseMatrix(M), 256 if kind == 'csr-pool-256' else 128, True)
    elif kind == 'csr-grouped':
        import scipy.sparse
        pat = rng.randint(0, 500, 40).astype(np.int32)
        rows_ = [pat] * 64 + [rng.randint(0, 500, 7).astype(np.int32) for _ in range(30)]
        ip = np.concatenate(([0], np.cumsum([len(r) for r in rows_]))).astype(np.int32)
        M = scipy.sparse.csr_matrix((rng.randn(int(ip[-1])).astype(np.float32), np.concatenate(rows_), ip), shape=(len(rows_), 500))
        (W, n, exact) = (ksp.SparseMatrix(M), 256, True)
    elif kind == 'dense':
        D = rng.randn(1025, 1025).astype(np.float32)
        D[-1, :] = 0
        D[-1, -1] = 1
        (W, n, exact) = (ksp.SparseMatrix(D), 256, False)
        assert W._dense_device_op(dev()) is not None
    else:
        from keynet_amd import direct as kdirect
        (cin, cout, hw) = (16, 128 if kind == 'conv-mfma-128-window' else 64, 6)      # (128 x 128 tiles: a half-batch window of 128 columns keeps max |y| in the store epilogue)
        w = (rng.randn(cout, cin, 3, 3) / 12).astype(np.float32)
        b = rng.randn(cout).astype(np.float32)
        (eo, ei, et) = ([], [], [])
        for (t, (_, S)) in enumerate(kdirect.shift_matrices((hw, hw), 3, 1)):
            S = S.tocoo()
            eo.append(S.row); ei.append(S.col); et.append(np.full(S.nnz, t))
        taps = np.stack([w[:, :, i, j] for i in range(3) for j in range(3)])
        W = ksp.Conv2dTiledMatrix.fromtaps((cin, hw, hw), (cout, hw, hw), taps, np.concatenate(eo).astype(np.int32), np.concatenate(ei).astype(np.int32),
                                           np.concatenate(et).astype(np.int32), None, np.concatenate((np.repeat(b, hw * hw), [1.0])).astype(np.float32))
        (n, exact) = ({'conv-mfma-256': 256, 'conv-mfma-128-window': 128, 'conv-mfma-narrow': 24, 'conv-exact': 256}[kind], kind == 'conv-exact')
    X = rng.randn(W.shape[1], n).astype(np.float32) * 3
    X[-1] = 1
    xd = torch.as_tensor(X).to(dev())
    for relu in (False, True):
        slot = torch.zeros(1, device=dev())
        y0 = W.torchdot(xd, relu=relu, exact=exact)
        y1 = W.torchdot(xd, relu=relu, exact=exact, absmax=slot)
        assert torch.equal(y0, y1)
        assert float(slot.item()) == float(y1.abs().max()), (kind, relu, float(slot.item()), float(y1.abs().max()))
        slot.fill_(1e30)                                   # raised only: a larger value already there stays
        W.torchdot(xd, relu=relu, exact=exact, absmax=slot)
        assert float(slot.item()) == pytest.approx(1e30)
    if kind == 'conv-mfma-256':
        with torch.cuda.device(dev()):
            assert 'absmax' not in W._device_op(dev()).plan(256, 0)       # folded into the store epilogue: no extra launch
    if kind == 'csr-pool-half-wave':
        with torch.cuda.device(dev()):
            assert 'csr_rows_pair_kernel' in W._device_op(dev()).plan(128, 2)


def test_absmax_ignores_nan_and_counts_inf():
    x = torch.randn(37, 24, device=dev())
    slot = torch.zeros(1, device=dev())
    with torch.cuda.device(dev()):
        _capi.absmax(x.data_ptr(), 37, 24, 24, slot.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert float(slot.item()) == float(x.abs().max())
        x[3, 5] = float('nan')
        slot.zero_()
        _capi.absmax(x.data_ptr(), 37, 24, 24, slot.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert float(slot.item()) == float(torch.nan_to_num(x, nan=0.0).abs().max())
        x[7, 1] = float('-inf')
        _capi.absmax(x.data_ptr(), 37, 24, 24, slot.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert float(slot.item()) == float('inf')
        # a column window of a wider block (ld > n_vecs)
        w = torch.randn(10, 64, device=dev())
        slot.zero_()
        _capi.absmax(w.data_ptr() + 4 * 16, 10, 64, 8, slot.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert float(slot.item()) == float(w[:, 16:24].abs().max())


@pytest.mark.parametrize('family', ['gain', 'orthogonal'])
@pytest.mark.parametrize('batch', [8, 256])
def test_contract_holds_when_later_batches_are_larger(golden, family, batch, caplog):
    """Calibrate a float-key mini-net on N(0,1) images, then feed 100x larger ones: every layer must still be within 1e-5 max(1, |y|) of the
    reference's arithmetic.  The one-shot calibration of round 3 never looked at max |x| again; now each forward gathers max |x| per
    calibrated layer on the device, and a layer whose input outgrew its calibration by more than 2x is re-calibrated on that batch.
    batch 256 = whole tiles (max |y| folded into the matrix-core epilogue) and the overlapped two-stream forward; 8 = the narrow path."""
    import warnings
    if family == 'gain':
        (sensor, knet) = gain_keynet(golden)
    else:
        np.random.seed(0)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            (sensor, knet) = ksys.TiledOrthogonalKeynet((2, 16, 16), mini_net(golden), 4)
    assert all(c._exact == 'auto' for (_, c) in keyed(knet))
    g = torch.Generator(device=dev()).manual_seed(5)
    x = torch.randn((batch, 2, 16, 16), generator=g, device=dev())
    xc = sensor.fromtensor(x).encrypt().astensor()
    knet.forward_linear(xc)
    rep0 = knet.contract_report()
    assert not rep0['undecided'] and rep0['recalibrations'] == 0
    cal = {r['name']: r['calibration']['max_abs_x'] for r in rep0['layers'] if r['screened']}
    assert cal or family == 'orthogonal', rep0
    assert layerwise_within_tolerance(knet, xc) <= 1.0
    # same magnitude again: nothing is re-decided
    knet.forward_linear(xc)
    assert knet.contract_report()['recalibrations'] == 0
    big = sensor.fromtensor(100.0 * x).encrypt().astensor()
    with caplog.at_level(logging.INFO, logger='keynet_amd'):
        yb = knet.forward_linear(big)
    rep1 = knet.contract_report()
    assert not rep1['undecided']
    if cal:
        assert rep1['recalibrations'] >= 1 and 're-calibrating' in caplog.text
        for r in rep1['layers']:
            if r['screened'] and r['name'] in cal:
                assert r['calibration']['max_abs_x'] > 2.0 * cal[r['name']], r         # decided again, on the large batch
    assert layerwise_within_tolerance(knet, big) <= 1.0
    # decided: the same batch again gives the same bits, whatever path (overlapped / plain) it takes
    assert torch.equal(yb, knet.forward_linear(big)) and torch.equal(yb, knet.forward_linear(big, overlap=False))
    # a layer that any batch moved to the reference's order stays there (conservative and sticky: a key-net calibrated from scratch on the
    # large batch may keep such a layer on the matrix cores); either way the contract holds
    was_exact = set(n for (n, c) in keyed(knet) if c._exact is True)
    knet.exact_mode('auto')
    knet.forward_linear(big)
    assert layerwise_within_tolerance(knet, big) <= 1.0
    assert set(n for (n, c) in keyed(knet) if c._exact is True) <= was_exact
    # and back to small inputs: decisions taken on larger activations cover smaller ones, no churn
    n0 = knet.contract_report()['recalibrations']
    knet.forward_linear(xc)
    assert knet.contract_report()['recalibrations'] == n0


def test_host_tensors_are_screened_too(golden):
    """The reference's users hand host tensors over; the forward moves them to the device once, so the screen covers that route as well."""
    (sensor, knet) = gain_keynet(golden)
    g = torch.Generator().manual_seed(4)
    x = torch.randn((8, 2, 16, 16), generator=g)
    xc = sensor.fromtensor(x.to(dev())).encrypt().astensor().cpu()
    y = knet.forward_linear(xc)
    assert not y.is_cuda and knet.contract_report()['recalibrations'] == 0 and any(r['screened'] for r in knet.contract_report()['layers'])
    yb = knet.forward_linear(200.0 * xc)
    assert not yb.is_cuda and knet.contract_report()['recalibrations'] >= 1
    assert torch.equal(yb, knet.forward_linear((200.0 * xc).to(dev())).cpu())


def test_rescreen_switch_and_forced_modes(golden, monkeypatch):
    (sensor, knet) = gain_keynet(golden)
    g = torch.Generator(device=dev()).manual_seed(6)
    x = torch.randn((8, 2, 16, 16), generator=g, device=dev())
    xc = sensor.fromtensor(x).encrypt().astensor()
    knet.forward_linear(xc)
    assert knet.contract_report()['rescreen'] is True
    big = sensor.fromtensor(1000.0 * x).encrypt().astensor()
    monkeypatch.setenv('KN_NO_RESCREEN', '1')                # A/B switch: the round-3 behaviour
    knet.forward_linear(big)
    assert knet.contract_report()['recalibrations'] == 0 and knet.contract_report()['rescreen'] is False
    monkeypatch.delenv('KN_NO_RESCREEN')
    knet.exact_mode(False)                                   # forced onto the matrix cores: the caller's responsibility, not screened
    knet.forward_linear(big)
    assert knet.contract_report()['recalibrations'] == 0 and not any(r['screened'] for r in knet.contract_report()['layers'])
    knet.exact_mode(True)
    y = knet.forward_linear(big)
    assert knet.contract_report()['rescreen'] is False and bool(torch.isfinite(y).all())


def test_graph_replay_keeps_the_screen(golden):
    """A captured forward gathers max |x| like the eager one; replay() checks it after the launch and, when a layer's input has outgrown
    its calibration, re-calibrates eagerly on that batch and captures a new graph."""
    (sensor, knet) = gain_keynet(golden)
    g = torch.Generator(device=dev()).manual_seed(8)
    x = torch.randn((256, 2, 16, 16), generator=g, device=dev())
    xc = sensor.fromtensor(x).encrypt().astensor()
    replay = knet.capture(xc)
    y = replay(xc).clone()
    assert torch.equal(y, knet.forward_linear(xc, overlap=False))
    g0 = replay.graph
    assert knet.contract_report()['recalibrations'] == 0
    big = sensor.fromtensor(50.0 * x).encrypt().astensor()
    yb = replay(big).clone()
    assert knet.contract_report()['recalibrations'] >= 1 and replay.graph is not g0
    assert torch.equal(yb, knet.forward_linear(big, overlap=False))
    assert layerwise_within_tolerance(knet, big) <= 1.0
    g1 = replay.graph
    assert torch.equal(replay(xc), y) or layerwise_within_tolerance(knet, xc) <= 1.0
    assert replay.graph is g1                                # smaller inputs: same graph


def test_capture_refuses_an_undecided_layer(golden):
    (sensor, knet) = gain_keynet(golden)
    x = torch.randn((8, 2, 16, 16), device=dev())
    xc = sensor.fromtensor(x).encrypt().astensor()
    c = dict(keyed(knet))['conv1']
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with pytest.raises(_capi.KeynetHipError, match='has not decided its arithmetic contract'):
        with torch.cuda.graph(graph, stream=side):
            c.forward(xc)
    torch.cuda.synchronize()


def test_saved_decisions_travel(golden, tmp_path):
    """Every loader of one saved key-net runs the same kernels: the decisions and their calibration records are in the file."""
    (sensor, knet) = gain_keynet(golden)
    g = torch.Generator(device=dev()).manual_seed(9)
    xc = sensor.fromtensor(torch.randn((8, 2, 16, 16), generator=g, device=dev())).encrypt().astensor()
    y = knet.forward_linear(xc)
    f = kio.save_keynet(knet, str(tmp_path / 'k.npz'))
    k2 = kio.load_keynet(f)
    assert {n: c._exact for (n, c) in keyed(k2)} == {n: c._exact for (n, c) in keyed(knet)}
    assert {n: c.screened() for (n, c) in keyed(k2)} == {n: c.screened() for (n, c) in keyed(knet)}
    assert torch.equal(k2.forward_linear(xc), y) and k2.contract_report()['recalibrations'] == 0
    k3 = kio.load_keynet(f, recalibrate=True)
    assert all(c._exact == 'auto' for (_, c) in keyed(k3))


def test_reserve_workspace_makes_a_first_dense_call_capturable():
    rng = np.random.RandomState(3)
    D = rng.randn(1025, 1025).astype(np.float32)
    D[-1, :] = 0
    D[-1, -1] = 1
    W = ksp.SparseMatrix(D)
    op = W._dense_device_op(dev())
    assert op is not None
    x = torch.randn(1025, 256, device=dev())
    y = torch.empty(1025, 256, device=dev())
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.device(dev()):
        op.reserve_workspace(256, side.cuda_stream)          # instead of an eager warm-up launch on the capture stream
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            op.spmm(x.data_ptr(), 256, 256, y.data_ptr(), 256, 0, torch.cuda.current_stream().cuda_stream)
        graph.replay()
        torch.cuda.synchronize()
    ref = W.torchdot(x, exact=False)
    assert torch.equal(y, ref)


def test_an_empty_first_batch_decides_nothing(golden):
    """A rank whose shard of a small batch is empty calls the key-net with zero images before it has seen any: the layers whose contract is still 'auto' must not calibrate
    on nothing (it used to raise from max() of an empty tensor) -- they return the empty block and decide on the first batch that holds an image."""
    z = golden('mini_tiled_stochastic.npz')
    knet = kio.keynet_from_arrays(z, recalibrate=True)
    x = torch.as_tensor(z['x_cipher']).to(dev())
    assert knet.contract_report()['undecided']
    y0 = knet.forward_linear(x[:0])
    assert tuple(y0.shape) == (0, 11) and knet.contract_report()['undecided']
    y = knet.forward_linear(x)
    assert not knet.contract_report()['undecided'] and y.shape[0] == x.shape[0]
    assert tuple(knet.forward_linear(x[:0]).shape) == (0, 11)
