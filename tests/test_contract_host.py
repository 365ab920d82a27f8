"""Host-side pieces of the float-key contract and of the round-3 ABI additions that need no GPU: the operator factor of the error screen
(max_abs_rowsum from the factored form without expanding it), the contract states and their transitions, their persistence in the neutral
file format, and argument validation of kn_chain_create / kn_spmm_plan."""
import ctypes
import os
import numpy as np
import pytest
import torch

from keynet_amd import io as kio
from keynet_amd import sparse as ksp
from keynet_amd import system as ksys
from keynet_amd import _capi
from keynet_amd.layer import KeyedLayer, _contract
from nets import MiniNet, load_weights


def _mini(golden, factory=ksys.TiledPermutationKeynet, **kw):
    z = golden('mini_tiled_permutation.npz')
    net = load_weights(MiniNet(), z)
    np.random.seed(0)
    return factory((2, 16, 16), net, 4, **kw)


@pytest.mark.parametrize('direct', [False, True])
def test_max_abs_rowsum_equals_the_expanded_operator(golden, direct):
    import warnings
    z = golden('mini_tiled_orthogonal.npz')
    net = load_weights(MiniNet(), z)
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        (sensor, knet) = ksys.TiledOrthogonalKeynet((2, 16, 16), net, 4, direct=direct)
    n = 0
    for c in knet._keynet.children():
        if isinstance(c, KeyedLayer):
            M = c.W.tocsr() if isinstance(c.W, ksp.TiledMatrix) else c.W._matrix.tocsr()
            want = float(abs(M).sum(axis=1).max())
            got = c.W.max_abs_rowsum()
            # the factored form sums |coef| * |tap| per entry: an upper bound that equals the expansion's row sum unless two entries hit one
            # (output, input) pixel pair with opposite signs
            assert got >= want * (1 - 1e-6) and got <= want * 1.5 + 1e-6, (type(c.W).__name__, got, want)
            n += 1
    assert n == 5


def _float_mini(golden, **kw):
    """The same mini-net under a key family WITH float coefficients (block permutation + photometric gain)."""
    z = golden('mini_tiled_permutation.npz')
    net = load_weights(MiniNet(), z)
    np.random.seed(0)
    return ksys.Keynet((2, 16, 16), net, local_geometric='permutation', local_photometric='uniform_random_gain', beta=0.5, tileshape=(4, 4), blocksize=4, **kw)


def test_contract_states_and_transitions(golden):
    assert _contract(None, True) is True and _contract(None, False) == 'auto'
    assert _contract(True, False) is True and _contract(False, True) is False and _contract('auto', True) == 'auto' and _contract('bf16x3', True) == 'bf16x3'
    assert _contract('exact', False) is True and _contract('mfma', True) is False
    with pytest.raises(AssertionError):
        _contract('fast', True)
    # north_star: "bit-exact for the permutation-only key, within 1e-5 for float keyed layers" -- a tiled key-net whose keys are permutations
    # (TiledPermutationKeynet, TiledIdentityKeynet) is bit-exact BY DEFAULT; the matrix cores are an explicit opt-in (exact='auto' / False)
    for factory in (ksys.TiledPermutationKeynet, ksys.TiledIdentityKeynet):
        (_, kperm) = _mini(golden, factory=factory)
        assert all(c._exact is True for c in kperm._keynet.children() if isinstance(c, KeyedLayer)), factory.__name__
        assert kperm.contract_report()['undecided'] == []
    # float keys: decided per layer at the first forward
    (sensor, knet) = _float_mini(golden)
    layers = [c for c in knet._keynet.children() if isinstance(c, KeyedLayer)]
    assert all(c._exact == 'auto' for c in layers)
    assert set(knet.contract_report()['undecided']) == {'conv1', 'pool1', 'conv2', 'pool2', 'fc1'}
    knet.exact_mode(True)
    assert all(c._exact is True for c in layers)
    knet.exact_mode(False)
    assert all(c._exact is False for c in layers) and not any(c.screened() for c in layers)     # forced: the caller's responsibility, never screened
    knet.exact_mode('auto-bf16x3')
    assert all(c._exact == 'auto' and c._allow_bf16x3 for c in layers)
    knet.exact_mode(None)
    assert all(c._exact == 'auto' and not c._allow_bf16x3 for c in layers)
    # the opt-in on a permutation-only key-net, and back
    (_, kp4) = _mini(golden, exact='auto')
    assert all(c._exact == 'auto' for c in kp4._keynet.children() if isinstance(c, KeyedLayer))
    (_, kp5) = _mini(golden)
    kp5.exact_mode('auto')
    assert all(c._exact == 'auto' for c in kp5._keynet.children() if isinstance(c, KeyedLayer))
    kp5.exact_mode(None)
    assert all(c._exact is True for c in kp5._keynet.children() if isinstance(c, KeyedLayer))
    (_, kp) = ksys.PermutationKeynet((2, 16, 16), load_weights(MiniNet(), golden('mini_tiled_permutation.npz')))
    assert all(c._exact is True for c in kp._keynet.children() if isinstance(c, KeyedLayer))      # untiled: bit-exact by default
    (_, kf) = _mini(golden, exact=False)
    assert all(c._exact is False for c in kf._keynet.children() if isinstance(c, KeyedLayer))


def test_rescreen_rule():
    """KeyedLayer.rescreen: a calibrated decision covers inputs up to RESCREEN_FACTOR x the calibrated max |x|; NaN never passes."""
    c = KeyedLayer.__new__(KeyedLayer)
    torch.nn.Module.__init__(c)
    (c._exact, c._contract_record) = (False, {'decided': 'mfma', 'max_abs_x': 4.0})
    assert c.screened()
    assert not c.rescreen(4.0) and not c.rescreen(7.9) and not c.rescreen(0.0)
    assert c.rescreen(8.1) and c.rescreen(float('inf')) and c.rescreen(float('nan'))
    c._exact = True
    assert not c.screened()
    (c._exact, c._contract_record) = (False, None)
    assert not c.screened()                                        # forced onto the matrix cores: not a calibration decision


def test_contract_survives_the_neutral_file_format(golden, tmp_path):
    (sensor, knet) = _float_mini(golden)
    # pretend calibration decided: the DECISION and its evidence are saved (replicas loading one file run the same kernels), the declared
    # contract rides along, and recalibrate=True returns to it
    (knet.conv2._exact, knet.conv2._contract_record) = (False, {'decided': 'mfma', 'max_abs_x': 3.5, 'measured_mfma_vs_exact': 1e-7, 'tol': 1e-5})
    (knet.conv1._exact, knet.conv1._contract_record) = ('bf16x3', {'decided': 'bf16x3', 'max_abs_x': 2.0})
    (knet.fc1._exact, knet.fc1._contract_record) = (True, {'decided': 'exact', 'bound': 1.0})
    f = kio.save_keynet(knet, str(tmp_path / 'k.npz'), sensor=sensor)
    k2 = kio.load_keynet(f)
    assert k2.conv2._exact is False and k2.conv2.screened() and k2.conv2._contract_record['max_abs_x'] == 3.5
    assert k2.conv1._exact == 'bf16x3' and k2.conv1.screened()
    assert k2.fc1._exact is True and k2.pool1._exact == 'auto'
    assert all(getattr(k2, n)._exact_decl == 'auto' for n in ('conv1', 'conv2', 'fc1', 'pool1'))
    k2.exact_mode(None)
    assert k2.conv2._exact == 'auto'
    k2r = kio.load_keynet(f, recalibrate=True)
    assert all(getattr(k2r, n)._exact == 'auto' for n in ('conv1', 'conv2', 'fc1', 'pool1')) and not k2r.conv2.screened()
    (_, kx) = _mini(golden, exact=True)
    k3 = kio.load_keynet(kio.save_keynet(kx, str(tmp_path / 'x.npz')))
    assert k3.conv1._exact is True
    # older archives (a bool, or the string 'auto') still load; anything else is refused
    z = dict(np.load(f, allow_pickle=False))
    z['L.conv2.exact'] = np.array(True)
    del z['L.conv2.exact_decl']
    np.savez(str(tmp_path / 'old.npz'), **z)
    assert kio.load_keynet(str(tmp_path / 'old.npz')).conv2._exact is True
    z['L.conv2.exact'] = np.array('fastest')
    np.savez(str(tmp_path / 'bad.npz'), **z)
    with pytest.raises(ValueError, match='unknown arithmetic contract'):
        kio.load_keynet(str(tmp_path / 'bad.npz'))


def test_chain_and_plan_argument_validation():
    L = _capi.lib()
    h = ctypes.c_void_p()
    assert L.kn_chain_create(0, None, None, ctypes.byref(h)) == 1 and h.value is None          # KN_ERR_INVALID: no operators
    assert L.kn_chain_create(2, None, None, ctypes.byref(h)) == 1
    assert L.kn_chain_create(1, (ctypes.c_void_p * 1)(None), None, None) == 1                 # NULL out handle
    buf = ctypes.create_string_buffer(16)
    assert L.kn_spmm_plan(None, 4, 4, 4, 0, buf, 16) == 1                                     # NULL handle
    assert len(L.kn_last_error()) > 0


def test_keyed_model_pickles_without_device_state(golden):
    """test/test_keynet.py:106 pickles (sensor, knet); the device caches a forward leaves behind (operator handles, overlap plans, the whole-net
    kernel) must not travel."""
    import pickle
    (sensor, knet) = _mini(golden)
    knet.__dict__['_overlap_plans'] = {'x': object()}
    knet.__dict__['_chain_ops'] = {0: ((), ctypes.c_void_p(1))}          # what a forward would have cached (not picklable)
    (s2, k2) = pickle.loads(pickle.dumps((sensor, knet)))
    assert '_chain_ops' not in k2.__dict__ and '_overlap_plans' not in k2.__dict__
    assert k2.conv1._exact is True and tuple(k2.conv1.W.shape) == tuple(knet.conv1.W.shape) and k2._outshape == knet._outshape
    assert (s2._encryptkey != sensor._encryptkey).nnz == 0


def test_split_form_of_a_filled_in_conv_is_the_same_operator():
    """Conv2dTiledMatrix._split_arrays (host side of the split application): (channel mixing on Z) . (I_Cin (x) K) equals the fused factored
    operator -- checked on the expansions, in float64, on a small filled-in operator with several taps per pixel pair; the 'split' contract state round-trips
    through the neutral archive vocabulary; the cost rule offers the split only to operators whose estimate is under half the fused launch."""
    import scipy.sparse
    from keynet_amd.layer import CONTRACTS, contract_name
    rng = np.random.RandomState(3)
    (Cin, Cout, H, fill) = (3, 5, 4, 6)
    HW = H * H
    taps = rng.randn(9, Cout, Cin).astype(np.float32)
    (eo, ei, et, ec) = ([], [], [], [])
    for t in range(9):
        for o in range(HW):
            ins = rng.choice(HW, size=fill, replace=False)
            eo.append(np.full(fill, o)); ei.append(ins); et.append(np.full(fill, t)); ec.append(rng.randn(fill).astype(np.float32))
    lastcol = np.concatenate((rng.randn(Cout * HW), [1.0])).astype(np.float32)
    W = ksp.Conv2dTiledMatrix.fromtaps((Cin, H, H), (Cout, H, H), taps, np.concatenate(eo), np.concatenate(ei), np.concatenate(et), np.concatenate(ec), lastcol)
    assert abs(W.fill_factor() - fill) < 1e-12 and W.split_capable()
    (K, second) = W._split_arrays()
    assert K.shape == (9 * HW, HW) and K.nnz == 9 * HW * fill
    F = ksp.Conv2dTiledMatrix.fromtaps(**second)
    assert F.shape == (Cout * HW + 1, Cin * 9 * HW + 1) and F.fill_factor() == 1.0 and not F.split_capable()
    # Z = (I_Cin (x) K) X on the feature rows, the homogeneous row passed through
    spatial = scipy.sparse.block_diag([scipy.sparse.kron(scipy.sparse.identity(Cin), K.astype(np.float64)), scipy.sparse.identity(1)], format='csr')
    composed = (F.tosparse('csr').astype(np.float64) @ spatial).toarray()
    fused = W.tosparse('csr').astype(np.float64).toarray()
    assert composed.shape == fused.shape and np.allclose(composed, fused, rtol=1e-6, atol=1e-6)
    assert 'split' in CONTRACTS and contract_name('split') == 'split' and _contract('split', True) == 'split'
    assert kio._contract_from_array(np.array('split'), 'L.x.exact') == 'split'
    # the cost rule (keynet_amd.sparse.Conv2dTiledMatrix.split_capable): a VGG-sized layer with 2 entries per (pixel, tap) -- Givens-like -- stays fused, 60 entries are offered
    class Shape(ksp.Conv2dTiledMatrix):
        pass
    def offered(per_tap, n):
        S = Shape.__new__(Shape)
        (S._inshape, S._outshape) = ((64, 224, 224), (64, 224, 224))
        S._taps = dict(taps=np.zeros((9, 1, 1), np.float32), ent_out=np.zeros(9 * 224 * 224 * per_tap, np.int8))
        return S.split_capable(n)
    assert offered(60, 64) and offered(60, 256) and not offered(2, 256) and not offered(1, 256)


def test_host_expansion_sums_a_pairs_terms_in_entry_order():
    """Conv2dTiledMatrix.tosparse / rows_csr on a factored operator with up to six terms per (output, input) pixel pair: the stored value is the sequential float32 sum of the
    terms fl(coef * tap) in ENTRY order -- the definition the device kernels (convtaps_exact_fill_kernel, the generic kernel) and kn_export_csr implement; scipy's own duplicate
    summation sorts unstably and defines no order.  Checked against an explicit per-pair loop, on the whole operator and on a pixel subset."""
    rng = np.random.RandomState(11)
    (Cin, Cout, H) = (3, 7, 5)
    HW = H * H
    taps = (rng.randn(9, Cout, Cin) * np.array([1e-3, 1, 1e3, 1, 1e-2, 1, 1e2, 1, 1])[:, None, None]).astype(np.float32)      # wide dynamic range: the order of the sum shows
    (eo, ei, et, ec) = ([], [], [], [])
    for o in range(HW):
        for i in rng.choice(HW, size=rng.randint(1, 6), replace=False):
            for t in rng.choice(9, size=rng.randint(1, 7), replace=False):
                eo.append(o); ei.append(i); et.append(t); ec.append(np.float32(rng.randn()))
    (eo, ei, et, ec) = (np.array(eo, np.int32), np.array(ei, np.int32), np.array(et, np.int32), np.array(ec, np.float32))
    lastcol = np.concatenate((rng.randn(Cout * HW), [1.0])).astype(np.float32)
    W = ksp.Conv2dTiledMatrix.fromtaps((Cin, H, H), (Cout, H, H), taps, eo, ei, et, ec, lastcol)
    want = {}
    for e in range(len(eo)):                                                        # entry order
        term = (ec[e] * taps[et[e]]).astype(np.float32)
        key = (int(eo[e]), int(ei[e]))
        want[key] = term if key not in want else (want[key] + term).astype(np.float32)
    M = W.tosparse('csr')
    assert M.shape == W.shape and M.nnz == len(want) * Cout * Cin + int(np.count_nonzero(lastcol)) and M.has_sorted_indices
    D = M.toarray()
    for ((o, i), V) in want.items():
        assert np.array_equal(D[o + np.arange(Cout) * HW][:, i + np.arange(Cin) * HW], V), (o, i)
    assert np.array_equal(D[:, -1], lastcol)
    px = np.array([3, 11, 24])
    S = W.rows_csr(px).toarray()
    rows = (np.arange(Cout)[:, None] * HW + px[None, :]).ravel()
    assert np.array_equal(S, D[rows])
