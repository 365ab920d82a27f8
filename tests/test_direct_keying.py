"""Direct-to-tiled keying (keynet_amd/direct.py) builds the SAME operators as the reference route (Toeplitz -> SpGEMM ->
tiler), entry for entry, on the golden mini-nets.  CPU only."""
import numpy as np
import pytest

import keynet_amd.system as ksys
import keynet_amd.sparse as ksp
from keynet_amd.layer import KeyedLayer
from nets import MiniNet, load_weights


@pytest.mark.parametrize('tag,tilesize', [('identity', 4), ('permutation', 4), ('permutation8', 8)])
def test_direct_equals_reference_route(golden, tag, tilesize):
    z = golden('mini_tiled_%s.npz' % tag)
    net = load_weights(MiniNet(), z)
    np.random.seed(0)
    factory = ksys.TiledIdentityKeynet if tag == 'identity' else ksys.TiledPermutationKeynet
    (sensor, knet) = factory((2, 16, 16), net, tilesize, direct=True)
    for (name, child) in knet._keynet.named_children():
        if not isinstance(child, KeyedLayer):
            continue
        p = 'L.%s.' % name
        c = child.W.tocsr() if isinstance(child.W, ksp.TiledMatrix) else child.W._matrix.tocsr()
        c.sort_indices()
        kind = str(z[p + 'kind'])
        if kind == 'conv2dtiled':
            assert child.W._taps is not None, 'conv layer was not built in factored form'
            # the reference's tiled expansion holds the dense channel matrices incl. zero-padded taps at the image border:
            # compare after dropping explicit zeros on both sides
            import scipy.sparse
            ref = scipy.sparse.csr_matrix((z[p + 'data'], z[p + 'indices'], z[p + 'indptr']), shape=c.shape)
            ref.eliminate_zeros()
            c.eliminate_zeros()
            assert np.array_equal(c.indptr, ref.indptr) and np.array_equal(c.indices, ref.indices) and np.array_equal(c.data, ref.data), name
        elif kind == 'tiled':
            assert np.array_equal(c.indptr, z[p + 'indptr']) and np.array_equal(c.indices, z[p + 'indices']) and np.array_equal(c.data, z[p + 'data']), name
            assert np.array_equal(np.array(child.W._blocks), z[p + 'blocks']) and child.W.nnz() == int(z[p + 'nnz'])


def test_direct_refuses_global_permutation(golden):
    """A global permutation is not channel-replicated: the factored route must refuse, not silently mis-key."""
    from keynet_amd import direct
    z = golden('lenet_perm.npz')
    import scipy.sparse
    shape = tuple(int(v) for v in z['sensor.shape'])
    A = scipy.sparse.csr_matrix((z['sensor.enc.data'], z['sensor.enc.indices'], z['sensor.enc.indptr']), shape=shape)
    with pytest.raises((ValueError, AssertionError)):
        direct.spatial_key(scipy.sparse.block_diag((A[:-1, :-1], A[:-1, :-1].T, [[1]])).tocsr(), 2, 784)   # channel 1 keyed differently


def test_direct_equals_reference_route_float_keys(golden):
    """Orthogonal key family (Givens rotations + affine photometric key + block memory order): the factored operator
    (coefficients K_t = a_out S_t a_in^-1, last column M_out(Wc gamma + b) + beta_out) equals the reference route's
    A.W.Ainv up to f32 rounding of the SpGEMM sums (the reference accumulates the same products in another order)."""
    import warnings
    import scipy.sparse
    z = golden('mini_tiled_orthogonal.npz')
    net = load_weights(MiniNet(), z)
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        (sensor, knet) = ksys.TiledOrthogonalKeynet((2, 16, 16), net, 4, direct=True)
    for (name, child) in knet._keynet.named_children():
        if not isinstance(child, KeyedLayer):
            continue
        p = 'L.%s.' % name
        kind = str(z[p + 'kind'])
        if kind == 'conv2dtiled':
            assert child.W._taps is not None and child.W._taps['ent_coef'] is not None     # float coefficients
        c = child.W.tocsr() if isinstance(child.W, ksp.TiledMatrix) else child.W._matrix.tocsr()
        ref = scipy.sparse.csr_matrix((z[p + 'data'], z[p + 'indices'], z[p + 'indptr']), shape=c.shape)
        (D, R) = (np.asarray(c.todense(), dtype=np.float64), np.asarray(ref.todense(), dtype=np.float64))
        scale = np.abs(R).max()
        assert np.abs(D - R).max() <= 2e-6 * scale, (name, np.abs(D - R).max(), scale)


def test_direct_equals_reference_route_filled_in_keys(golden):
    """The doubly-stochastic family (test/test_keynet.py:116-129): one (output pixel, input pixel) pair is hit by several taps, so ONE stored entry of
    the reference's matrix is a sum of terms coef * tap that scipy's SpGEMM (keynet/layer.py:35) accumulates in ITS order.  The factored operator holds
    the terms (and forms the entry as their f32 sum in entry order): equal to the reference's stored values up to that re-association, 2e-6 of the
    largest entry -- NOT bit for bit -- and with the same stored structure (every reference entry is present, no extra ones beyond exact zeros)."""
    import sys, os, warnings
    import scipy.sparse
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from keygen_case_table import STOCHASTIC_KW
    z = golden('mini_tiled_stochastic.npz')
    net = load_weights(MiniNet(), z)
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        (sensor, knet) = ksys.Keynet((2, 16, 16), net, direct=True, **STOCHASTIC_KW)
    multi = 0
    for (name, child) in knet._keynet.named_children():
        if not isinstance(child, KeyedLayer):
            continue
        p = 'L.%s.' % name
        kind = str(z[p + 'kind'])
        c = child.W.tocsr() if isinstance(child.W, ksp.TiledMatrix) else child.W._matrix.tocsr()
        ref = scipy.sparse.csr_matrix((z[p + 'data'], z[p + 'indices'], z[p + 'indptr']), shape=c.shape)
        (D, R) = (np.asarray(c.todense(), dtype=np.float64), np.asarray(ref.todense(), dtype=np.float64))
        scale = np.abs(R).max()
        assert np.abs(D - R).max() <= 2e-6 * scale, (name, np.abs(D - R).max(), scale)
        if kind == 'conv2dtiled':
            t = child.W._taps
            assert t is not None and t['ent_coef'] is not None
            # several taps per pixel pair: more (pixel, tap) entries than distinct (output pixel, input pixel) pairs
            pairs = len(set(zip(t['ent_out'].tolist(), t['ent_in'].tolist())))
            multi += int(len(t['ent_out']) > pairs)
            # rows_csr (what the GPU tests hand the oracle for a direct operator) against the reference's own stored rows
            (Cout, Hout, Wout) = child.W._outshape
            pix = np.arange(0, Hout * Wout, 7)
            S = child.W.rows_csr(pix)
            for co in range(Cout):
                mine = np.asarray(S[co * len(pix):(co + 1) * len(pix)].todense(), dtype=np.float64)
                assert np.abs(mine - R[co * Hout * Wout + pix]).max() <= 2e-6 * scale, (name, co)
    assert multi == 2, 'the fixture is meant to exercise pixel pairs hit by several taps'
