"""Host-side fuzzer (CPU, no GPU): the direct-to-tiled keying route (keynet_amd/direct.py: operators built in factored form) against the reference route
(Toeplitz -> SpGEMM -> tiler: keynet/layer.py:35, keynet/sparse.py:543-571) on random small source networks under the same drawn keys -- permutation keys entry for
entry, the orthogonal family to the f32 rounding of the SpGEMM sums -- and a save / load round trip of every key-net."""
import os
import tempfile
import warnings

import numpy as np
import scipy.sparse
import torch

import keynet_amd.sparse as ksp
import keynet_amd.system as ksys
from keynet_amd import io as kio
from keynet_amd.layer import KeyedLayer
from fuzz_nets import random_net


def _csr(W):
    c = W.tocsr() if isinstance(W, ksp.TiledMatrix) else W._matrix.tocsr()
    c = scipy.sparse.csr_matrix(c)
    c.sum_duplicates()
    c.eliminate_zeros()                         # (the reference's tiled expansion stores the zero-padded taps at the image border)
    c.sort_indices()
    return c


def _key(factory, inshape, net, tile, seed, direct):
    np.random.seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return factory(inshape, net, tile, direct=direct)


def test_direct_route_equals_reference_route_on_random_networks():
    rng = np.random.RandomState(2718)
    (n_perm, n_float, n_conv_factored) = (0, 0, 0)
    for case in range(60):
        torch.manual_seed(int(rng.randint(1 << 30)))
        (net, inshape, names) = random_net(rng, sides=(8, 16))
        tile = int(rng.choice([2, 4, 8]))
        float_keys = bool(rng.rand() < 0.35)
        factory = (lambda i, n, t, direct: ksys.TiledOrthogonalKeynet(i, n, 4, direct=direct)) if float_keys else ksys.TiledPermutationKeynet
        seed = int(rng.randint(1 << 30))
        (_, ka) = _key(factory, inshape, net, tile, seed, True)
        (_, kb) = _key(factory, inshape, net, tile, seed, False)
        la = [(n, c) for (n, c) in ka._keynet.named_children() if isinstance(c, KeyedLayer)]
        lb = [(n, c) for (n, c) in kb._keynet.named_children() if isinstance(c, KeyedLayer)]
        assert [n for (n, _) in la] == [n for (n, _) in lb]
        for ((name, a), (_, b)) in zip(la, lb):
            (ca, cb) = (_csr(a.W), _csr(b.W))
            assert ca.shape == cb.shape, (case, name)
            n_conv_factored += isinstance(a.W, ksp.Conv2dTiledMatrix) and a.W._taps is not None
            if float_keys:
                (D, R) = (np.asarray(ca.todense(), dtype=np.float64), np.asarray(cb.todense(), dtype=np.float64))
                scale = max(np.abs(R).max(), 1e-30)
                assert np.abs(D - R).max() <= 2e-6 * scale, (case, name, names, np.abs(D - R).max(), scale)
            else:
                assert np.array_equal(ca.indptr, cb.indptr) and np.array_equal(ca.indices, cb.indices) and np.array_equal(ca.data, cb.data), (case, name, names, inshape, tile)
        n_float += float_keys
        n_perm += not float_keys
        # neutral on-disk format: what is loaded is what was saved, operator by operator
        with tempfile.TemporaryDirectory() as d:
            kio.save_keynet(ka, os.path.join(d, 'k.npz'))
            kc = kio.load_keynet(os.path.join(d, 'k.npz'))
        lc = [(n, c) for (n, c) in kc._keynet.named_children() if isinstance(c, KeyedLayer)]
        assert [n for (n, _) in lc] == [n for (n, _) in la]
        for ((name, a), (_, c)) in zip(la, lc):
            (ca, cc) = (_csr(a.W), _csr(c.W))
            assert np.array_equal(ca.indptr, cc.indptr) and np.array_equal(ca.indices, cc.indices) and np.array_equal(ca.data, cc.data), (case, name)
    assert n_perm >= 20 and n_float >= 10 and n_conv_factored >= 30, (n_perm, n_float, n_conv_factored)
