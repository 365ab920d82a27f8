"""Neutral on-disk format (keynet_amd.io): save_keynet -> load_keynet reproduces every operator kind array for
array and keeps each layer's arithmetic contract; on the GPU the reloaded key-net computes bit-identical logits.  The importer
for reference pickles (tests/golden/import_pickle.py) is exercised where the reference is mounted (build container only)."""
import os
import subprocess
import sys
import numpy as np
import pytest
import scipy.sparse
import torch
from torch import nn

from keynet_amd import io as kio
from keynet_amd import sparse as ksp
from keynet_amd import system as ksys
from keynet_amd.layer import KeyedLayer
from nets import MiniNet, load_weights

HERE = os.path.dirname(os.path.abspath(__file__))


def _arrays(W):
    out = {}
    kio.operator_to_arrays(W, '', out)
    return out


def _same_operator(A, B):
    (a, b) = (_arrays(A), _arrays(B))
    assert sorted(a.keys()) == sorted(b.keys())
    for k in a:
        assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k
    assert type(A) is type(B) or (isinstance(A, ksp.DiagonalTiledMatrix) and type(B) is ksp.TiledMatrix)
    assert tuple(A.shape) == tuple(B.shape) and A.nnz() == B.nnz()


def _mini(golden, direct, exact=None, factory=ksys.TiledPermutationKeynet):
    z = golden('mini_tiled_permutation.npz')
    net = load_weights(MiniNet(), z)
    np.random.seed(0)
    (sensor, knet) = factory((2, 16, 16), net, 4, direct=direct, exact=exact)
    return (z, sensor, knet)


@pytest.mark.parametrize('direct,exact', [(False, None), (True, None), (True, True), (False, False)])
def test_save_load_roundtrip_host(golden, tmp_path, direct, exact):
    """conv2dtiled (reference route) / convtaps (direct keying), tiled pools, csr Linear, ReLU markers, sensor keys, exact flags."""
    (z, sensor, knet) = _mini(golden, direct, exact)
    f = str(tmp_path / 'knet.npz')
    kio.save_keynet(knet, f, sensor=sensor)
    (sensor2, knet2) = kio.load_keynet(f, with_sensor=True)
    names = [n for (n, _) in knet._keynet.named_children()]
    assert [n for (n, _) in knet2._keynet.named_children()] == names and tuple(knet2._outshape) == tuple(knet._outshape)
    kinds = set()
    for ((n, a), (_, b)) in zip(knet._keynet.named_children(), knet2._keynet.named_children()):
        if isinstance(a, KeyedLayer):
            _same_operator(a.W, b.W)
            assert b._layertype == a._layertype and b._exact == a._exact, n
            kinds.add(str(_arrays(a.W)['kind']))
        else:
            assert isinstance(a, nn.ReLU) and isinstance(b, nn.ReLU)
    assert kinds == ({'convtaps', 'tiled', 'csr'} if direct else {'conv2dtiled', 'tiled', 'csr'})
    for (A, B) in zip(sensor.keypair(), sensor2.keypair()):
        (A, B) = (A.tocsr(), B.tocsr())
        assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices) and np.array_equal(A.data, B.data)
    assert sensor2._inshape == sensor._inshape
    assert kio.load_keynet(f).num_parameters() == knet.num_parameters()


def test_diagonal_tiled_and_dense_operators_roundtrip(tmp_path):
    """The remaining container kinds: a DiagonalTiledMatrix layer (saved as its tile list) and a dense ndarray operator."""
    rng = np.random.RandomState(0)
    D = ksp.DiagonalTiledMatrix(rng.rand(3, 3).astype(np.float32), shape=(10, 10))
    E = ksp.SparseMatrix(rng.rand(4, 10).astype(np.float32))
    knet = ksys.KeyedModel.fromlayers({'diag': KeyedLayer.fromoperator(D, 'diag'), 'relu1': nn.ReLU(), 'dense': KeyedLayer.fromoperator(E, 'dense', exact=True)}, (3, 1, 1))
    f = str(tmp_path / 'k.npz')
    kio.save_keynet(knet, f)
    k2 = kio.load_keynet(f)
    assert np.array_equal(np.asarray(k2.diag.W.tocsr().todense()), np.asarray(D.tocsr().todense()))
    assert np.array_equal(np.asarray(k2.dense.W._matrix.todense()), E._matrix)
    assert k2.dense._exact is True and isinstance(k2.relu1, nn.ReLU)


@pytest.mark.gpu
@pytest.mark.parametrize('direct,exact', [(False, None), (True, None), (True, True)])
def test_reloaded_keynet_computes_identical_logits(golden, tmp_path, direct, exact):
    """GPU: logits of the reloaded key-net == logits of the original, bit for bit (tolerance-mode nets stay on the MFMA path,
    exact-mode nets stay bit-exact with the reference vectors after the reload)."""
    assert torch.cuda.is_available()
    (z, sensor, knet) = _mini(golden, direct, exact)
    f = str(tmp_path / 'knet.npz')
    kio.save_keynet(knet, f, sensor=sensor)
    (sensor2, knet2) = kio.load_keynet(f, with_sensor=True)
    x = torch.as_tensor(z['x_plain']).to('cuda:0')
    xc = sensor.fromtensor(x).encrypt().astensor()
    xc2 = sensor2.fromtensor(x).encrypt().astensor()
    assert torch.equal(xc, xc2) and np.array_equal(xc.cpu().numpy(), z['x_cipher'])
    (y, y2) = (knet.forward_linear(xc), knet2.forward_linear(xc2))
    assert torch.equal(y, y2)
    if exact:
        assert np.array_equal(y2.cpu().numpy()[:, :-1], z['logits_keyed'])        # the reference's own logits, bit for bit
    else:
        assert np.abs(y2.cpu().numpy()[:, :-1] - z['logits_keyed']).max() <= 1e-5


def test_float64_operators_keep_their_dtype(golden, tmp_path):
    """The public challenge key-net (float64 conv / pool operators, float32 fc): loaded, saved and re-loaded, every operator keeps the dtype
    the reference computes it in; integer matrices count as float64 (numpy's up-cast against float32 activations)."""
    z = golden('challenge_kat.npz')
    knet = kio.keynet_from_arrays(z)
    want = {n: z['L.%s.data' % n].dtype for n in [str(v) for v in z['layer_names']] if ('L.%s.data' % n) in z.files}
    assert sorted(str(d) for d in set(want.values())) == ['float32', 'float64']
    f = kio.save_keynet(knet, str(tmp_path / 'challenge_again.npz'))
    k2 = kio.load_keynet(f)
    for (name, c) in k2._keynet.named_children():
        if isinstance(c, KeyedLayer):
            assert c.W._matrix.dtype == want[name] and c.W.is_float64() == (want[name] == np.float64), name
            assert np.array_equal(c.W._matrix.data, z['L.%s.data' % name]), name
            (ip, ix, dt) = ksp._stored_order_csr(c.W._matrix)
            assert dt.dtype == want[name]
    assert ksp.SparseMatrix(scipy.sparse.eye(3, dtype=np.int64).tocsr()).is_float64()
    assert not ksp.SparseMatrix(scipy.sparse.eye(3, dtype=np.float32).tocsr()).is_float64()
    assert not ksp.SparseMatrix(np.eye(3)).is_float64()            # dense ndarray operators: BLAS in the reference, float32 here


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='the reference is only mounted in the build container')
def test_import_reference_pickle(golden, tmp_path):
    """tests/golden/import_pickle.py on the pickle the reference ships (demo/keynet_challenge_lenet_10AUG20.pkl): the converted
    archive holds the same operators as the committed challenge fixture and loads without the reference."""
    out = str(tmp_path / 'challenge.npz')
    p = subprocess.run([sys.executable, os.path.join(HERE, 'golden', 'import_pickle.py'), '/root/reference/demo/keynet_challenge_lenet_10AUG20.pkl', out],
                       capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    (z, ref) = (np.load(out, allow_pickle=False), golden('challenge_kat.npz'))
    assert [str(n) for n in z['layer_names']] == [str(n) for n in ref['layer_names']]
    for n in [str(n) for n in z['layer_names']]:
        for k in ('kind', 'indptr', 'indices', 'data', 'shape'):
            key = 'L.%s.%s' % (n, k)
            if key in ref.files:
                assert np.array_equal(z[key], ref[key]) and z[key].dtype == ref[key].dtype, key     # (float64 conv / pool values stay float64)
    knet = kio.load_keynet(out)
    assert knet.num_parameters() > 0 and len(list(knet._keynet.children())) == len(z['layer_names'])
