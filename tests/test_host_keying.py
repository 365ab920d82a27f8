"""Host-side keying (keynet_amd.system / .layer / .sparse) reproduces the reference's stored operators bit for bit
under the same numpy seed.  CPU only: device handles are created lazily and never touched here."""
import numpy as np
import pytest
import torch

import keynet_amd.system as ksys
import keynet_amd.sparse as ksp
from keynet_amd.layer import KeyedLayer
from nets import LeNet_AvgPool, MiniNet, TinyAllConv, TinyBN, load_weights


def _check_layers(z, knet):
    names = [str(n) for n in z['layer_names']]
    assert [n for (n, _) in knet._keynet.named_children()] == names
    for (name, child) in knet._keynet.named_children():
        p = 'L.%s.' % name
        kind = str(z[p + 'kind'])
        if kind == 'relu':
            assert isinstance(child, torch.nn.ReLU)
            continue
        assert isinstance(child, KeyedLayer) and child._layertype == str(z[p + 'layertype'])
        W = child.W
        assert tuple(W.shape) == tuple(int(v) for v in z[p + 'shape'])
        assert W.nnz() == int(z[p + 'nnz']), name
        if kind == 'csr':
            M = W._matrix
            assert M.format == 'csr'
            assert np.array_equal(M.indptr, z[p + 'indptr']) and np.array_equal(M.indices, z[p + 'indices']), 'stored order of %s differs' % name
            assert np.array_equal(M.data, z[p + 'data']), name
        elif kind == 'tiled':
            assert isinstance(W, ksp.TiledMatrix) and tuple(W.tileshape()) == tuple(int(v) for v in z[p + 'tileshape'])
            assert np.array_equal(np.array(W._blocks), z[p + 'blocks'])
            (ptr, tr, tc, tv) = W._tile_arrays()
            assert np.array_equal(ptr, z[p + 'tile_ptr']) and np.array_equal(tr, z[p + 'tile_row']) and np.array_equal(tc, z[p + 'tile_col']) and np.array_equal(tv, z[p + 'tile_val'])
        elif kind == 'conv2dtiled':
            assert isinstance(W, ksp.Conv2dTiledMatrix)
            (bl, tk, ib, ch, bs) = W._golden_arrays()
            assert np.array_equal(bl, z[p + 'blocks']) and np.array_equal(tk, z[p + 'tile_keys']) and np.array_equal(ib.astype(bool), z[p + 'tile_isbias'])
            assert np.array_equal(ch, z[p + 'tile_chan']) and np.array_equal(bs, z[p + 'tile_bias'])
            c = W.tocsr()
            assert np.array_equal(c.indptr, z[p + 'indptr']) and np.array_equal(c.indices, z[p + 'indices']) and np.array_equal(c.data, z[p + 'data'])


def _check_sensor(z, sensor):
    E = sensor._encryptkey.tocsr()
    assert np.array_equal(E.indptr, z['sensor.enc.indptr']) and np.array_equal(E.indices, z['sensor.enc.indices']) and np.array_equal(E.data, z['sensor.enc.data'])


def test_permutation_keynet_lenet_matches_reference(golden):
    z = golden('lenet_perm.npz')
    net = load_weights(LeNet_AvgPool(), z)
    np.random.seed(0)
    (sensor, knet) = ksys.PermutationKeynet((1, 28, 28), net)
    _check_sensor(z, sensor)
    _check_layers(z, knet)
    assert knet.num_parameters() == 323491     # demo/lenet.ipynb cell 2 total


def test_permutation_keynet_allconv_tiny_matches_reference(golden):
    """stride-2 convs, 1x1 convs, dropout layers that are bypassed but still draw keys."""
    z = golden('allconv_tiny_perm.npz')
    net = load_weights(TinyAllConv(), z)
    np.random.seed(0)
    (sensor, knet) = ksys.PermutationKeynet((3, 16, 16), net)
    _check_sensor(z, sensor)
    _check_layers(z, knet)


@pytest.mark.parametrize('tag,tilesize', [('identity', 4), ('permutation', 4), ('permutation8', 8)])
def test_tiled_keynets_match_reference(golden, tag, tilesize):
    z = golden('mini_tiled_%s.npz' % tag)
    net = load_weights(MiniNet(), z)
    np.random.seed(0)
    factory = ksys.TiledIdentityKeynet if tag == 'identity' else ksys.TiledPermutationKeynet
    (sensor, knet) = factory((2, 16, 16), net, tilesize)
    _check_sensor(z, sensor)
    _check_layers(z, knet)


@pytest.mark.parametrize('tag', ['perm', 'identity'])
def test_batchnorm_fold_matches_reference(golden, tag):
    """conv -> '<conv>_bn' -> (dropout) -> relu: the folded conv (keynet/torch.py:99-113 association) and the separately keyed
    ReLU reproduce the reference's stored operators bit for bit (random running statistics, so the fold is not the identity)."""
    import warnings
    z = golden('bn_tiny_%s.npz' % tag)
    net = load_weights(TinyBN(), z)
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        (sensor, knet) = (ksys.PermutationKeynet if tag == 'perm' else ksys.IdentityKeynet)((3, 8, 8), net)
    _check_sensor(z, sensor)
    _check_layers(z, knet)


def test_layergen_backend_seam():
    """keynet/system.py:311-314: unknown backends raise ValueError('invalid backend ...')."""
    with pytest.raises(ValueError, match='invalid backend'):
        ksys.layergen(torch.nn.ReLU(), (1, 2, 2), (1, 2, 2), None, None, backend='scipy')
    with pytest.raises(ValueError, match='invalid backend'):
        ksys.Keynet((1, 28, 28), None, backend='cupy')


def test_effective_tileshape_is_snapped():
    """'tile 64' on VGG-16 becomes 56/28/14/7 (keynet/util.py:16-28)."""
    from keynet_amd.util import find_closest_positive_divisor as f
    assert [f(h, 64) for h in (224, 112, 56, 28, 14, 7)] == [56, 56, 56, 28, 14, 7]
    assert f(28, 4) == 4 and f(14, 4) == 2 and f(7, 4) == 7 and f(27, 4) == 3


def test_toeplitz_matches_torch():
    """test/test_sparse.py:223-272: Toeplitz conv (stride 2, bias) == F.conv2d; avgpool == F.avg_pool2d(3,2,padding=1)."""
    import torch.nn.functional as F
    rng = np.random.RandomState(0)
    (N, C, U, V, M) = (2, 3, 8, 16, 4)
    img = rng.rand(N, C, U, V).astype(np.float32)
    f = rng.randn(M, C, 3, 3).astype(np.float32)
    b = rng.randn(M).astype(np.float32)
    for stride in (1, 2):
        T = ksp.sparse_toeplitz_conv2d((C, U, V), f, b, stride=stride)
        yh = T.dot(np.hstack((img.reshape(N, -1), np.ones((N, 1), np.float32))).T).T[:, :-1].reshape(N, M, U // stride, V // stride)
        y = F.conv2d(torch.tensor(img), torch.tensor(f), bias=torch.tensor(b), padding=1, stride=stride).numpy()
        assert np.allclose(y, yh, atol=1e-5)
    T = ksp.sparse_toeplitz_avgpool2d((C, U, V), (C, C, 3, 3), stride=2)
    yh = T.dot(np.hstack((img.reshape(N, -1), np.ones((N, 1), np.float32))).T).T[:, :-1].reshape(N, C, U // 2, V // 2)
    assert np.allclose(F.avg_pool2d(torch.tensor(img), 3, stride=2, padding=1).numpy(), yh, atol=1e-6)


def test_homogeneous_roundtrip_cpu():
    """test/test_sparse.py:37-50 on CPU tensors (device versions are covered by the gpu tests)."""
    from keynet_amd.torch import affine_to_linear, linear_to_affine, affine_to_linear_matrix
    x = torch.rand(2, 2, 3, 3)
    xl = affine_to_linear(x)
    assert xl.shape == (2, 19) and torch.all(xl[:, -1] == 1)
    assert torch.equal(linear_to_affine(xl, (2, 2, 3, 3)), x)
    with pytest.raises(ValueError):
        linear_to_affine(xl * 1.01)
    W = torch.rand(18, 18)
    b = torch.rand(18)
    Wh = affine_to_linear_matrix(W, b)
    assert np.allclose(linear_to_affine(torch.matmul(affine_to_linear(x), Wh)).numpy(), (torch.matmul(x.view(2, -1), W.t()) + b).numpy(), atol=1e-5)
