"""Pins the CPU oracle (oracle/) against vectors produced by the reference itself (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import oracle

NETS = ['lenet_perm.npz', 'allconv_tiny_perm.npz', 'mini_tiled_identity.npz', 'mini_tiled_permutation.npz',
        'mini_tiled_permutation8.npz', 'mini_tiled_orthogonal.npz', 'mini_tiled_stochastic.npz', 'bn_tiny_perm.npz', 'bn_tiny_identity.npz']


@pytest.mark.parametrize('name', NETS)
@pytest.mark.parametrize('structured', [False, True])
def test_oracle_forward_bit_exact(golden, name, structured):
    """oracle forward (C csr_matvecs restatement; tiles re-expanded when structured) == reference outputs, bit for bit."""
    z = golden(name)
    layers = oracle.load_golden_layers(z, structured=structured)
    (y, outs) = oracle.keynet_forward(layers, z['x_cipher'], collect=True)
    for (lname, _, _) in layers:
        ref = z['Y.%s' % lname]
        assert outs[lname].dtype == ref.dtype == np.float32
        assert np.array_equal(outs[lname], ref), 'layer %s of %s differs from the reference' % (lname, name)
    assert np.array_equal(y[:, :-1], z['logits_keyed'])
    assert np.allclose(z['logits_keyed'], z['logits_plain'], atol=2e-5)


@pytest.mark.parametrize('name', NETS)
def test_tile_expansion_matches_reference_csr(golden, name):
    """tiled_to_csr / conv2dtiled_to_csr reproduce the reference's tocsr() triplets exactly (keynet/sparse.py:621-641, 781-835)."""
    z = golden(name)
    for lname in [str(n) for n in z['layer_names']]:
        p = 'L.%s.' % lname
        if str(z[p + 'kind']) in ('tiled', 'conv2dtiled', 'diagtiled'):
            (shape, ip, ix, dt) = oracle.operator_from_golden(z, p)
            assert np.array_equal(ip, z[p + 'indptr']) and np.array_equal(ix, z[p + 'indices']) and np.array_equal(dt, z[p + 'data'])


def test_sensor_encrypt_is_csr_matvecs(golden):
    """KeyedSensor.encrypt (keynet/system.py:250-255) = the same primitive with W = encrypt key."""
    for name in NETS:
        z = golden(name)
        shape = tuple(int(v) for v in z['sensor.shape'])
        op = (shape, z['sensor.enc.indptr'], z['sensor.enc.indices'], z['sensor.enc.data'])
        xl = oracle.affine_to_linear(z['x_plain'])
        assert np.array_equal(xl, z['x_linear'])
        assert np.array_equal(oracle.layer_forward(op, xl), z['x_cipher'])
        dec = (shape, z['sensor.dec.indptr'], z['sensor.dec.indices'], z['sensor.dec.data'])
        back = oracle.layer_forward(dec, z['x_cipher'])
        assert np.allclose(oracle.linear_to_affine(back), z['x_linear'][:, :-1], atol=1e-4)


def test_owl_config1(golden):
    """BASELINE.json configs[0]: PermutationKeynet LeNet_AvgPool forward on owl.jpg (28x28 grey), N=1, CPU."""
    z = golden('lenet_perm.npz')
    layers = oracle.load_golden_layers(z)
    y = oracle.keynet_forward(layers, z['owl_cipher'])
    out = oracle.linear_to_affine(y, (10, 1, 1))
    assert np.array_equal(out, z['owl_forward'])
    assert np.array_equal(oracle.linear_to_affine(oracle.keynet_forward(layers, z['x_cipher'][0:1]), (10, 1, 1)), z['forward_n1'])


def test_challenge_known_answer(golden):
    """demo/challenge.ipynb cell 5: the only literal known answer in the reference repo (float64 operators)."""
    z = golden('challenge_kat.npz')
    layers = oracle.load_golden_layers(z)
    (y, outs) = oracle.keynet_forward(layers, z['x_linear'], collect=True)
    for (lname, _, _) in layers:
        assert np.array_equal(outs[lname], z['Y.%s' % lname]), lname
    assert np.array_equal(np.round(y.flatten()[:-1].astype(np.float64), 4), z['published'])   # all 4 printed decimals


def test_tiled_cases(golden):
    """test/test_sparse.py:122-199 shapes: ragged tiles, stride 2, bias tiles, tile >= plane, diagonal repeat."""
    z = golden('tiled_cases.npz')
    names = sorted({k.split('.')[1] for k in z.files if k.startswith('C.')})
    assert len(names) == 11
    for n in names:
        p = 'C.%s.' % n
        op = oracle.operator_from_golden(z, p)
        assert np.array_equal(op[1], z[p + 'indptr']) and np.array_equal(op[2], z[p + 'indices']) and np.array_equal(op[3], z[p + 'data']), n
        y = oracle.csr_matvecs(op[0], op[1], op[2], op[3], z[p + 'x'])
        assert np.array_equal(y, z[p + 'y']), n
        # and against the dense source matrix (the reference's own criterion, atol 1e-5)
        W = np.zeros(op[0], dtype=np.float64)
        np.add.at(W, (z[p + 'src.row'], z[p + 'src.col']), z[p + 'src.val'])
        assert np.allclose(W.dot(z[p + 'x']), y, atol=1e-4, rtol=1e-5), n
    assert np.allclose(z['D.y_dense'], z['D.W'].dot(z['D.x']), atol=1e-5)
    assert np.allclose(z['D.y_coo'], z['D.W'].dot(z['D.x']), atol=1e-5)


def _tile_list(z, p):
    """tiles[k] = [nnz_k, 3] arrays of (ii, jj, v): the operand form of keynet.torch.TiledMatrix._torchdot (keynet/torch.py:173-184)."""
    (ptr, tr, tc, tv) = (z[p + 'tile_ptr'], z[p + 'tile_row'], z[p + 'tile_col'], z[p + 'tile_val'])
    return [np.stack([tr[ptr[k]:ptr[k + 1]], tc[ptr[k]:ptr[k + 1]], tv[ptr[k]:ptr[k + 1]]], axis=1).astype(np.float64) for k in range(len(ptr) - 1)]


def test_explicit_tile_loop_oracle(golden):
    """the serial reading of the reference's explicit tile loop equals the reference's TiledMatrix.torchdot results (golden y,
    produced through tocsr() + csr_matvecs) to its own 1e-5, on every tiled fixture; and the host half of the product's version
    (keynet_amd.torch.TiledMatrix._loop_csr: the loop's visiting order as an order-preserving CSR) reproduces the loop bit for bit through the oracle."""
    from keynet_amd.torch import TiledMatrix as TorchTiled
    z = golden('tiled_cases.npz')
    names = sorted({k.split('.')[1] for k in z.files if k.startswith('C.') and k.endswith('.tile_ptr')})
    assert len(names) == 3                                       # the TiledMatrix / DiagonalTiledMatrix fixtures (conv tiles are channel matrices: another operand form)
    for n in names:
        p = 'C.%s.' % n
        (shape, tileshape, blocks, tiles) = (tuple(int(v) for v in z[p + 'shape']), tuple(int(v) for v in z[p + 'tileshape']), z[p + 'blocks'], _tile_list(z, p))
        y = oracle.tiled_torchdot_loop(z[p + 'x'], tileshape, shape, tiles, blocks)
        assert np.allclose(y, z[p + 'y'], atol=1e-5, rtol=1e-5), n
        (ip, ix, dt) = TorchTiled._loop_csr(tileshape, shape, tiles, blocks)
        assert np.array_equal(oracle.csr_matvecs(shape, ip, ix, dt, z[p + 'x']), y), n


def test_oracle_matches_scipy_on_random_unsorted():
    """The C restatement == scipy's own csr_matvecs on a non-canonical matrix (the third-party engine the reference calls)."""
    import scipy.sparse
    rng = np.random.RandomState(0)
    (m, n, nnz, b) = (257, 123, 4000, 17)
    indptr = np.sort(np.concatenate(([0, nnz], rng.randint(0, nnz, m - 1)))).astype(np.int32)
    indices = rng.randint(0, n, nnz).astype(np.int32)     # unsorted, with duplicates
    data = rng.randn(nnz).astype(np.float32)
    X = rng.randn(n, b).astype(np.float32)
    W = scipy.sparse.csr_matrix((data, indices, indptr), shape=(m, n))
    assert np.array_equal(oracle.csr_matvecs((m, n), indptr, indices, data, X), W.dot(X))
    Xt = np.asfortranarray(X)
    assert np.array_equal(oracle.csr_matvecs((m, n), indptr, indices, data, Xt), W.dot(Xt))
