#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz by running the REFERENCE (visym/keynet @ /root/reference) in this container.

    cd /tmp && python /root/repo/tests/golden/make_golden.py [f1 f2 f3 f4 f5]

Only data leaves this script: inputs, operator triplets, expected outputs.  No reference source is copied.
The reference is imported through tests/golden/_refimport.py (stand-ins for the absent third-party packages
numba / vipy / torchvision).  The GPU box has no /root/reference; tests read the committed .npz files only.

Fixtures:
  F1 lenet_perm.npz      PermutationKeynet(LeNet_AvgPool) seed 0: weights, keyed CSR triplets in STORED order,
                         sensor key, x[8], per-layer outputs of knet._keynet, owl.jpg 28x28 case
  F2 challenge_kat.npz   demo/keynet_challenge_lenet_10AUG20.{pkl,png}: CSR triplets + pixels + published floats
  F3 mini_tiled_*.npz    MiniNet (2,16,16) Tiled{Identity,Permutation,Orthogonal}Keynet: blocks/tiles dumps, CSR, I/O
  F4 tiled_cases.npz     TiledMatrix / Conv2dTiledMatrix / DiagonalTiledMatrix cases mirroring test/test_sparse.py:122-199
  F5 allconv_tiny_perm.npz  reduced-channel AllConvNet-shaped PermutationKeynet (stride 2, 1x1 conv, dropout bypass)
  F6 keygen_cases.npz    (A, Ainv) of keynet.system.keygen for every key family on small shapes (host-keying parity)
  F7 bn_tiny_{perm,identity}.npz  conv -> '<conv>_bn' BatchNorm2d (random running stats) -> dropout -> relu nets: the batch-norm fold
                         (keynet/torch.py:99-113) feeds the stored operators, so its f32 association is part of the contract
  F8 mini_tiled_stochastic.npz  MiniNet (2,16,16) under the key family of test/test_keynet.py:116-129 (test_vgg16_stochastic:
                         hierarchical permutation + doubly-stochastic local keys + uniform random affine): FILLED-IN operators
                         (keynet/sparse.py:335-353 keys, SpGEMM fill-in at keynet/layer.py:35), blocks/tiles dumps, CSR, I/O
"""
import os
import sys
import json
import hashlib
import numpy as np
import scipy.sparse
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

keynet = _refimport.import_reference()


def csr_triplet(M):
    """Stored-order CSR triplet of a scipy matrix WITHOUT canonicalising it (order is part of the contract)."""
    assert scipy.sparse.issparse(M)
    if M.format != 'csr':
        M = M.tocsr()
    return (np.asarray(M.indptr, dtype=np.int32), np.asarray(M.indices, dtype=np.int32), np.asarray(M.data))


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def dump_operator(prefix, W, out):
    """Dump the object stored in KeyedLayer.W (keynet/layer.py:81-82) under keys prefixed `prefix`."""
    S = keynet.sparse
    if isinstance(W, S.Conv2dTiledMatrix):
        out[prefix + 'kind'] = np.array('conv2dtiled')
        out[prefix + 'shape'] = np.array(W.shape, dtype=np.int64)
        out[prefix + 'inshape'] = np.array(W._inshape, dtype=np.int64)
        out[prefix + 'outshape'] = np.array(W._outshape, dtype=np.int64)
        out[prefix + 'tileshape'] = np.array(W._tileshape, dtype=np.int64)
        out[prefix + 'blocks'] = np.array(W._blocks, dtype=np.int64).reshape(-1, 3)
        keys = list(W._tiles.keys())  # insertion order == dict iteration order used by _tosparse (sparse.py:804)
        chan = [(k, v) for (k, v) in zip(keys, W._tiles.values()) if v.shape != (1, 1) or (W._inshape[0] == 1 and W._outshape[0] == 1 and False)]
        # split by shape: channel matrices are (Cout,Cin); bias tiles are (1,1) (sparse.py:772)
        (Cout, Cin) = (W._outshape[0], W._inshape[0])
        is_bias = np.array([not (v.shape == (Cout, Cin)) for v in W._tiles.values()], dtype=bool)
        if Cout == 1 and Cin == 1:
            # ambiguous by shape: bias tiles are those whose k >= k_offset, appended last (sparse.py:769-772)
            nb = 0
            ks = [k[2] for k in keys]
            # bias tile ids only appear in blocks whose column offset == Cin*Hin*Win
            biask = set(int(b[2]) for b in W._blocks if b[1] == Cin * W._inshape[1] * W._inshape[2])
            spatialk = set(int(b[2]) for b in W._blocks if b[1] != Cin * W._inshape[1] * W._inshape[2])
            assert not (biask & spatialk)
            is_bias = np.array([k[2] in biask for k in keys], dtype=bool)
        out[prefix + 'tile_keys'] = np.array(keys, dtype=np.int64).reshape(-1, 3)
        out[prefix + 'tile_isbias'] = is_bias
        cm = [np.asarray(v, dtype=np.float32) for (v, b) in zip(W._tiles.values(), is_bias) if not b]
        bm = [np.asarray(v, dtype=np.float32).reshape(()) for (v, b) in zip(W._tiles.values(), is_bias) if b]
        out[prefix + 'tile_chan'] = np.stack(cm) if len(cm) else np.zeros((0, Cout, Cin), np.float32)
        out[prefix + 'tile_bias'] = np.array(bm, dtype=np.float32)
        (ip, ix, dt) = csr_triplet(W.tocsr())
    elif isinstance(W, S.TiledMatrix):
        out[prefix + 'kind'] = np.array('diagtiled' if isinstance(W, S.DiagonalTiledMatrix) else 'tiled')
        out[prefix + 'shape'] = np.array(W.shape, dtype=np.int64)
        out[prefix + 'tileshape'] = np.array(W._tileshape, dtype=np.int64)
        out[prefix + 'blocks'] = np.array(list(W), dtype=np.int64).reshape(-1, 3)   # __iter__ (sparse.py:576-578 / 683-687)
        tiles = [t.tocoo() for t in W._tiles]
        out[prefix + 'tile_shapes'] = np.array([t.shape for t in tiles], dtype=np.int64).reshape(-1, 2)
        out[prefix + 'tile_ptr'] = np.cumsum([0] + [t.nnz for t in tiles]).astype(np.int64)
        out[prefix + 'tile_row'] = np.concatenate([t.row for t in tiles]).astype(np.int32)
        out[prefix + 'tile_col'] = np.concatenate([t.col for t in tiles]).astype(np.int32)
        out[prefix + 'tile_val'] = np.concatenate([t.data for t in tiles]).astype(np.float32)
        (ip, ix, dt) = csr_triplet(W.tocsr())
    else:
        assert isinstance(W, S.SparseMatrix)
        out[prefix + 'kind'] = np.array('csr')
        out[prefix + 'shape'] = np.array(W.shape, dtype=np.int64)
        (ip, ix, dt) = csr_triplet(W._matrix)
    out[prefix + 'indptr'] = ip
    out[prefix + 'indices'] = ix
    out[prefix + 'data'] = dt
    out[prefix + 'nnz'] = np.array(W.nnz(), dtype=np.int64)
    return sha(ip, ix, dt)


def dump_keynet(sensor, knet, net, x, out, manifest):
    """Layer operators + batched per-layer outputs of knet._keynet (system.py:115,132)."""
    names = []
    for (name, child) in knet._keynet.named_children():
        names.append(name)
        if isinstance(child, keynet.layer.KeyedLayer):
            manifest['layers'][name] = {'sha256': dump_operator('L.%s.' % name, child.W, out), 'shape': list(child.W.shape), 'nnz': int(child.nnz()),
                                        'layertype': child._layertype}
            out['L.%s.layertype' % name] = np.array(child._layertype)
        else:
            assert isinstance(child, nn.ReLU)
            out['L.%s.kind' % name] = np.array('relu')
    out['layer_names'] = np.array(names)
    # sensor keys (system.py:163-167); encrypt key is what W.torchdot applies
    (ip, ix, dt) = csr_triplet(sensor._encryptkey)
    (out['sensor.enc.indptr'], out['sensor.enc.indices'], out['sensor.enc.data']) = (ip, ix, dt)
    (ip, ix, dt) = csr_triplet(sensor._decryptkey)
    (out['sensor.dec.indptr'], out['sensor.dec.indices'], out['sensor.dec.data']) = (ip, ix, dt)
    out['sensor.shape'] = np.array(sensor._encryptkey.shape, dtype=np.int64)

    out['x_plain'] = x.numpy()
    x_lin = keynet.torch.affine_to_linear(x)                       # torch.py:65-68
    x_cipher = sensor.fromtensor(x).encrypt().astensor()          # system.py:209-218,250-255  [N, D+1]
    out['x_linear'] = x_lin.numpy()
    out['x_cipher'] = np.ascontiguousarray(x_cipher.numpy())
    y = x_cipher
    for (name, child) in knet._keynet.named_children():
        y = child.forward(y)
        out['Y.%s' % name] = np.ascontiguousarray(y.detach().numpy())
    out['logits_keyed'] = np.ascontiguousarray(y.detach().numpy()[:, :-1])
    with torch.no_grad():
        out['logits_plain'] = net(x).numpy()
    # N=1 API path (system.py:130-133) for the first image
    y1 = knet.forward(sensor.fromtensor(x[0:1]).encrypt().astensor())
    out['forward_n1'] = y1.detach().numpy()


def state_arrays(net, out):
    for (k, v) in net.state_dict().items():
        out['net.%s' % k] = v.numpy()


def save(name, out, manifest):
    f = os.path.join(HERE, name)
    np.savez_compressed(f, **out)
    manifest['file'] = name
    manifest['bytes'] = os.path.getsize(f)
    print('[make_golden]: %s  %d bytes' % (f, manifest['bytes']))
    return manifest


# ----------------------------------------------------------------------------------------------------------------
def f1():
    torch.manual_seed(0)
    net = keynet.mnist.LeNet_AvgPool().eval()
    np.random.seed(0)
    (sensor, knet) = keynet.system.PermutationKeynet((1, 28, 28), net)
    out = {}
    manifest = {'layers': {}, 'recipe': 'torch.manual_seed(0); LeNet_AvgPool(); np.random.seed(0); PermutationKeynet((1,28,28), net); torch.manual_seed(1); x=randn(8,1,28,28)'}
    state_arrays(net, out)
    torch.manual_seed(1)
    x = torch.randn(8, 1, 28, 28)
    dump_keynet(sensor, knet, net, x, out, manifest)
    # config 1: owl.jpg -> 28x28 grey (PIL; vipy's resize is parity-unpinned, pinned from the 28x28 tensor onward)
    from PIL import Image
    im = Image.open(os.path.join(_refimport.REFERENCE, 'demo', 'owl.jpg')).convert('L').resize((28, 28), Image.BILINEAR)
    x_owl = torch.as_tensor(np.asarray(im, dtype=np.float32)).reshape(1, 1, 28, 28)
    out['owl_plain'] = x_owl.numpy()
    xc = sensor.fromtensor(x_owl).encrypt().astensor()
    out['owl_cipher'] = np.ascontiguousarray(xc.numpy())
    out['owl_forward'] = knet.forward(xc).detach().numpy()
    with torch.no_grad():
        out['owl_plain_logits'] = net(x_owl).numpy()
    assert np.allclose(out['owl_forward'].flatten(), out['owl_plain_logits'].flatten(), atol=1e-2, rtol=1e-4)
    assert np.allclose(out['logits_keyed'], out['logits_plain'], atol=1e-5)
    # the known answer of demo/lenet.ipynb cell 2 (shape/nnz per layer)
    expect = [((4705, 785), 45049), ((1177, 4705), 10087), ((3137, 1177), 156737), ((785, 3137), 6401), ((121, 785), 94201), ((85, 121), 10165), ((11, 85), 851)]
    got = [(tuple(v['shape']), v['nnz']) for v in manifest['layers'].values()]
    assert got == expect, got
    return save('lenet_perm.npz', out, manifest)


def f2():
    """demo/challenge.ipynb cell 5 known answer."""
    import pickle
    from PIL import Image
    with open(os.path.join(_refimport.REFERENCE, 'demo', 'keynet_challenge_lenet_10AUG20.pkl'), 'rb') as f:
        obj = pickle.load(f)
    knet = obj[1] if isinstance(obj, (tuple, list)) else obj
    out = {}
    manifest = {'layers': {}, 'recipe': 'pickle.load(demo/keynet_challenge_lenet_10AUG20.pkl); png red channel/255; affine_to_linear; knet._keynet.forward'}
    names = []
    for (name, child) in knet._keynet.named_children():
        names.append(name)
        if isinstance(child, keynet.layer.KeyedLayer):
            manifest['layers'][name] = {'sha256': dump_operator('L.%s.' % name, child.W, out), 'shape': list(child.W.shape), 'nnz': int(child.nnz())}
            out['L.%s.layertype' % name] = np.array(child._layertype)
        else:
            out['L.%s.kind' % name] = np.array('relu')
    out['layer_names'] = np.array(names)
    im = np.asarray(Image.open(os.path.join(_refimport.REFERENCE, 'demo', 'keynet_challenge_lenet_10AUG20.png')))
    red = im[:, :, 0] if im.ndim == 3 else im
    out['png_red_u8'] = red.astype(np.uint8)
    x = torch.as_tensor(red.astype(np.float32) / 255.0).reshape(1, 1, *red.shape)
    x_lin = keynet.torch.affine_to_linear(x)
    out['x_linear'] = x_lin.numpy()
    y = x_lin
    for (name, child) in knet._keynet.named_children():
        y = child.forward(y)
        out['Y.%s' % name] = np.ascontiguousarray(y.detach().numpy())
    published = np.array([-0.0592, -0.0604, 0.0438, -0.0802, 0.0204, 0.0233, -0.0330, 0.0081, 0.0433, -0.0841], dtype=np.float64)
    out['published'] = published
    got = y.detach().numpy().flatten()[:-1]
    print('challenge got     ', np.round(got, 4))
    assert np.allclose(np.round(got, 4), published, atol=5.1e-5), (got, published)
    return save('challenge_kat.npz', out, manifest)


class MiniNet(nn.Module):
    """Generator-owned test net (2,16,16): conv-relu-pool-conv-relu-pool-fc, layer naming as keynet requires."""
    def __init__(self):
        super(MiniNet, self).__init__()
        self.conv1 = nn.Conv2d(2, 4, 3, stride=1, padding=1)
        self.relu1 = nn.ReLU()
        self.pool1 = nn.AvgPool2d(3, stride=2, padding=1)
        self.conv2 = nn.Conv2d(4, 4, 3, stride=1, padding=1)
        self.relu2 = nn.ReLU()
        self.pool2 = nn.AvgPool2d(3, stride=2, padding=1)
        self.fc1 = nn.Linear(4 * 4 * 4, 10)

    def forward(self, x):
        x = self.pool1(self.relu1(self.conv1(x)))
        x = self.pool2(self.relu2(self.conv2(x)))
        return self.fc1(x.view(-1, 4 * 4 * 4))


def f3():
    ms = []
    for (tag, factory, tilesize) in [('identity', lambda s, n, t: keynet.system.TiledIdentityKeynet(s, n, t), 4),
                                     ('permutation', lambda s, n, t: keynet.system.TiledPermutationKeynet(s, n, t), 4),
                                     ('permutation8', lambda s, n, t: keynet.system.TiledPermutationKeynet(s, n, t), 8),
                                     ('orthogonal', lambda s, n, t: keynet.system.TiledOrthogonalKeynet(s, n, t), 4)]:
        torch.manual_seed(0)
        net = MiniNet().eval()
        np.random.seed(0)
        (sensor, knet) = factory((2, 16, 16), net, tilesize)
        out = {}
        manifest = {'layers': {}, 'recipe': 'torch.manual_seed(0); MiniNet(); np.random.seed(0); Tiled%sKeynet((2,16,16), net, %d); torch.manual_seed(1); x=randn(4,2,16,16)' % (tag, tilesize)}
        state_arrays(net, out)
        torch.manual_seed(1)
        x = torch.randn(4, 2, 16, 16)
        dump_keynet(sensor, knet, net, x, out, manifest)
        err = np.abs(out['logits_keyed'] - out['logits_plain']).max()
        print('mini %s: |keyed-plain|=%g' % (tag, err))
        assert err < 1e-4
        ms.append(save('mini_tiled_%s.npz' % tag, out, manifest))
    return ms


def f4():
    """Cases of test/test_sparse.py:122-199 (same shapes, seeded)."""
    S = keynet.sparse
    conv = S.sparse_toeplitz_conv2d
    np.random.seed(42)
    out = {}
    manifest = {'layers': {}, 'cases': []}

    def case(name, W, T, x=None):
        """W: source scipy matrix, T: tiled object, x: [W.shape[1], n] dense."""
        out['C.%s.src.' % name + 'row'] = W.tocoo().row.astype(np.int32)
        out['C.%s.src.' % name + 'col'] = W.tocoo().col.astype(np.int32)
        out['C.%s.src.' % name + 'val'] = W.tocoo().data.astype(np.float32)
        manifest['layers'][name] = {'sha256': dump_operator('C.%s.' % name, T, out)}
        assert np.allclose(np.asarray(W.todense()).astype(np.float32), np.asarray(T.tocoo().todense()), atol=1e-5)
        if x is None:
            x = np.random.randn(W.shape[1], 3).astype(np.float32)
        out['C.%s.x' % name] = x
        out['C.%s.y' % name] = T.torchdot(torch.as_tensor(x)).numpy()
        manifest['cases'].append(name)

    (U, V) = (32, 32)
    W = conv((2, U, V), 0 * np.random.rand(4, 2, 3, 3), bias=1000 * np.ones(4).astype(np.float32), stride=1).astype(np.float32)
    case('zero_filter_bias', W, S.Conv2dTiledMatrix(W, inshape=(2, U, V), outshape=(4, U, V), tileshape=(4, 4), bias=True))

    W = scipy.sparse.coo_matrix(np.random.rand(474, 78).astype(np.float32))
    case('dense_ragged_14', W, S.TiledMatrix(W, tileshape=(14, 14)))

    W = conv((1, 8, 8), np.random.rand(1, 1, 3, 3))
    case('toeplitz8_t4', W, S.TiledMatrix(W, tileshape=(4, 4)))

    W = conv((1, 27, 26), np.random.rand(1, 1, 3, 3))
    case('conv_27x26_t3', W, S.Conv2dTiledMatrix(W, inshape=(1, 27, 26), outshape=(1, 27, 26), tileshape=(3, 3), bias=False))

    W = conv((1, 8, 8), np.random.rand(1, 1, 3, 3))
    case('conv_8x8_t4', W, S.Conv2dTiledMatrix(W, (1, 8, 8), (1, 8, 8), tileshape=(4, 4), bias=False))

    (U, V) = (17, 32)
    W = conv((1, U, V), np.random.rand(1, 1, 3, 3), bias=None)
    case('conv_17x32_t4', W, S.Conv2dTiledMatrix(W, inshape=(1, U, V), outshape=(1, U, V), tileshape=(4, 4), bias=False))

    Wb = conv((1, U, V), np.random.rand(1, 1, 3, 3), bias=np.random.rand(1))
    case('conv_17x32_t16_bias', Wb, S.Conv2dTiledMatrix(Wb, inshape=(1, U, V), outshape=(1, U, V), tileshape=(16, 16), bias=True))
    case('conv_17x32_t68_bias', Wb, S.Conv2dTiledMatrix(Wb, inshape=(1, U, V), outshape=(1, U, V), tileshape=(U * 4, U * 4), bias=True))

    B = np.random.rand(3, 3).astype(np.float32)
    T1 = S.DiagonalTiledMatrix(B, shape=(10, 10))
    case('diag_3_in_10', T1.tocoo(), T1, x=np.random.rand(10, 1).astype(np.float32))

    (U, V) = (32, 32)
    W = conv((2, U, V), np.random.rand(4, 2, 3, 3), bias=None, stride=2)
    case('conv_s2_t2x4', W, S.Conv2dTiledMatrix(W, inshape=(2, U, V), outshape=(4, U // 2, V // 2), tileshape=(2, 4), bias=False))
    W = conv((2, U, V), np.random.rand(4, 2, 3, 3), bias=np.random.rand(4), stride=2)
    case('conv_s2_t2x4_bias', W, S.Conv2dTiledMatrix(W, inshape=(2, U, V), outshape=(4, U // 2, V // 2), tileshape=(2, 4), bias=True))

    # SparseMatrix.dot/torchdot on dense + coo (test_sparse.py:304-329)
    Wd = np.random.rand(3, 3).astype(np.float32)
    xd = np.random.rand(3, 1).astype(np.float32)
    out['D.W'] = Wd
    out['D.x'] = xd
    out['D.y_dense'] = np.asarray(S.SparseMatrix(Wd).torchdot(torch.as_tensor(xd)).numpy())
    out['D.y_coo'] = np.asarray(S.SparseMatrix(scipy.sparse.coo_matrix(Wd)).torchdot(torch.as_tensor(xd)).numpy())
    return save('tiled_cases.npz', out, manifest)


class TinyAllConv(nn.Module):
    """Generator-owned reduced-channel net with the layer sequence/naming of AllConvNet (keynet/cifar10.py:14-45)."""
    def __init__(self):
        super(TinyAllConv, self).__init__()
        self.dropout0 = nn.Dropout(p=0.2)
        self.conv1 = nn.Conv2d(3, 6, 3, padding=1)
        self.relu1 = nn.ReLU()
        self.conv2 = nn.Conv2d(6, 6, 3, padding=1)
        self.relu2 = nn.ReLU()
        self.conv3 = nn.Conv2d(6, 6, 3, padding=1, stride=2)
        self.dropout3 = nn.Dropout(p=0.5)
        self.relu3 = nn.ReLU()
        self.conv4 = nn.Conv2d(6, 8, 3, padding=1)
        self.relu4 = nn.ReLU()
        self.conv6 = nn.Conv2d(8, 8, 3, padding=1, stride=2)
        self.dropout6 = nn.Dropout(p=0.5)
        self.relu6 = nn.ReLU()
        self.conv8 = nn.Conv2d(8, 8, 1)
        self.relu8 = nn.ReLU()
        self.conv9 = nn.Conv2d(8, 4, 1)
        self.relu9 = nn.ReLU()
        self.fc1 = nn.Linear(4 * 4 * 4, 12)
        self.relu10 = nn.ReLU()
        self.fc2 = nn.Linear(12, 10)

    def forward(self, x):
        x = self.relu1(self.conv1(self.dropout0(x)))
        x = self.relu2(self.conv2(x))
        x = self.relu3(self.dropout3(self.conv3(x)))
        x = self.relu4(self.conv4(x))
        x = self.relu6(self.dropout6(self.conv6(x)))
        x = self.relu8(self.conv8(x))
        x = self.relu9(self.conv9(x))
        x = self.relu10(self.fc1(x.view(-1, 4 * 4 * 4)))
        return self.fc2(x)


def f5():
    torch.manual_seed(0)
    net = TinyAllConv().eval()
    np.random.seed(0)
    (sensor, knet) = keynet.system.PermutationKeynet((3, 16, 16), net)
    out = {}
    manifest = {'layers': {}, 'recipe': 'torch.manual_seed(0); TinyAllConv(); np.random.seed(0); PermutationKeynet((3,16,16), net); torch.manual_seed(1); x=randn(5,3,16,16)'}
    state_arrays(net, out)
    torch.manual_seed(1)
    x = torch.randn(5, 3, 16, 16)
    dump_keynet(sensor, knet, net, x, out, manifest)
    err = np.abs(out['logits_keyed'] - out['logits_plain']).max()
    print('tiny allconv: |keyed-plain|=%g' % err)
    assert err < 1e-5
    return save('allconv_tiny_perm.npz', out, manifest)


class TinyBN(nn.Module):
    """Generator-owned net with the batch-norm placement of AllConvNet(batchnorm=True) (keynet/cifar10.py:14-45):
    conv -> '<conv>_bn' -> dropout -> relu, twice (stride 2 and stride 1), then a Linear."""
    def __init__(self):
        super(TinyBN, self).__init__()
        self.conv1 = nn.Conv2d(3, 6, 3, padding=1)
        self.relu1 = nn.ReLU()
        self.conv3 = nn.Conv2d(6, 8, 3, padding=1, stride=2)
        self.conv3_bn = nn.BatchNorm2d(8)
        self.dropout3 = nn.Dropout(p=0.5)
        self.relu3 = nn.ReLU()
        self.conv4 = nn.Conv2d(8, 5, 3, padding=1)
        self.conv4_bn = nn.BatchNorm2d(5)
        self.relu4 = nn.ReLU()
        self.fc1 = nn.Linear(5 * 4 * 4, 7)

    def forward(self, x):
        x = self.relu1(self.conv1(x))
        x = self.relu3(self.dropout3(self.conv3_bn(self.conv3(x))))
        x = self.relu4(self.conv4_bn(self.conv4(x)))
        return self.fc1(x.view(-1, 5 * 4 * 4))


def randomize_bn(net, seed):
    """Non-trivial eval-mode statistics and affine parameters (a fresh BatchNorm2d folds to the identity)."""
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g))
            m.running_var.copy_(torch.rand(m.num_features, generator=g) * 2 + 0.25)
            m.weight.data.copy_(torch.randn(m.num_features, generator=g))
            m.bias.data.copy_(torch.randn(m.num_features, generator=g))


def f7():
    import warnings
    ms = []
    for (tag, factory) in [('perm', lambda s, n: keynet.system.PermutationKeynet(s, n)), ('identity', lambda s, n: keynet.system.IdentityKeynet(s, n))]:
        torch.manual_seed(0)
        net = TinyBN().eval()
        randomize_bn(net, 3)
        np.random.seed(0)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            (sensor, knet) = factory((3, 8, 8), net)
        out = {}
        manifest = {'layers': {}, 'recipe': 'torch.manual_seed(0); TinyBN(); randomize_bn(net, 3); np.random.seed(0); %s((3,8,8), net); torch.manual_seed(1); x=randn(5,3,8,8)' % tag}
        state_arrays(net, out)
        torch.manual_seed(1)
        x = torch.randn(5, 3, 8, 8)
        dump_keynet(sensor, knet, net, x, out, manifest)
        err = np.abs(out['logits_keyed'] - out['logits_plain']).max()
        print('tiny bn %s: |keyed-plain|=%g' % (tag, err))
        assert err < 1e-4
        ms.append(save('bn_tiny_%s.npz' % tag, out, manifest))
    return ms


from keygen_case_table import STOCHASTIC_KW  # noqa: E402  (pure data)


def f8():
    """The filled-in key family (test/test_keynet.py:116-129 scaled to MiniNet): the reference's own keyed operators and per-layer outputs."""
    import warnings
    torch.manual_seed(0)
    net = MiniNet().eval()
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        (sensor, knet) = keynet.system.Keynet((2, 16, 16), net, **STOCHASTIC_KW)
    out = {}
    manifest = {'layers': {}, 'recipe': 'torch.manual_seed(0); MiniNet(); np.random.seed(0); Keynet((2,16,16), net, %s); torch.manual_seed(1); x=randn(4,2,16,16)'
                % ', '.join('%s=%r' % kv for kv in STOCHASTIC_KW.items())}
    state_arrays(net, out)
    torch.manual_seed(1)
    x = torch.randn(4, 2, 16, 16)
    dump_keynet(sensor, knet, net, x, out, manifest)
    err = np.abs(out['logits_keyed'] - out['logits_plain']).max()
    print('mini stochastic: |keyed-plain|=%g' % err)
    assert np.allclose(out['logits_keyed'], out['logits_plain'], atol=1e-5)        # the reference's own criterion for this family (test_keynet.py:128)
    for (name, rec) in manifest['layers'].items():
        if out['L.%s.kind' % name] == 'conv2dtiled':
            # fill-in: stored entries per output row of channel 0 against the 9 taps of the unkeyed operator
            (ip, shape) = (out['L.%s.indptr' % name], out['L.%s.shape' % name])
            rec['max_row_nnz'] = int(np.diff(ip).max())
            print('   %s: nnz %d, longest row %d' % (name, rec['nnz'], rec['max_row_nnz']))
    return save('mini_tiled_stochastic.npz', out, manifest)


from keygen_case_table import KEYGEN_CASES  # noqa: E402  (pure data: names, shapes, keyword arguments)


def f6():
    """(A, Ainv) pairs of the reference's keygen (keynet/system.py:317-469) for every key family, np.random.seed(7) each."""
    import warnings
    out = {}
    manifest = {'layers': {}, 'cases': [c[0] for c in KEYGEN_CASES], 'recipe': 'np.random.seed(7); keynet.system.keygen(shape, **kwargs)'}
    for (name, shape, kw) in KEYGEN_CASES:
        np.random.seed(7)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            (A, Ainv) = keynet.system.keygen(shape, **kw)
        for (tag, M) in (('A', A), ('Ainv', Ainv)):
            M = M.tocsr()
            out['K.%s.%s.indptr' % (name, tag)] = np.asarray(M.indptr, dtype=np.int32)
            out['K.%s.%s.indices' % (name, tag)] = np.asarray(M.indices, dtype=np.int32)
            out['K.%s.%s.data' % (name, tag)] = np.asarray(M.data)
            out['K.%s.%s.shape' % (name, tag)] = np.array(M.shape, dtype=np.int64)
        err = np.abs((A.dot(Ainv) - scipy.sparse.eye(A.shape[0])).todense()).max()
        print('keygen %-18s nnz(A)=%6d  |A.Ainv - I|=%.2e  dtype=%s' % (name, A.nnz, err, A.dtype))
        manifest['layers'][name] = {'nnz': int(A.nnz), 'dtype': str(A.dtype)}
    return save('keygen_cases.npz', out, manifest)


if __name__ == '__main__':
    os.chdir('/tmp')
    which = sys.argv[1:] or ['f1', 'f2', 'f3', 'f4', 'f5', 'f6', 'f7', 'f8']
    mf = os.path.join(HERE, 'MANIFEST.json')
    manifest = json.load(open(mf)) if os.path.exists(mf) else {}
    manifest['_versions'] = {'numpy': np.__version__, 'scipy': scipy.__version__, 'torch': torch.__version__, 'python': sys.version.split()[0]}
    for w in which:
        r = {'f1': f1, 'f2': f2, 'f3': f3, 'f4': f4, 'f5': f5, 'f6': f6, 'f7': f7, 'f8': f8}[w]()
        manifest[w] = r
    with open(mf, 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
