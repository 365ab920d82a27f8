"""Key-net assembly and the top of the hot path (mirror of keynet/system.py:26-516).

    (sensor, model) = PermutationKeynet(inshape, net)          # factories, keynet/system.py:472-510
    y = model.forward(sensor.fromtensor(x).encrypt().astensor())   # keynet/system.py:130-133, README.md:27-34

Keying (host, offline) follows the reference's algebra call for call so that, under the same numpy seed, the stored
operators are identical (tests/test_host_keying.py compares CSR triplets with the golden vectors).  The forward runs on
the MI355X: layers exchange feature-major [D+1, N] blocks, ReLU is fused into the producing kernel, and batches are
supported (the reference's KeyedModel.forward is N=1 only).
"""
import copy
import warnings
from collections import OrderedDict
import numpy as np
import scipy.sparse
import torch
from torch import nn

from . import torch as ktorch
from . import sparse as ksp
from . import layer as klayer
from . import _capi
from .globals import verbose
from .util import find_closest_positive_divisor
from .sparse import sparse_permutation_matrix, sparse_identity_matrix, sparse_affine_to_linear, DiagonalTiledMatrix
from . import keys as kkeys


class KeyedModel(object):
    def __init__(self, net, inshape, inkey, f_layername_to_keypair, f_module_to_keyedmodule=None, do_output_encryption=False):
        net.eval()
        shapes = ktorch.netshape(net, inshape)

        # Splice identity layers out of the prev/next chain.  As in the reference (keynet/system.py:33-40) the entries
        # themselves stay in the table (its filter is a substring test of the KEY inside 'dropout'), so every dropout
        # layer still draws one key below -- this matters for RNG reproducibility.
        for marker in ['dropout']:
            shapes = OrderedDict((k, v) for (k, v) in shapes.items() if k not in marker)
            for (k, v) in shapes.items():
                if v['nextlayer'] is not None and marker in v['nextlayer']:
                    v['nextlayer'] = shapes[v['nextlayer']]['nextlayer']
                elif v['prevlayer'] is not None and marker in v['prevlayer']:
                    v['prevlayer'] = shapes[v['prevlayer']]['prevlayer']

        last = shapes['output']['prevlayer']
        drawn = OrderedDict((k, {'pair': f_layername_to_keypair(k, v['outshape']), 'prev': v['prevlayer']})
                            for (k, v) in shapes.items() if k not in ('input', 'output'))
        keys = {k: {'A': d['pair'][0] if (k != last or do_output_encryption) else None,
                    'Ainv': inkey if d['prev'] == 'input' else drawn[d['prev']]['pair'][1]} for (k, d) in drawn.items()}
        keys['input'] = inkey
        keys['output'] = drawn[last]['pair'][1] if do_output_encryption else None

        layernames = set(k for (k, m) in net.named_children())
        keyed = OrderedDict()
        for (k, m) in net.named_children():
            if verbose():
                print('[keynet_amd.KeyedModel]: keying "%s"' % k)
            assert k in keys and k in shapes, 'no key / shape for layer "%s"' % k

            if isinstance(m, nn.BatchNorm2d):
                assert '_bn' in k, "Batchnorm layers must be named 'mylayername_bn' for corresponding linear layer mylayername"
                kp = k.split('_')[0]
                assert shapes[k]['prevlayer'] == kp, "Batchnorm layer named 'mylayer_bn' must come right after 'mylayer'"
                mp = copy.deepcopy(getattr(net, kp))
                (w, b) = ktorch.fuse_conv2d_and_bn(mp.weight, mp.bias, m.running_mean, m.running_var, 1E-5, m.weight, m.bias)
                (mp.weight, mp.bias) = (torch.nn.Parameter(w), torch.nn.Parameter(b))
                B = keys[k]['A'].dot(keys[k]['Ainv'])
                keyed[kp] = f_module_to_keyedmodule(mp, shapes[kp]['inshape'], shapes[k]['outshape'], B.dot(keys[kp]['A']), keys[kp]['Ainv'])

            elif isinstance(m, nn.ReLU):
                kp = shapes[k]['prevlayer']
                if '_bn' not in kp:
                    # the preceding linear layer is keyed with THIS ReLU's output key; the ReLU itself stays unkeyed
                    B = keys[k]['A'].dot(keys[k]['Ainv'])
                    keyed[kp] = f_module_to_keyedmodule(getattr(net, kp), shapes[kp]['inshape'], shapes[kp]['outshape'], B.dot(keys[kp]['A']), keys[kp]['Ainv'])
                    keyed[k] = copy.deepcopy(m)
                else:
                    warnings.warn('Keying ReLU since previous layer "%s" is already keyed - Avoid sequential batchnorm and ReLU layers for efficient keying' % kp)
                    keyed[k] = f_module_to_keyedmodule(m, shapes[k]['inshape'], shapes[k]['outshape'], keys[k]['A'], keys[k]['Ainv'])

            elif isinstance(m, nn.Dropout):
                pass   # identity in eval(): absent from the keyed network

            elif shapes[k]['nextlayer'] is not None and (('%s_bn' % k) == shapes[k]['nextlayer'] or 'relu' in shapes[k]['nextlayer']):
                pass   # keyed together with the batchnorm / ReLU that follows
            else:
                keyed[k] = f_module_to_keyedmodule(m, shapes[k]['inshape'], shapes[k]['outshape'], keys[k]['A'], keys[k]['Ainv'])

        self._keynet = nn.Sequential(keyed)
        self._embeddingkey = keys['output']
        self._imagekey = keys['input']
        self._layernames = layernames
        self._outshape = shapes['output']['outshape']

    @classmethod
    def fromlayers(cls, layers, outshape, imagekey=None, embeddingkey=None):
        """Assemble from already keyed layers (OrderedDict name -> KeyedLayer | nn.ReLU): public key-nets, fixtures."""
        self = cls.__new__(cls)
        self._keynet = nn.Sequential(OrderedDict(layers))
        (self._embeddingkey, self._imagekey, self._layernames, self._outshape) = (embeddingkey, imagekey, set(layers.keys()), outshape)
        return self

    def __repr__(self):
        return self._keynet.__repr__()

    def __getattr__(self, attr):
        if attr.startswith('__') or '_keynet' not in self.__dict__:
            raise AttributeError(attr)
        return getattr(self.__dict__['_keynet'], attr)

    # -- the hot path -------------------------------------------------------------------------------------------
    def forward_linear(self, img_cipher):
        """[N, D0+1] -> [N, classes+1]: the nn.Sequential of keynet/system.py:132 with the unkeyed ReLUs fused into the
        producing layer's kernel epilogue.  Stream-ordered on torch's current HIP stream; no host sync."""
        children = list(self._keynet.children())
        y = img_cipher
        i = 0
        while i < len(children):
            c = children[i]
            if isinstance(c, klayer.KeyedLayer):
                fuse = (i + 1 < len(children)) and isinstance(children[i + 1], nn.ReLU)
                y = c.forward(y, fuse_relu=fuse)
                i += 2 if fuse else 1
            elif isinstance(c, nn.ReLU):
                y = _relu_block(y)
                i += 1
            else:
                raise ValueError('unsupported module in a key-net: %s' % str(type(c)))
        return y

    def capture(self, img_cipher):
        """Capture forward_linear for this input shape into a HIP graph (torch.cuda.CUDAGraph on ROCm) and return a callable
        `replay(x) -> [N, classes+1]`.  Small key-nets are launch-bound (LeNet at N=1024: 7 kernels in 0.25 ms); one graph
        launch replaces them.  The operators must already be resident (one eager forward is run first); the returned tensor is
        the graph's static output buffer (clone it to keep a result across replays)."""
        assert img_cipher.is_cuda, 'capture() needs a device tensor'
        static_in = img_cipher.detach().clone()
        # keep the layout the layers expect: a transposed view of a feature-major block
        if not static_in.t().is_contiguous():
            static_in = static_in.t().contiguous().t()
        self.forward_linear(static_in)                      # uploads operators, sizes workspaces (not capturable)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.forward_linear(static_in)                  # warm-up on the capture stream
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = self.forward_linear(static_in)

        def replay(x):
            static_in.copy_(x)
            graph.replay()
            return static_out
        replay.graph = graph
        return replay

    def forward(self, img_cipher, outkey=None):
        """Encrypted image(s) [N, D0+1] -> logits.  N == 1 returns the reference's shape `outshape` = (C,1,1)
        (keynet/system.py:130-133); N > 1 (an extension: the reference cannot) returns (N, C, 1, 1)."""
        outkey = outkey if outkey is not None else self.embeddingkey()
        y = self.forward_linear(img_cipher)
        if outkey is not None:
            y = self.decrypt(y, outkey)
        n = y.shape[0]
        return ktorch.linear_to_affine(y, self._outshape if n == 1 else (n,) + tuple(self._outshape))

    def decrypt(self, y_cipher, outkey=None):
        """Apply the embedding key (the reference's version constructs KeyedLayer(W=outkey), an invalid call:
        keynet/system.py:137; the intent -- one more torchdot with the key -- is what is implemented)."""
        outkey = outkey if outkey is not None else self.embeddingkey()
        if outkey is None:
            return y_cipher
        W = outkey if isinstance(outkey, ksp.SparseMatrix) else ksp.SparseMatrix(outkey)
        return W.torchdot(y_cipher.t()).t()

    def imagekey(self):
        return self._imagekey

    def embeddingkey(self):
        return self._embeddingkey

    def public(self):
        """Strip the private keys before releasing the key-net (keynet/system.py:147-151)."""
        (self._imagekey, self._embeddingkey) = (None, None)
        return self

    def num_parameters(self):
        return sum([c.nnz() for (k, c) in self._keynet.named_children() if isinstance(c, klayer.KeyedLayer)])

    def layers(self):
        return self._layernames


def _relu_block(y):
    """Stand-alone nn.ReLU on an [N, D+1] activation (only when it could not be fused into a producer)."""
    if not torch.cuda.is_available():
        raise _capi.KeynetHipError('keynet_amd: no MI355X visible -- the keyed forward has no CPU fallback')
    src = y.device
    yt = y.detach().t().float()
    yt = (yt if yt.is_cuda else yt.cuda()).contiguous().clone()
    _capi.relu(yt.data_ptr(), yt.shape[0], yt.shape[1], yt.shape[1], torch.cuda.current_stream().cuda_stream)
    out = yt.t()
    return out if src.type == 'cuda' else out.to(src)


class KeyedSensor(klayer.KeyedLayer):
    """The paired sensor: applies the image key A_0 to the homogeneous image (keynet/system.py:160-263)."""

    def __init__(self, inshape, keypair):
        assert isinstance(inshape, tuple) and len(inshape) == 3
        nn.Module.__init__(self)
        (self._encryptkey, self._decryptkey) = keypair
        self._inshape = (1, *inshape)
        self._tensor = None
        self._im = None
        self.W = ksp.SparseMatrix(self._encryptkey)
        self._layertype = 'input'
        self._repr = 'KeyedSensor'
        self._tileshape = None
        self._outshape = None

    def __repr__(self):
        return str('<KeyedSensor: height=%d, width=%d, channels=%d>' % (self._inshape[2], self._inshape[3], self._inshape[1]))

    def load(self, imgfile):
        """Read + resize an image file to the sensor shape as a float tensor in [0,255] (keynet/system.py:183-201 via
        vipy; here PIL bilinear -- the resize interpolation is parity-unpinned)."""
        from PIL import Image
        (C, H, W) = self._inshape[1:]
        im = Image.open(imgfile).convert('L' if C == 1 else 'RGB').resize((W, H), Image.BILINEAR)
        a = np.asarray(im, dtype=np.float32)
        a = a.reshape(H, W, 1) if a.ndim == 2 else a
        self._tensor = torch.as_tensor(np.ascontiguousarray(a.transpose(2, 0, 1))).unsqueeze(0)
        return self

    def fromtensor(self, x):
        if x is not None:
            self._tensor = x.clone().float()
        return self

    def tensor(self):
        return self._tensor.unsqueeze(0) if self._tensor.ndim == 3 else self._tensor

    def astensor(self):
        return self.tensor()

    def totensor(self):
        return self.tensor()

    def keypair(self):
        return (self._encryptkey, self._decryptkey)

    def key(self):
        return self._decryptkey

    def isloaded(self):
        return self._tensor is not None

    def isencrypted(self):
        """Encrypted = homogenised [N, C*H*W+1] (the reference only recognises N == 1; batches are an extension)."""
        return self.isloaded() and self._tensor.ndim == 2 and self._tensor.shape[1] == int(np.prod(self._inshape)) + 1

    def encrypt(self):
        """NxCxHxW -> Nx(C*H*W+1) homogenised and keyed (keynet/system.py:250-255)."""
        assert self.isloaded(), 'Load image first'
        if not self.isencrypted():
            self._tensor = self.forward(ktorch.affine_to_linear(self._tensor))
        return self

    def decrypt(self):
        assert self.isloaded(), 'Load image first'
        if self.isencrypted():
            x_raw = super(KeyedSensor, self).decrypt(self._decryptkey, self._tensor)
            n = x_raw.shape[0]
            self._tensor = ktorch.linear_to_affine(x_raw, (n,) + tuple(self._inshape[1:]))
        return self


class PublicKeyedSensor(KeyedSensor):
    """Sensor with the identity key: only homogenises (keynet/system.py:266-284)."""

    def __init__(self, inshape):
        n = int(np.prod(inshape)) + 1
        super(PublicKeyedSensor, self).__init__(inshape, (sparse_identity_matrix(n), sparse_identity_matrix(n)))

    def __repr__(self):
        return str('<PublicKeyedSensor: height=%d, width=%d, channels=%d>' % (self._inshape[2], self._inshape[3], self._inshape[1]))

    def encrypt(self):
        raise ValueError('PublicKeyedSensor has no encryption keys')

    def decrypt(self):
        raise ValueError('PublicKeyedSensor has no decryption keys')

    def tensor(self):
        assert self.isloaded(), 'Load image first'
        if not self.isencrypted():
            self._tensor = self.forward(ktorch.affine_to_linear(self._tensor))
        return self._tensor


# ------------------------------------------------------------------------------------------------------------------
BACKENDS = ('hip',)


def layergen(module, inshape, outshape, A, Ainv, tileshape=None, backend='hip', direct=None, exact=None):
    """The plug-in seam of the reference (keynet/system.py:303-314): snaps the requested tile to divisors of the
    layer's spatial sizes, then dispatches on `backend`.  The reference accepts only 'scipy'; this build registers
    'hip'.  Anything else raises ValueError('invalid backend ...') exactly like the reference."""
    if exact is None:
        exact = tileshape is None       # untiled key-nets: bit-exact; tiled key-nets: float-key tolerance (MFMA)
    if tileshape is not None:
        tileshape = (find_closest_positive_divisor(outshape[1], tileshape[0]), find_closest_positive_divisor(inshape[1], tileshape[1]))
    if backend == 'hip':
        return klayer.KeyedLayer(module, inshape, outshape, A, Ainv, tileshape=tileshape, direct=direct, exact=exact)
    raise ValueError('invalid backend "%s"' % backend)


def _diag_repeat(block, shape):
    """Block repeated down the diagonal as COO float32 (DiagonalTiledMatrix(...).tocoo() of keynet/system.py:394-395)."""
    return DiagonalTiledMatrix(block, shape=shape).tocoo().astype(np.float32)


def _tolist(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


def keygen(shape, global_geometric, local_geometric, global_photometric, local_photometric, memoryorder='channel', alpha=None, beta=None,
           gamma=None, seed=None, hierarchical_blockshape=None, hierarchical_permute_at_level=None, blocksize=None, tileshape=None, strict=False):
    """(A, Ainv) = C^-1 . p . g . P . G . C for one layer output of `shape` (keynet/system.py:317-469): memory-order
    change C, global geometric G, global photometric P, local (block-repeated) geometric g and photometric p.  Same
    option names, same validation, same order of RNG draws and scipy formats as the reference, so a seeded call returns
    the reference's matrices bit for bit (tests/test_keygen_families.py)."""
    (channels, height, width) = shape
    N = int(np.prod(shape))
    if seed is not None:
        np.random.seed(seed)

    if blocksize is not None:
        if tileshape is not None:
            assert blocksize == tileshape[0] and blocksize == tileshape[1]
        if height == 1 and width == 1:
            (blocksize, H, blocknumel) = (N, N, N)
        else:
            if not strict and (height % blocksize != 0 or width % blocksize != 0):
                assert height == width, 'Image must be square to correct ragged blocksize'
                blocksize = find_closest_positive_divisor(height, blocksize)
            (H, blocknumel) = (height * width, blocksize * blocksize)

    if memoryorder == 'channel':
        (c, cinv) = (sparse_identity_matrix(N), sparse_identity_matrix(N))
    elif memoryorder == 'block':
        assert blocksize is not None
        (c, cinv) = kkeys.channelorder_to_blockorder_matrix(shape, blocksize, withinverse=True)
    else:
        raise ValueError("Invalid memory order '%s' - must be in ['channel', 'block']" % memoryorder)
    (C, Cinv) = (sparse_affine_to_linear(c), sparse_affine_to_linear(cinv))

    if global_geometric == 'identity':
        (G, Ginv) = (sparse_identity_matrix(N), sparse_identity_matrix(N))
    elif global_geometric == 'permutation':
        assert tileshape is None, 'Global permutation is not tile compressible'
        (G, Ginv) = sparse_permutation_matrix(N, withinverse=True)
    elif global_geometric in ('hierarchical_permutation', 'hierarchical_rotation'):
        assert hierarchical_blockshape is not None and hierarchical_permute_at_level is not None
        levels = _tolist(hierarchical_permute_at_level)
        levels = levels if max(height, width) / np.power(2, max(levels)) >= 8 else []
        levels = [] if (height == 1 and width == 1) else levels
        (Ap, Apinv) = kkeys.channelorder_to_pixelorder_matrix((channels, height, width), withinverse=True)
        (G, Ginv) = kkeys.hierarchical_block_permutation_matrix((height, width, channels), hierarchical_blockshape, levels, min_blocksize=8, seed=seed,
                                                                twist=(global_geometric == 'hierarchical_rotation'), withinverse=True, strict=False)
        (G, Ginv) = (Apinv.dot(G).dot(Ap), Apinv.dot(Ginv).dot(Ap))     # CxHxW -> HxWxC -> permute -> CxHxW
        if memoryorder != 'channel':
            (G, Ginv) = (c.dot(G).dot(cinv), c.dot(Ginv).dot(cinv))
    elif global_geometric == 'givens_orthogonal':
        assert alpha is not None
        assert tileshape is None, 'Global givens rotation orthogonal matrix is not tile compressible'
        (G, Ginv) = kkeys.givens_orthogonal(N, int(alpha), withinverse=True)
    else:
        raise ValueError("Invalid global geometric transform '%s'" % global_geometric)
    (G, Ginv) = (sparse_affine_to_linear(G), sparse_affine_to_linear(Ginv))

    if local_geometric == 'identity':
        (g, ginv) = (sparse_identity_matrix(N), sparse_identity_matrix(N))
    elif local_geometric == 'permutation':
        assert blocksize is not None and height == width
        g = _diag_repeat(_diag_repeat(sparse_permutation_matrix(blocknumel), (H, H)), (N, N))   # spatial repeat, then channel repeat
        ginv = g.transpose()
    elif local_geometric == 'doubly_stochastic':
        assert blocksize is not None and alpha is not None and height == width
        assert blocksize < 8192, 'Blocksize %d must be less than 8192, since doubly_stochastic requires the direct inverse of a dense matrix' % blocksize
        (g, ginv) = kkeys.diagonally_dominant_doubly_stochastic(blocknumel, int(alpha), withinverse=True)
        g = DiagonalTiledMatrix(DiagonalTiledMatrix(g, shape=(H, H)).tocoo(), shape=(N, N)).tocoo()
        ginv = DiagonalTiledMatrix(DiagonalTiledMatrix(ginv, shape=(H, H)).tocoo(), shape=(N, N)).tocoo()
    elif local_geometric == 'givens_orthogonal':
        assert alpha is not None and blocksize is not None and height == width
        (g, ginv) = kkeys.givens_orthogonal(blocknumel, int(alpha), withinverse=True)
        (Ap, Apinv) = sparse_permutation_matrix(blocknumel, withinverse=True)
        (g, ginv) = (Ap.dot(g), ginv.dot(Apinv))
        g = _diag_repeat(DiagonalTiledMatrix(g, shape=(H, H)).tocoo(), (N, N))
        ginv = _diag_repeat(DiagonalTiledMatrix(ginv, shape=(H, H)).tocoo(), (N, N))
    else:
        raise ValueError("Invalid local geometric transform '%s'" % local_geometric)
    (g, ginv) = (sparse_affine_to_linear(g), sparse_affine_to_linear(ginv))

    eye_lin = (lambda: sparse_affine_to_linear(sparse_identity_matrix(N)))
    if global_photometric == 'identity':
        (P, Pinv) = (eye_lin(), eye_lin())
    elif global_photometric == 'uniform_random_gain':
        assert tileshape is None, 'Global permutation is not tile compressible'
        assert beta is not None and beta > 0
        (P, Pinv) = kkeys.uniform_random_diagonal(N, beta, bias=1, withinverse=True)
        (P, Pinv) = (sparse_affine_to_linear(P), sparse_affine_to_linear(Pinv))
    elif global_photometric == 'uniform_random_bias':
        assert gamma is not None and gamma > 0
        (P, Pinv) = diagonal_affine_to_linear(sparse_identity_matrix(N), gamma * np.random.rand(N, 1), withinverse=True)
    elif global_photometric == 'linear_bias':
        assert gamma is not None and gamma > 0
        (P, Pinv) = diagonal_affine_to_linear(sparse_identity_matrix(N), (gamma / float(N)) * np.array(range(0, N)).reshape(N, 1), withinverse=True)
    elif global_photometric == 'uniform_random_affine':
        assert tileshape is None, 'Global permutation is not tile compressible'
        assert beta is not None and beta > 0 and gamma is not None and gamma > 0
        Pd = kkeys.uniform_random_diagonal(N, beta, bias=1)
        (P, Pinv) = diagonal_affine_to_linear(Pd, gamma * np.random.rand(N, 1), withinverse=True)
    elif global_photometric == 'blockwise_constant_bias':
        assert gamma is not None and gamma > 0
        assert blocksize is not None
        bias = gamma * np.random.rand(int(np.ceil(N // blocksize)), 1).dot(np.ones((1, blocknumel))).flatten()[0:N].reshape(N, 1)
        (P, Pinv) = diagonal_affine_to_linear(sparse_identity_matrix(N), bias, withinverse=True)
    else:
        raise ValueError("Invalid global photometric transform '%s'" % global_photometric)

    if local_photometric == 'identity':
        (p, pinv) = (eye_lin(), eye_lin())
    elif local_photometric == 'uniform_random_gain':
        assert blocksize is not None
        assert beta is not None and beta > 0
        (p, pinv) = kkeys.uniform_random_diagonal(blocknumel, beta, bias=1, withinverse=True)
        (p, pinv) = (kkeys.block_diagonal(p, (N, N)), kkeys.block_diagonal(pinv, (N, N)))
        (p, pinv) = (sparse_affine_to_linear(p), sparse_affine_to_linear(pinv))
    elif local_photometric == 'uniform_random_bias':
        assert blocksize is not None
        assert gamma is not None and gamma > 0
        bias = np.tile(gamma * np.random.rand(blocknumel), int(np.ceil(N / blocknumel)))[0:N].reshape(N, 1)
        (p, pinv) = diagonal_affine_to_linear(sparse_identity_matrix(N), bias=bias, withinverse=True)
    elif local_photometric == 'uniform_random_affine':
        assert blocksize is not None
        assert beta is not None and beta > 0 and gamma is not None and gamma > 0
        pd = kkeys.uniform_random_diagonal(blocknumel, beta, bias=1)
        bias = np.tile(gamma * np.random.rand(blocknumel), int(np.ceil(N / blocknumel)))[0:N].reshape(N, 1)
        (p, pinv) = diagonal_affine_to_linear(kkeys.block_diagonal(pd, (N, N)), bias=bias, withinverse=True)
    elif local_photometric == 'blockwise_constant_bias':
        raise ValueError('blockwise_constant_bias supported for global_photometric testing only')
    else:
        raise ValueError("Invalid local photometric transform '%s'" % local_photometric)

    A = Cinv.dot(p.dot(g.dot(P.dot(G.dot(C)))))
    Ainv = Cinv.dot(Ginv.dot(Pinv.dot(ginv.dot(pinv.dot(C)))))
    return (A, Ainv)


def ksp_uniform_random_diagonal(n, scale=1, bias=0, eps=1E-6, dtype=np.float32):
    """diag(scale*U[0,1) + eps + bias) and its inverse (keynet/sparse.py:318-321); one np.random.rand(n) draw."""
    D = scipy.sparse.diags(np.array(scale * np.random.rand(n) + eps + bias))
    return (D.astype(dtype), scipy.sparse.diags(1.0 / D.diagonal()).astype(dtype))


def diagonal_affine_to_linear(A, bias=None, withinverse=False, dtype=np.float32):
    """[[A, b], [0, 1]] for diagonal A and, by the rank-one (Woodbury) update, its inverse (keynet/sparse.py:99-119)."""
    assert ksp.is_scipy_sparse(A) and A.shape[0] == A.shape[1]
    n = A.shape[0] + 1
    L = sparse_affine_to_linear(A, bias=bias, dtype=np.float64)
    if not withinverse:
        return L.astype(dtype)
    if bias is not None:
        d = L.diagonal()
        d[-1] = 0.5
        Dinv = scipy.sparse.spdiags(1.0 / d, 0, n, n)
        u = scipy.sparse.csr_matrix(np.vstack((bias, np.array([0.5]))))
        v = scipy.sparse.csr_matrix(np.hstack((np.zeros_like(bias).flatten(), np.array([1.0]))))
        Linv = Dinv - ((Dinv.dot(u).dot(v.dot(Dinv))) / float(1 + (v.dot(Dinv).dot(u).todense())))
    else:
        Linv = scipy.sparse.spdiags(1.0 / L.diagonal(), 0, n, n).tocoo()
    return (L.astype(dtype), Linv.astype(dtype))


def Keynet(inshape, net=None, backend='hip', global_photometric='identity', local_photometric='identity', global_geometric='identity',
           local_geometric='identity', memoryorder='channel', do_output_encryption=False, alpha=None, beta=None, gamma=None,
           hierarchical_blockshape=None, hierarchical_permute_at_level=None, blocksize=None, tileshape=None, direct=None, exact=None):
    """(sensor, model) for `net` under the chosen key family (keynet/system.py:472-486).  ReLU outputs only admit keys
    that commute with ReLU: a 'relu*' layer keeps identity where identity was asked and otherwise falls back to the
    positive-gain / permutation members of the family (keynet/system.py:476-480)."""
    def f_layergen(module, inshape_, outshape_, A, Ainv):
        return layergen(module, inshape_, outshape_, A, Ainv, tileshape=tileshape, backend=backend, direct=direct, exact=exact)

    def f_keypair(layername, shape):
        isrelu = 'relu' in layername
        return keygen(shape,
                      global_photometric=global_photometric if (not isrelu or global_photometric == 'identity') else 'identity',
                      local_photometric=local_photometric if (not isrelu or local_photometric == 'identity') else 'uniform_random_gain',
                      global_geometric=global_geometric if (not isrelu or global_geometric == 'identity') else 'identity',
                      local_geometric=local_geometric if (not isrelu or local_geometric == 'identity') else 'permutation',
                      memoryorder=memoryorder, blocksize=blocksize, tileshape=tileshape, alpha=alpha, beta=beta, gamma=gamma,
                      hierarchical_blockshape=hierarchical_blockshape, hierarchical_permute_at_level=hierarchical_permute_at_level)

    if backend not in BACKENDS:
        raise ValueError('invalid backend "%s"' % backend)
    sensor = KeyedSensor(inshape, f_keypair('input', inshape))
    model = KeyedModel(net, inshape, sensor.key(), f_keypair, f_layergen, do_output_encryption=do_output_encryption) if net is not None else None
    return (sensor, model)


def IdentityKeynet(inshape, net, backend='hip'):
    return Keynet(inshape, net, backend=backend)


def PermutationKeynet(inshape, net, do_output_encryption=False):
    return Keynet(inshape, net, global_geometric='permutation', do_output_encryption=do_output_encryption)


def TiledOrthogonalKeynet(inshape, net, tilesize, hierarchical_permute_at_level=(0, 1), direct=None, exact=None):
    """Hierarchical block permutation + block-local Givens rotations + block-local affine photometric key, block memory
    order (keynet/system.py:504-510): the float-key family (1e-5 contract)."""
    return Keynet(inshape, net, tileshape=(tilesize, tilesize), global_geometric='hierarchical_permutation', hierarchical_blockshape=(2, 2),
                  hierarchical_permute_at_level=hierarchical_permute_at_level, global_photometric='identity', local_geometric='givens_orthogonal',
                  alpha=tilesize, blocksize=tilesize, local_photometric='uniform_random_affine', beta=0.1, gamma=100.0, memoryorder='block',
                  direct=direct, exact=exact)


def TiledIdentityKeynet(inshape, net, tilesize, direct=None, exact=None):
    return Keynet(inshape, net, tileshape=(tilesize, tilesize), direct=direct, exact=exact)


def TiledPermutationKeynet(inshape, net, tilesize, direct=None, exact=None):
    return Keynet(inshape, net, local_geometric='permutation', tileshape=(tilesize, tilesize), blocksize=tilesize, direct=direct, exact=exact)
