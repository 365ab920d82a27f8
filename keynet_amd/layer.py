"""KeyedLayer: one keyed linear layer  W_hat = A . W . A_prev^-1  of a key-net (mirror of keynet/layer.py:15-106).

Construction (host, offline) restates the reference's keying; `forward` -- the drop-in boundary -- hands the activation
block to the HIP operator stored in `self.W` (its torchdot), optionally fusing the ReLU that follows.
"""
import numpy as np
import scipy.sparse
import torch
from torch import nn
import torch.nn.functional as F

from .globals import verbose
from . import sparse as ksp
from . import direct as kdirect
from .sparse import SparseMatrix, sparse_toeplitz_conv2d, sparse_toeplitz_avgpool2d
from .torch import affine_to_linear_matrix


class KeyedLayer(nn.Module):
    DIRECT_THRESHOLD = 20000000   # Toeplitz entries above which tiled conv/pool layers are keyed in factored form

    def __init__(self, module, inshape, outshape, A, Ainv, tileshape=None, direct=None, exact=None):
        """`direct`: None = automatic (factored, Toeplitz-free keying for tiled conv/avgpool layers whose Toeplitz matrix
        would exceed DIRECT_THRESHOLD entries -- the reference route cannot build those at all); True / False force it.
        `exact`: True = every product in the reference's accumulation order and rounding (bit-exact with scipy);
        False = float-key tolerance (1e-5): conv-taps and large dense operators run on the matrix cores.  None = exact
        for untiled layers (the permutation key-nets), tolerance for tiled ones (BASELINE north_star)."""
        super(KeyedLayer, self).__init__()
        self._exact = (tileshape is None) if exact is None else bool(exact)
        self._layertype = str(type(module))
        self._tileshape = tileshape
        self._inshape = inshape
        self._outshape = outshape

        if isinstance(module, nn.Conv2d):
            assert len(module.kernel_size) == 1 or (len(module.kernel_size) == 2 and module.kernel_size[0] == module.kernel_size[1]), 'Kernel must be square'
            assert len(module.stride) == 1 or (len(module.stride) == 2 and module.stride[0] == module.stride[1]), 'Strides must be isotropic'
            assert len(inshape) == 3, 'Inshape must be (C,H,W) for the shape of the tensor at the input to this layer'
            assert module.padding[0] == module.kernel_size[0] // 2 and module.padding[1] == module.kernel_size[1] // 2, 'Padding is assumed to be equal to (kernelsize-1)/2'
            stride = module.stride[0]
            self._repr = 'Conv2d: in_channels=%d, out_channels=%d, kernel_size=%s, stride=%s' % (module.in_channels, module.out_channels, str(module.kernel_size), str(stride))
            if direct is None:
                direct = tileshape is not None and kdirect.toeplitz_entries('conv', inshape, outshape, module.kernel_size[0]) > self.DIRECT_THRESHOLD
            if direct:
                kw = kdirect.keyed_conv_taps(module.weight.detach().numpy(), module.bias.detach().numpy(), inshape, outshape, stride, A, Ainv)
                W = ksp.Conv2dTiledMatrix.fromtaps(tileshape=tileshape, **kw)
            else:
                W = sparse_toeplitz_conv2d(inshape, module.weight.detach().numpy(), bias=module.bias.detach().numpy(), stride=stride)
                W = A.dot(W).dot(Ainv)    # the key: same association as the reference so the stored order matches (keynet/layer.py:35)
                if tileshape is not None:
                    W = ksp.Conv2dTiledMatrix(W, self._inshape, self._outshape, self._tileshape, bias=True, sanitycheck=False)
            self.W = W

        elif isinstance(module, nn.ReLU):
            self._repr = 'ReLU'
            self.W = A.dot(Ainv)

        elif isinstance(module, nn.AvgPool2d):
            assert isinstance(module.kernel_size, int) or (len(module.kernel_size) == 2 and module.kernel_size[0] == module.kernel_size[1]), 'Kernel must be square'
            assert isinstance(module.stride, int) or (len(module.stride) == 2 and module.stride[0] == module.stride[1]), 'Strides must be isotropic'
            assert len(inshape) == 3, 'Inshape must be (C,H,W) for the shape of the tensor at the input to this layer'
            stride = module.stride if isinstance(module.stride, int) else module.stride[0]
            kernel_size = module.kernel_size if isinstance(module.kernel_size, int) else module.kernel_size[0]
            self._repr = 'AvgPool2d: kernel_size=%s, stride=%s' % (str(kernel_size), str(stride))
            if direct is None:
                direct = tileshape is not None and kdirect.toeplitz_entries('pool', inshape, (inshape[0],) + tuple(outshape[1:]), kernel_size) > self.DIRECT_THRESHOLD
            if direct:
                W = kdirect.keyed_avgpool_csr(inshape[0], (inshape[1], inshape[2]), kernel_size, stride, A, Ainv)
            else:
                W = sparse_toeplitz_avgpool2d(inshape, (inshape[0], inshape[0], kernel_size, kernel_size), stride)
                W = A.dot(W).dot(Ainv) if A is not None else W.dot(Ainv)
            if tileshape is not None:
                W = ksp.TiledMatrix(W, self._tileshape)
            self.W = W

        elif isinstance(module, nn.Linear):
            self._repr = 'Linear: in_features=%d, out_features=%d' % (module.in_features, module.out_features)
            W = scipy.sparse.coo_matrix(affine_to_linear_matrix(module.weight, module.bias).detach().numpy()).transpose()
            self.W = W.dot(Ainv) if A is None else A.dot(W).dot(Ainv)

        elif isinstance(module, nn.BatchNorm2d):
            raise ValueError('batchnorm layer should be named "mylayer_bn" for batchnorm of "mylayer" and should come right before "mylayer" to merge keyed layers')
        elif isinstance(module, nn.Dropout):
            raise ValueError('dropout layer should be skipped during keying and removed from final network')
        else:
            raise ValueError('unsupported layer type "%s"' % str(type(module)))

        if not isinstance(self.W, SparseMatrix):
            self.W = SparseMatrix(self.W)

    @classmethod
    def fromoperator(cls, W, layertype, inshape=None, outshape=None, repr_=None, exact=None):
        """Wrap an already keyed operator (a public key-net loaded from a neutral file, a fixture, a direct build)."""
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self._exact = (not isinstance(W, ksp.Conv2dTiledMatrix)) if exact is None else bool(exact)
        (self._layertype, self._tileshape, self._inshape, self._outshape) = (layertype, None, inshape, outshape)
        self._repr = repr_ if repr_ is not None else layertype
        self.W = W if isinstance(W, SparseMatrix) else SparseMatrix(W)
        return self

    def extra_repr(self):
        return str('<%s, backend=hip, shape=%s, nnz=%d>' % (self._repr, str(self.W.shape), self.nnz()))

    def iskeyedrelu(self):
        return 'ReLU' in self._layertype

    def forward(self, x_affine, fuse_relu=False):
        """[N, Din+1] -> [N, Dout+1] (keynet/layer.py:88-93).  The result is a transposed view of the feature-major
        [Dout+1, N] block the kernel wrote, so the next layer's x.t() is free.  `fuse_relu` folds the unkeyed nn.ReLU
        that follows this layer in the key-net (keynet/system.py:92) into the kernel epilogue."""
        if verbose():
            print('[keynet_amd.layer]: forward %s' % str(self))
        y = self.W.torchdot(x_affine.t(), relu=(fuse_relu or self.iskeyedrelu()), exact=getattr(self, '_exact', True)).t()
        return y

    def decrypt(self, Ainv, x_affine):
        """Apply a decryption key to this layer's output (keynet/layer.py:95-99)."""
        if scipy.sparse.issparse(Ainv):
            Ainv = SparseMatrix(Ainv)
        return Ainv.torchdot(x_affine.t()).t()

    def nnz(self):
        assert self.W is not None, 'Layer not keyed'
        return self.W.nnz()
