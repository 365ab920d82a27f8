"""Key-matrix generators of the key-net families (host, offline): random sparse invertible matrices with their inverses.

Restated from the reference's generators (keynet/sparse.py:53-84, 215-353; keynet/blockpermute.py:6-79) with the Python
per-entry loops vectorised.  They draw from numpy's GLOBAL RNG in the same order and hand scipy the same formats, so
that under one np.random.seed the composed keys of keynet_amd.system.keygen equal the reference's bit for bit
(tests/test_keygen_families.py against tests/golden/keygen_cases.npz).
"""
import warnings
import numpy as np
import scipy.sparse
from sklearn.preprocessing import normalize

from .util import blockview, find_closest_positive_divisor
from .sparse import sparse_permutation_matrix, sparse_identity_matrix, sparse_affine_to_linear, DiagonalTiledMatrix, is_scipy_sparse


def channelorder_to_pixelorder_matrix(shape, withinverse=False):
    """Permutation taking a CxHxW-flattened image to HxWxC order (keynet/sparse.py:53-62)."""
    n = int(np.prod(shape))
    cols = np.moveaxis(np.arange(n).reshape(shape), 0, 2).flatten()
    P = scipy.sparse.coo_matrix((np.ones(n, dtype=np.int64), (np.arange(n), cols)), shape=(n, n), dtype=np.float32)
    return P if not withinverse else (P, P.transpose())


def channelorder_to_blockorder_matrix(shape, blocksize, withinverse=True):
    """Permutation taking CxHxW order to Cx(H/b)x(W/b)xbxb 'block' memory order (keynet/sparse.py:65-84)."""
    assert isinstance(shape, tuple) and len(shape) == 3, 'Shape must be (C,H,W) tuple'
    (C, H, W) = shape
    if (H * W) % blocksize != 0:
        warnings.warn('[keynet_amd.keys]: ragged blockorder for blocksize=%d and shape=%s' % (blocksize, str(shape)))
    (Hp, Wp) = (int(blocksize * np.ceil(H / float(blocksize))), int(blocksize * np.ceil(W / float(blocksize))))
    order = blockview(np.arange(Hp * Wp).reshape(Hp, Wp), blocksize).flatten()[0:H * W]
    rows = (np.arange(H * W)[None, :] + (np.arange(C) * H * W)[:, None]).ravel()
    cols = (order[None, :] + (np.arange(C) * H * W)[:, None]).ravel()
    A = scipy.sparse.coo_matrix((np.ones(len(rows), dtype=np.int64), (rows, cols)), dtype=np.float32).tocsr()
    return A if not withinverse else (A, A.transpose())


def block_diagonal(mat, shape):
    """One sparse block repeated down the diagonal of a `shape` matrix, clipped at the border, as COO in block-by-block
    entry order (keynet/sparse.py:215-235 for a single block)."""
    (U, V) = shape
    b = scipy.sparse.coo_matrix(mat)
    (bh, bw) = mat.shape
    nblk = min(len(range(0, U, bh)), len(range(0, V, bw)))
    rows = (b.row[None, :] + (np.arange(nblk) * bh)[:, None]).ravel()
    cols = (b.col[None, :] + (np.arange(nblk) * bw)[:, None]).ravel()
    data = np.tile(b.data, nblk)
    keep = (rows < U) & (cols < V)
    return scipy.sparse.coo_matrix((data[keep], (rows[keep], cols[keep])), shape=(U, V)).asformat('coo')


def uniform_random_diagonal(n, scale=1, bias=0, eps=1E-6, dtype=np.float32, withinverse=False):
    """diag(scale*U[0,1) + eps + bias) (keynet/sparse.py:318-321); one np.random.rand(n) draw."""
    D = scipy.sparse.diags(np.array(scale * np.random.rand(n) + eps + bias))
    return (D.astype(dtype), scipy.sparse.diags(1.0 / D.diagonal()).astype(dtype)) if withinverse else D.astype(dtype)


def _givens_plan(n, k_iter):
    """RNG protocol of the reference's 'balanced' Givens product (keynet/sparse.py:288-309), separated from the algebra: per
    rotation one np.random.rand() for the angle, then -- only when fewer than two indices are left -- one
    np.random.permutation(n) pushed UNDER the leftovers; the plane is the two indices popped off the top.
    Returns [(i, j, theta)] in application order."""
    (plan, pool) = ([], [])
    for _ in range(k_iter):
        theta = 2 * np.pi * np.random.rand()
        if len(pool) < 2:
            pool = np.random.permutation(n).tolist() + pool
        (i, j) = (pool.pop(), pool.pop())
        plan.append((i, j, theta))
    return plan


def _plane_rotation(n, i, j, theta):
    """The n x n rotation by theta in the (i, j) plane as canonical float64 CSR, assembled from index arithmetic."""
    (c, s) = (np.cos(theta), np.sin(theta))
    keep = np.setdiff1d(np.arange(n), (i, j))
    rows = np.concatenate((keep, (i, i, j, j)))
    cols = np.concatenate((keep, (i, j, i, j)))
    vals = np.concatenate((np.ones(len(keep)), (c, -s, s, c)))
    return scipy.sparse.csr_matrix((vals, (rows, cols)), shape=(n, n))


def givens_orthogonal(n, k_iter, withinverse=False, dtype=np.float32):
    """Product G_k ... G_2 G_1 of k_iter random Givens rotations (float64 products, cast at the end); the inverse of an
    orthogonal matrix is its transpose."""
    assert n >= 2
    S = None
    for (i, j, theta) in _givens_plan(n, k_iter):
        G = _plane_rotation(n, i, j, theta)
        S = G if S is None else G.dot(S)
    return S.astype(dtype) if not withinverse else (S.astype(dtype), S.transpose().astype(dtype))


def _band_offsets(k):
    """Diagonal offsets of a k-banded matrix, main diagonal first, then the others in ascending order (for even k the extra
    diagonal goes below the main one)."""
    lo = -((k - 1) // 2) if k % 2 else -(k // 2)
    hi = (k - 1) // 2 if k % 2 else k // 2 - 1
    return [0] + [o for o in range(lo, hi + 1) if o != 0]


def _sinkhorn(A, n_iter):
    """Alternate column / row l1 normalisation (sklearn's normalize, as the reference uses: its summation order is part of the
    bit-exact contract of the key values)."""
    for _ in range(n_iter):
        A = normalize(normalize(A, norm='l1', axis=0), norm='l1', axis=1)
    return A


def diagonally_dominant_doubly_stochastic(n, k, n_iter=100, withinverse=False):
    """k-banded, diagonally dominant, (numerically) doubly stochastic matrix, hidden between two random permutations; the inverse
    is dense (keynet/sparse.py:335-353).  Draw order: one rand(k, n) for the bands, then the left and the right permutation."""
    bands = np.random.rand(k, n)
    bands[0] = np.maximum(bands[0], bands[1:].sum(axis=0) + 0.1)       # main diagonal dominates the sum of the others
    bands /= bands.sum(axis=0, keepdims=True)
    A = _sinkhorn(scipy.sparse.spdiags(bands, _band_offsets(k), n, n, format='csr'), 10 if k <= 3 else n_iter)
    (left, right) = (sparse_permutation_matrix(n), sparse_permutation_matrix(n))
    A = left.dot(A).dot(right)
    if not withinverse:
        return A
    if n > 8096:
        warnings.warn('direct inverse of large matrix (%dx%d)' % (n, n))
    return (A, scipy.sparse.coo_matrix(np.linalg.inv(A.todense())))


def _permuted_pixel_indices(P, blockshape, levels, min_blocksize, twist, strict):
    """Index form of the reference's top-down hierarchical block permutation (keynet/blockpermute.py:22-68).  P is an [h, w]
    array holding, per position, the flat index of the SOURCE pixel currently there; the result is P after the permutation of
    this region.  Level 0 permutes the block rows and the block columns of the region independently (two np.random.permutation
    draws; a 'twist' is one np.random.rand() choosing a quarter turn), deeper levels descend into every block in row-major
    order -- which fixes the order of the draws."""
    levels = [int(l) for l in levels]
    if not levels:
        return P.copy()
    (h, w) = P.shape
    (bh, bw) = blockshape
    if h % bh != 0 and w % bw != 0:
        if strict:
            raise ValueError('Recursive image size %s and block layout %s must be divisible' % (str((h, w)), str(blockshape)))
        (bh, bw) = (find_closest_positive_divisor(h, bh), find_closest_positive_divisor(w, bw))
    (ch, cw) = (h // bh, w // bw)
    out = P.copy()
    if 0 in levels:
        if twist:
            out = np.rot90(out, k=(1 if np.random.rand() > 0.5 else 3)).copy()
        else:
            assert h % ch == 0 and w % cw == 0, 'Blocksize must be evenly divisible with image shape'
            (to_row, to_col) = (np.random.permutation(h // ch), np.random.permutation(w // cw))   # source block a lands on block to_*[a]
            blocks = out.reshape(h // ch, ch, w // cw, cw)
            out = blocks[np.argsort(to_row)][:, :, np.argsort(to_col)].reshape(h, w)
    if levels == [0] or max(levels) <= 0:
        return out
    if min(ch, cw) < min_blocksize:
        raise ValueError('Recursive blockshape=%s < minimum blockshape=%d' % (str((ch, cw)), min_blocksize))
    deeper = [l - 1 for l in levels]
    for i in range(0, h, ch):
        for j in range(0, w, cw):
            out[i:i + ch, j:j + cw] = _permuted_pixel_indices(out[i:i + ch, j:j + cw], (bh, bw), deeper, min_blocksize, twist, True)
    return out


def hierarchical_block_permute(img, blockshape, permute_at_level, min_blocksize=8, twist=False, strict=True):
    """The hierarchically block-permuted (or twisted) HxWxC image: every pixel moves with all its channels."""
    (H, W) = img.shape[0:2]
    src = _permuted_pixel_indices(np.arange(H * W).reshape(H, W), blockshape, list(np.atleast_1d(permute_at_level)), min_blocksize, twist, strict)
    return img.reshape((H * W,) + img.shape[2:])[src.ravel()].reshape(src.shape + img.shape[2:])


def hierarchical_block_permutation_matrix(imgshape, blockshape, permute_at_level, min_blocksize=8, seed=None, twist=False, withinverse=False, strict=True):
    """P with P.dot(img.flatten()).reshape(shape) == hierarchical_block_permute(img) for an HxWxC image
    (keynet/blockpermute.py:71-79): row (y, x, c) picks source pixel src[y, x], channel c."""
    if seed is not None:
        np.random.seed(seed)
    (H, W, C) = imgshape
    src = _permuted_pixel_indices(np.arange(H * W).reshape(H, W), blockshape, list(np.atleast_1d(permute_at_level)), min_blocksize, twist, strict)
    cols = (src.reshape(-1, 1) * C + np.arange(C)[None, :]).ravel()
    n = H * W * C
    P = scipy.sparse.coo_matrix((np.ones(n, dtype=np.int64), (np.arange(n), cols)), shape=(n, n), dtype=np.float32)
    return P if not withinverse else (P, P.transpose())


# ------------------------------------------------------------------------------------------------------------------
# keygen: one layer's key pair, composed from five stages
def diagonal_affine_to_linear(A, bias=None, withinverse=False, dtype=np.float32):
    """L = [[D, b], [0, 1]] for a diagonal D, and its inverse in closed form [[D^-1, -D^-1 b], [0, 1]] (float64, cast at the end).
    The reference reaches the same numbers through a rank-one (Woodbury) update (keynet/sparse.py:99-119); its last column is
    -((1/d_i) * b_i) -- the reciprocal first, then the product -- which is the association kept here."""
    assert is_scipy_sparse(A) and A.shape[0] == A.shape[1]
    m = A.shape[0]
    L = sparse_affine_to_linear(A, bias=bias, dtype=np.float64)
    if not withinverse:
        return L.astype(dtype)
    rdiag = 1.0 / np.asarray(A.diagonal(), dtype=np.float64)
    if bias is None:
        return (L.astype(dtype), scipy.sparse.spdiags(np.concatenate((rdiag, [1.0])), 0, m + 1, m + 1).tocoo().astype(dtype))
    lastcol = -(rdiag * np.asarray(bias, dtype=np.float64).ravel())
    nz = np.flatnonzero(lastcol)                                       # exact zeros are not stored
    rows = np.concatenate((np.arange(m), nz, [m]))
    cols = np.concatenate((np.arange(m), np.full(len(nz), m), [m]))
    vals = np.concatenate((rdiag, lastcol[nz], [1.0]))
    return (L.astype(dtype), scipy.sparse.csr_matrix((vals, (rows, cols)), shape=(m + 1, m + 1)).astype(dtype))


class _Ctx(object):
    """What the stages of one keygen call share: the layer shape and the (possibly snapped) block geometry."""
    def __init__(self, shape, blocksize, tileshape, strict, alpha, beta, gamma, memoryorder, seed):
        (self.channels, self.height, self.width) = shape
        self.shape = shape
        self.N = int(np.prod(shape))
        (self.alpha, self.beta, self.gamma, self.tileshape, self.memoryorder, self.seed) = (alpha, beta, gamma, tileshape, memoryorder, seed)
        (self.blocksize, self.plane, self.blocknumel) = (blocksize, None, None)
        if blocksize is not None:
            if tileshape is not None:
                assert blocksize == tileshape[0] == tileshape[1], 'blocksize and tileshape disagree'
            if self.height == 1 and self.width == 1:                      # a vector (fc output): one global block
                (self.blocksize, self.plane, self.blocknumel) = (self.N, self.N, self.N)
            else:
                if not strict and (self.height % blocksize or self.width % blocksize):
                    assert self.height == self.width, 'ragged blocksize needs a square image'
                    self.blocksize = find_closest_positive_divisor(self.height, blocksize)
                (self.plane, self.blocknumel) = (self.height * self.width, self.blocksize * self.blocksize)

    def eye(self):
        return sparse_identity_matrix(self.N)

    def need(self, **named):
        for (k, v) in named.items():
            assert v is not None, 'option "%s" is required for this key family' % k

    def not_tiled(self, what):
        assert self.tileshape is None, '%s is not tile compressible' % what

    def spread(self, block):
        """block [blocknumel^2] -> repeated over the plane, then over the channels (COO)."""
        return DiagonalTiledMatrix(DiagonalTiledMatrix(block, shape=(self.plane, self.plane)).tocoo(), shape=(self.N, self.N)).tocoo()

    def tiled_bias(self):
        return np.tile(self.gamma * np.random.rand(self.blocknumel), int(np.ceil(self.N / self.blocknumel)))[0:self.N].reshape(self.N, 1)


def _stage_memoryorder(c):
    if c.memoryorder == 'channel':
        return (c.eye(), c.eye())
    if c.memoryorder == 'block':
        c.need(blocksize=c.blocksize)
        return channelorder_to_blockorder_matrix(c.shape, c.blocksize, withinverse=True)
    raise ValueError("unknown memoryorder '%s' (channel | block)" % c.memoryorder)


def _gg_permutation(c, order):
    c.not_tiled('a global permutation')
    return sparse_permutation_matrix(c.N, withinverse=True)


def _gg_hierarchical(twist):
    def build(c, order):
        c.need(hierarchical_blockshape=c.hblockshape, hierarchical_permute_at_level=c.hlevels)
        levels = list(c.hlevels) if isinstance(c.hlevels, (list, tuple)) else [c.hlevels]
        if max(c.height, c.width) / np.power(2, max(levels)) < 8 or (c.height == 1 and c.width == 1):
            levels = []
        (to_px, from_px) = channelorder_to_pixelorder_matrix((c.channels, c.height, c.width), withinverse=True)
        (Q, Qinv) = hierarchical_block_permutation_matrix((c.height, c.width, c.channels), c.hblockshape, levels, min_blocksize=8, seed=c.seed,
                                                          twist=twist, withinverse=True, strict=False)
        (Q, Qinv) = (from_px.dot(Q).dot(to_px), from_px.dot(Qinv).dot(to_px))
        if c.memoryorder != 'channel':
            (o, oinv) = order
            (Q, Qinv) = (o.dot(Q).dot(oinv), o.dot(Qinv).dot(oinv))
        return (Q, Qinv)
    return build


def _gg_givens(c, order):
    c.need(alpha=c.alpha)
    c.not_tiled('a global Givens rotation')
    return givens_orthogonal(c.N, int(c.alpha), withinverse=True)


def _lg_permutation(c):
    assert c.blocksize is not None and c.height == c.width, 'local permutation needs a blocksize and a square image'
    fwd = c.spread(sparse_permutation_matrix(c.blocknumel)).astype(np.float32)
    return (fwd, fwd.transpose())


def _lg_doubly_stochastic(c):
    assert c.blocksize is not None and c.alpha is not None and c.height == c.width
    assert c.blocksize < 8192, 'doubly_stochastic inverts a dense blocksize^2 matrix: blocksize %d is too large' % c.blocksize
    (m, minv) = diagonally_dominant_doubly_stochastic(c.blocknumel, int(c.alpha), withinverse=True)
    return (c.spread(m), c.spread(minv))


def _lg_givens(c):
    assert c.alpha is not None and c.blocksize is not None and c.height == c.width
    (rot, rotinv) = givens_orthogonal(c.blocknumel, int(c.alpha), withinverse=True)
    (shuf, shufinv) = sparse_permutation_matrix(c.blocknumel, withinverse=True)
    return (c.spread(shuf.dot(rot)).astype(np.float32), c.spread(rotinv.dot(shufinv)).astype(np.float32))


def _lin(pair):
    return (sparse_affine_to_linear(pair[0]), sparse_affine_to_linear(pair[1]))


def _gp_gain(c):
    c.not_tiled('a global gain')
    assert c.beta is not None and c.beta > 0
    return _lin(uniform_random_diagonal(c.N, c.beta, bias=1, withinverse=True))


def _gp_bias(c):
    assert c.gamma is not None and c.gamma > 0
    return diagonal_affine_to_linear(c.eye(), c.gamma * np.random.rand(c.N, 1), withinverse=True)


def _gp_linear_bias(c):
    assert c.gamma is not None and c.gamma > 0
    return diagonal_affine_to_linear(c.eye(), (c.gamma / float(c.N)) * np.array(range(0, c.N)).reshape(c.N, 1), withinverse=True)


def _gp_affine(c):
    c.not_tiled('a global affine photometric key')
    assert c.beta is not None and c.beta > 0 and c.gamma is not None and c.gamma > 0
    gain = uniform_random_diagonal(c.N, c.beta, bias=1)
    return diagonal_affine_to_linear(gain, c.gamma * np.random.rand(c.N, 1), withinverse=True)


def _gp_blockwise_bias(c):
    assert c.gamma is not None and c.gamma > 0 and c.blocksize is not None
    b = c.gamma * np.random.rand(int(np.ceil(c.N // c.blocksize)), 1).dot(np.ones((1, c.blocknumel))).flatten()[0:c.N].reshape(c.N, 1)
    return diagonal_affine_to_linear(c.eye(), b, withinverse=True)


def _lp_gain(c):
    assert c.blocksize is not None and c.beta is not None and c.beta > 0
    (d, dinv) = uniform_random_diagonal(c.blocknumel, c.beta, bias=1, withinverse=True)
    return _lin((block_diagonal(d, (c.N, c.N)), block_diagonal(dinv, (c.N, c.N))))


def _lp_bias(c):
    assert c.blocksize is not None and c.gamma is not None and c.gamma > 0
    return diagonal_affine_to_linear(c.eye(), bias=c.tiled_bias(), withinverse=True)


def _lp_affine(c):
    assert c.blocksize is not None and c.beta is not None and c.beta > 0 and c.gamma is not None and c.gamma > 0
    d = uniform_random_diagonal(c.blocknumel, c.beta, bias=1)
    return diagonal_affine_to_linear(block_diagonal(d, (c.N, c.N)), bias=c.tiled_bias(), withinverse=True)


_GLOBAL_GEOMETRIC = {'permutation': _gg_permutation, 'hierarchical_permutation': _gg_hierarchical(False), 'hierarchical_rotation': _gg_hierarchical(True),
                     'givens_orthogonal': _gg_givens}
_LOCAL_GEOMETRIC = {'permutation': _lg_permutation, 'doubly_stochastic': _lg_doubly_stochastic, 'givens_orthogonal': _lg_givens}
_GLOBAL_PHOTOMETRIC = {'uniform_random_gain': _gp_gain, 'uniform_random_bias': _gp_bias, 'linear_bias': _gp_linear_bias, 'uniform_random_affine': _gp_affine,
                       'blockwise_constant_bias': _gp_blockwise_bias}
_LOCAL_PHOTOMETRIC = {'uniform_random_gain': _lp_gain, 'uniform_random_bias': _lp_bias, 'uniform_random_affine': _lp_affine}


def keygen(shape, global_geometric, local_geometric, global_photometric, local_photometric, memoryorder='channel', alpha=None, beta=None,
           gamma=None, seed=None, hierarchical_blockshape=None, hierarchical_permute_at_level=None, blocksize=None, tileshape=None, strict=False):
    """Key pair (A, Ainv) of one layer output of `shape` = (C,H,W):   A = O^-1 . lp . lg . gp . gg . O

        O   memory-order change ('channel' | 'block')            gg  global geometric key      gp  global photometric key
        lg  local (block-repeated, channel-replicated) geometric key                              lp  local photometric key

    Same option names and accepted values as the reference's keygen (keynet/system.py:317-469); the stages draw from
    numpy's global RNG in the reference's order (gg, lg, gp, lp) and hand scipy the same formats, so a seeded call
    returns the reference's matrices bit for bit (tests/test_keygen_families.py, 20 cases).  Unknown values raise
    ValueError; a tile-incompatible choice with `tileshape` set raises AssertionError, as there."""
    if seed is not None:
        np.random.seed(seed)
    c = _Ctx(tuple(shape), blocksize, tileshape, strict, alpha, beta, gamma, memoryorder, seed)
    (c.hblockshape, c.hlevels) = (hierarchical_blockshape, hierarchical_permute_at_level)

    order = _stage_memoryorder(c)
    (O, Oinv) = _lin(order)

    def stage(kind, name, table, linearise, *extra):
        if name == 'identity':
            return _lin((c.eye(), c.eye()))
        if name not in table:
            raise ValueError("unknown %s key '%s' (identity | %s)" % (kind, name, ' | '.join(sorted(table))))
        pair = table[name](c, *extra)
        return _lin(pair) if linearise else pair

    (GG, GGinv) = stage('global geometric', global_geometric, _GLOBAL_GEOMETRIC, True, order)
    (LG, LGinv) = stage('local geometric', local_geometric, _LOCAL_GEOMETRIC, True)
    (GP, GPinv) = stage('global photometric', global_photometric, _GLOBAL_PHOTOMETRIC, False)
    if local_photometric == 'blockwise_constant_bias':
        raise ValueError("'blockwise_constant_bias' exists as a global photometric key only")
    (LP, LPinv) = stage('local photometric', local_photometric, _LOCAL_PHOTOMETRIC, False)

    A = Oinv.dot(LP.dot(LG.dot(GP.dot(GG.dot(O)))))
    Ainv = Oinv.dot(GGinv.dot(GPinv.dot(LGinv.dot(LPinv.dot(O)))))
    return (A, Ainv)
