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
from .sparse import sparse_permutation_matrix


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


def givens_orthogonal(n, k_iter, withinverse=False, dtype=np.float32):
    """Product of k_iter random Givens rotations on index pairs drawn without replacement ('balanced';
    keynet/sparse.py:288-309).  Per rotation: theta first, then a fresh permutation whenever fewer than two indices are left."""
    assert n >= 2
    S = None
    pool = []
    for _ in range(0, k_iter):
        theta = np.random.rand() * 2 * np.pi
        pool = np.random.permutation(range(0, n)).tolist() + pool if len(pool) <= 1 else pool
        G = scipy.sparse.eye(n).todok()
        (i, j) = (pool.pop(), pool.pop())
        G[i, i] = np.cos(theta)
        G[i, j] = -np.sin(theta)
        G[j, i] = np.sin(theta)
        G[j, j] = np.cos(theta)
        S = G.dot(S) if S is not None else G
    return S.astype(dtype) if not withinverse else (S.astype(dtype), S.transpose().astype(dtype))


def diagonally_dominant_doubly_stochastic(n, k, n_iter=100, withinverse=False):
    """Banded (k diagonals) diagonally dominant matrix, Sinkhorn-normalised to doubly stochastic, conjugated by two
    random permutations; dense inverse (keynet/sparse.py:335-353)."""
    n_iter = 10 if k <= 3 else n_iter
    d = np.random.rand(k, n)
    d[0, :] = np.maximum(d[0, :], np.sum(d[1:, :], axis=0) + 0.1)
    d = d / np.sum(d, axis=0).reshape(1, n)
    offs = list(range(-((k - 1) // 2), 1 + ((k - 1) // 2))) if k % 2 == 1 else list(range(-(k // 2), k // 2))
    offs.remove(0)
    offs = [0] + offs
    A = scipy.sparse.spdiags(d, offs, n, n, format='csr')
    for _ in range(0, n_iter):
        A = normalize(A, norm='l1', axis=0)
        A = normalize(A, norm='l1', axis=1)
    A = sparse_permutation_matrix(n).dot(A).dot(sparse_permutation_matrix(n))
    if withinverse and n > 8096:
        warnings.warn('direct inverse of large matrix (%dx%d)' % (n, n))
    return A if not withinverse else (A, scipy.sparse.coo_matrix(np.linalg.inv(A.todense())))


def _block_permute(img, cropshape):
    """Permute the non-overlapping cropshape blocks of an HxWxC image, rows and columns independently
    (keynet/blockpermute.py:6-19): two np.random.permutation draws."""
    assert img.shape[0] % cropshape[0] == 0 and img.shape[1] % cropshape[1] == 0, 'Blocksize must be evenly divisible with image shape'
    (ri, cj) = (np.arange(0, img.shape[0], cropshape[0]), np.arange(0, img.shape[1], cropshape[1]))
    (U, V) = (np.random.permutation(ri), np.random.permutation(cj))
    out = np.copy(img)
    for (i, ip) in zip(ri, U):
        for (j, jp) in zip(cj, V):
            out[ip:ip + cropshape[0], jp:jp + cropshape[1]] = img[i:i + cropshape[0], j:j + cropshape[1]]
    return out


def hierarchical_block_permute(img, blockshape, permute_at_level, min_blocksize=8, twist=False, strict=True):
    """Top-down hierarchical block permutation (or 90-degree 'twist') of an HxWxC image: level 0 acts on the whole image
    split into `blockshape` blocks, level k on each block of level k-1 (keynet/blockpermute.py:22-68)."""
    if len(permute_at_level) == 0 or blockshape == img.shape:
        return np.copy(img)
    if (img.shape[0] % blockshape[0] != 0 and img.shape[1] % blockshape[1] != 0):
        if strict:
            raise ValueError('Recursive image size %s and block layout %s must be divisible' % (str(img.shape[0:2]), str(blockshape)))
        blockshape = (find_closest_positive_divisor(img.shape[0], blockshape[0]), find_closest_positive_divisor(img.shape[1], blockshape[1]))
    cropshape = (img.shape[0] // blockshape[0], img.shape[1] // blockshape[1])
    out = np.copy(img)
    if 0 in permute_at_level:
        if twist:
            out = np.rot90(out, k=(1 if np.random.rand() > 0.5 else 3))
        else:
            out = _block_permute(out, cropshape)
    if len(permute_at_level) == 1 and permute_at_level[0] == 0:
        return out
    for i in range(0, img.shape[0], cropshape[0]):
        for j in range(0, img.shape[1], cropshape[1]):
            sub = out[i:i + cropshape[0], j:j + cropshape[1]]
            if min(cropshape) >= min_blocksize and max(permute_at_level) > 0:
                out[i:i + cropshape[0], j:j + cropshape[1]] = hierarchical_block_permute(sub, blockshape, np.array(permute_at_level) - 1,
                                                                                         min_blocksize=min_blocksize, twist=twist)
            elif max(permute_at_level) > 0:
                raise ValueError('Recursive blockshape=%s < minimum blockshape=%d' % (sub.shape[0:2], min_blocksize))
    return out


def hierarchical_block_permutation_matrix(imgshape, blockshape, permute_at_level, min_blocksize=8, seed=None, twist=False, withinverse=False, strict=True):
    """The permutation matrix of hierarchical_block_permute acting on the HxWxC-flattened image (keynet/blockpermute.py:71-79)."""
    if seed is not None:
        np.random.seed(seed)
    n = int(np.prod(imgshape))
    cols = hierarchical_block_permute(np.arange(n).reshape(imgshape), blockshape, permute_at_level, min_blocksize, twist=twist, strict=strict).flatten()
    P = scipy.sparse.coo_matrix((np.ones(n, dtype=np.int64), (np.arange(n), cols)), shape=(n, n), dtype=np.float32)
    return P if not withinverse else (P, P.transpose())
