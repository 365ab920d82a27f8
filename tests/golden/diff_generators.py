#!/usr/bin/env python3
"""Differential check of the host key generators against the REFERENCE itself (build container only; run by
tests/test_keygen_families.py::test_generators_against_reference_randomised when /root/reference is mounted).

For randomised arguments, each generator of keynet_amd.keys / keynet_amd.sparse is called under the same numpy seed as its
reference counterpart and must return the same matrix: identical dense values, and identical stored CSR triplets where the
stored order feeds later products.  Prints one line per family and 'ALL OK'."""
import os
import sys
import warnings
import numpy as np
import scipy.sparse

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _refimport  # noqa: E402

ref = _refimport.import_reference()
import keynet.blockpermute  # noqa: E402
from keynet_amd import keys as kk  # noqa: E402
from keynet_amd import sparse as ks  # noqa: E402


def same(A, B, what, stored=True):
    (A, B) = (scipy.sparse.csr_matrix(A) if not scipy.sparse.issparse(A) else A.tocsr(), scipy.sparse.csr_matrix(B) if not scipy.sparse.issparse(B) else B.tocsr())
    assert A.shape == B.shape and A.dtype == B.dtype, (what, A.shape, B.shape, A.dtype, B.dtype)
    if stored:
        assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices) and np.array_equal(A.data, B.data), what
    else:
        assert np.array_equal(np.asarray(A.todense()), np.asarray(B.todense())), what


def both(seed, f_ref, f_own):
    np.random.seed(seed)
    r = f_ref()
    np.random.seed(seed)
    o = f_own()
    assert np.random.rand() == (np.random.seed(seed), f_ref(), np.random.rand())[2], 'RNG consumption differs'   # same number of draws
    return (r, o)


def main():
    rng = np.random.RandomState(123)
    warnings.simplefilter('ignore')
    for t in range(12):
        (n, k) = (int(rng.randint(2, 40)), int(rng.randint(1, 60)))
        ((R, Ri), (O, Oi)) = both(t, lambda: ref.sparse.sparse_orthogonal_matrix(n, k, withinverse=True), lambda: kk.givens_orthogonal(n, k, withinverse=True))
        same(R, O, 'givens n=%d k=%d' % (n, k)); same(Ri, Oi, 'givens inverse')
    print('givens_orthogonal ok')
    for t in range(10):
        (n, k) = (int(rng.randint(6, 48)), int(rng.randint(2, 7)))
        ((R, Ri), (O, Oi)) = both(100 + t, lambda: ref.sparse.sparse_random_diagonally_dominant_doubly_stochastic_matrix(n, k, withinverse=True),
                                  lambda: kk.diagonally_dominant_doubly_stochastic(n, k, withinverse=True))
        same(R, O, 'doubly stochastic n=%d k=%d' % (n, k)); same(Ri, Oi, 'doubly stochastic inverse')
    print('diagonally_dominant_doubly_stochastic ok')
    for t in range(12):
        (H, C) = (int(rng.choice([16, 32, 64])), int(rng.randint(1, 4)))
        levels = [list(range(0, d + 1)) for d in range(0, int(np.log2(H)) - 3 + 1)][int(rng.randint(0, int(np.log2(H)) - 2))]
        twist = bool(rng.randint(0, 2))
        img = rng.rand(H, H, C).astype(np.float32)
        (R, O) = both(200 + t, lambda: keynet.blockpermute.hierarchical_block_permute(img, (2, 2), levels, min_blocksize=8, twist=twist),
                      lambda: kk.hierarchical_block_permute(img, (2, 2), levels, min_blocksize=8, twist=twist))
        assert np.array_equal(R, O), ('hierarchical_block_permute', H, C, levels, twist)
        ((R, Ri), (O, Oi)) = both(300 + t, lambda: keynet.blockpermute.hierarchical_block_permutation_matrix((H, H, C), (2, 2), levels, min_blocksize=8, twist=twist, withinverse=True),
                                  lambda: kk.hierarchical_block_permutation_matrix((H, H, C), (2, 2), levels, min_blocksize=8, twist=twist, withinverse=True))
        same(R, O, 'hierarchical matrix'); same(Ri, Oi, 'hierarchical matrix inverse')
    (R, O) = both(7, lambda: keynet.blockpermute.hierarchical_block_permute(rng.rand(24, 40, 2), (3, 5), [0]), lambda: kk.hierarchical_block_permute(rng.rand(24, 40, 2), (3, 5), [0]))
    print('hierarchical_block_permute / _matrix ok')
    for t in range(10):
        n = int(rng.randint(1, 30))
        d = scipy.sparse.diags(rng.rand(n) + 0.5)
        bias = rng.randn(n, 1) * (rng.rand(n, 1) > 0.3) if t % 3 else None
        for fmt in ('dia', 'csr', 'coo'):
            A = d.asformat(fmt)
            ((R, Ri), (O, Oi)) = (ref.sparse.diagonal_affine_to_linear(A, bias, withinverse=True), kk.diagonal_affine_to_linear(A, bias, withinverse=True))
            same(R, O, 'diagonal_affine_to_linear L'); same(Ri, Oi, 'diagonal_affine_to_linear Linv')
            same(ref.sparse.sparse_affine_to_linear(A, bias), ks.sparse_affine_to_linear(A, bias), 'sparse_affine_to_linear')
    print('diagonal_affine_to_linear / sparse_affine_to_linear ok')
    for t in range(6):
        n = int(rng.randint(1, 50))
        ((R, Ri), (O, Oi)) = both(400 + t, lambda: ref.sparse.sparse_permutation_matrix(n, withinverse=True), lambda: ks.sparse_permutation_matrix(n, withinverse=True))
        same(R, O, 'permutation'); same(Ri, Oi, 'permutation inverse')
        (h, H) = (int(rng.randint(1, 6)), int(rng.randint(3, 20)))
        B = rng.rand(h, h).astype(np.float32)
        (R, O) = (ref.sparse.DiagonalTiledMatrix(B, shape=(H, H)), ks.DiagonalTiledMatrix(B, shape=(H, H)))
        assert list(R) == list(O), 'DiagonalTiledMatrix blocks'
        same(R.tocsr(), O.tocsr(), 'DiagonalTiledMatrix expansion')
        Bs = scipy.sparse.random(h, h, density=0.6, format='csr', dtype=np.float32, random_state=t)
        (R, O) = (ref.sparse.DiagonalTiledMatrix(Bs, shape=(H, H + 1)), ks.DiagonalTiledMatrix(Bs, shape=(H, H + 1)))
        assert list(R) == list(O)
        same(R.tocsr(), O.tocsr(), 'DiagonalTiledMatrix sparse block')
    print('sparse_permutation_matrix / DiagonalTiledMatrix ok')
    print('ALL OK')


if __name__ == '__main__':
    os.chdir('/tmp')
    main()
