"""Small host helpers whose semantics the keyed forward depends on."""
import numpy as np


def find_closest_positive_divisor(a, b):
    """Divisor d > 1 of `a` nearest to `b` (ties: the larger one is tried first); a itself when a <= b.
    Decides the EFFECTIVE tile size per layer: "tile 64" on 224x224 becomes 56 (keynet/util.py:16-28, used by
    keynet/system.py:304-309)."""
    assert a > 0 and b > 0
    if a <= b:
        return a
    for delta in range(0, a - b + 1):
        for cand in (b + delta, b - delta):
            if cand > 1 and a % cand == 0:
                return cand
    return a


def blockview(A, n):
    """[H,W] -> [H//n, W//n, n, n] strided view with blockview(A,n)[i,j] == A[i*n:(i+1)*n, j*n:(j+1)*n] (keynet/util.py:40-45)."""
    assert A.ndim == 2
    return np.lib.stride_tricks.as_strided(A, shape=(A.shape[0] // n, A.shape[1] // n, n, n),
                                           strides=(n * A.strides[0], n * A.strides[1]) + A.strides)
