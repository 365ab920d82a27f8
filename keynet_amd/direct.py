"""Direct-to-tiled keying: build the keyed operator of a conv / avgpool layer WITHOUT materialising its Toeplitz matrix.

The reference keys a conv layer as scipy SpGEMMs on the explicit (Cout*HoWo+1) x (Cin*HiWi+1) Toeplitz matrix and then
tiles the result (keynet/layer.py:24-41, keynet/sparse.py:720-776).  For VGG-16 that is 15 G non-zeros (120 GB; the
Toeplitz builder alone preallocates 22 GB for conv1_2), so the reference route cannot build BASELINE configs 4-5.

Every tile-compressible key family of the reference acts on the pixels only and is replicated over channels:
    A = [[I_C (x) a, beta], [0, 1]]              (keynet/system.py:390-410: DiagonalTiledMatrix channel repeat)
and a conv layer is  W = [[sum_t F_t (x) S_t, b (x) 1], [0, 1]]  with F_t the Cout x Cin matrix of filter tap t and S_t
the HoWo x HiWi 0/1 shift matrix of that tap (stride and zero padding included).  Therefore
    A_out W A_in^-1 = [[ sum_t F_t (x) (a_out S_t a_in^-1),  M_out (Wc gamma + b) + beta_out ], [0, 1]]
with gamma the last column of A_in^-1.  K_t = a_out S_t a_in^-1 is a SPATIAL sparse product (HoWo x HiWi, ~9 nnz/row):
its entries (o, i, coef) together with the tap matrices are exactly the factored form libkeynet_hip consumes
(kn_convtaps_create) -- the same operator Conv2dTiledMatrix stores after de-duplicating its channel matrices.
tests/test_direct_keying.py checks it against the reference route entry by entry on the golden mini-nets.
"""
import numpy as np
import scipy.sparse
import torch
import torch.nn.functional as F

from . import sparse as ksp


def _perturb(w):
    """fl32(fl32(w + off) - off), off = |min w| + 1: the value the reference's Toeplitz builder leaves (sparse.py:184-187)."""
    w = np.asarray(w, dtype=np.float32)
    off = np.float32(np.abs(np.min(w)) + np.float32(1.0))
    return ((w + off) - off).astype(np.float32)


def spatial_key(A, channels, numel):
    """Split a channel-replicated key A = [[I_C (x) a, beta],[0,1]] into (a [numel x numel] csr, beta [C*numel]).
    Raises ValueError when A is not of that form (e.g. a global permutation: not tile compressible)."""
    A = A.tocsr()
    N = channels * numel
    assert A.shape == (N + 1, N + 1)
    a = A[0:numel, 0:numel].tocsr()
    beta = np.asarray(A[0:N, N].todense()).ravel().astype(np.float32)
    nb = int(np.count_nonzero(beta)) if A[0:N, N].nnz else 0
    core_nnz = A.nnz - A[0:N, N].nnz - A[N, :].nnz
    if core_nnz != channels * a.nnz:
        raise ValueError('key is not channel-replicated (I_C (x) a): direct keying does not apply')
    if channels > 1:
        a1 = A[numel:2 * numel, numel:2 * numel].tocsr()
        if (a1 != a).nnz != 0:
            raise ValueError('key differs between channels: direct keying does not apply')
    if A[N, 0:N].nnz != 0 or A[N, N] != 1:
        raise ValueError('key is not homogeneous ([..., 0 ... 0 1] last row)')
    return (a, beta)


def shift_matrices(inhw, ksize, stride):
    """[(tap index (i,j), S_t csr [HoWo x HiWi])] for a 'same'-padded k x k window (keynet/sparse.py:128-158 geometry)."""
    (U, V) = inhw
    (Us, Vs) = (U // stride, V // stride)
    (ku, kv) = np.meshgrid(np.arange(Us), np.arange(Vs), indexing='ij')
    (u, v) = (ku * stride, kv * stride)
    out = []
    r = (ksize - 1) // 2
    for (i, p) in enumerate(range(-r, r + 1)):
        for (j, q) in enumerate(range(-r, r + 1)):
            ok = ((u + p) >= 0) & ((u + p) < U) & ((v + q) >= 0) & ((v + q) < V)
            (o, ii) = ((ku * Vs + kv)[ok], ((u + p) * V + (v + q))[ok])
            out.append(((i, j), scipy.sparse.csr_matrix((np.ones(len(o), dtype=np.float32), (o, ii)), shape=(Us * Vs, U * V))))
    return out


def keyed_conv_taps(weight, bias, inshape, outshape, stride, A, Ainv):
    """Factored keyed conv operator.  Returns kwargs for Conv2dTiledMatrix.fromtaps."""
    (Cin, Hin, Win) = inshape
    (Cout, Hout, Wout) = outshape
    (HoWo, HiWi) = (Hout * Wout, Hin * Win)
    weight = _perturb(weight)
    ksize = weight.shape[2]
    (a_out, beta_out) = spatial_key(A, Cout, HoWo) if A is not None else (scipy.sparse.eye(HoWo, dtype=np.float32, format='csr'), np.zeros(Cout * HoWo, np.float32))
    (a_in, gamma) = spatial_key(Ainv, Cin, HiWi)
    (taps, eo, ei, et, ec) = ([], [], [], [], [])
    for (t, ((i, j), S)) in enumerate(shift_matrices((Hin, Win), ksize, stride)):
        K = a_out.dot(S).dot(a_in).tocoo()
        taps.append(weight[:, :, i, j])
        eo.append(K.row.astype(np.int32))
        ei.append(K.col.astype(np.int32))
        et.append(np.full(K.nnz, t, dtype=np.int32))
        ec.append(K.data.astype(np.float32))
    # last column: M_out (Wc gamma + b) + beta_out, then the homogeneous 1
    bvec = np.repeat(_perturb(bias) if bias is not None else np.zeros(Cout, np.float32), HoWo).astype(np.float32)
    if np.any(gamma != 0):
        g = torch.as_tensor(gamma.reshape(1, Cin, Hin, Win))
        with torch.no_grad():
            wg = F.conv2d(g, torch.as_tensor(weight), bias=None, stride=stride, padding=(ksize - 1) // 2).numpy().reshape(-1)
        bvec = bvec + wg
    col = np.concatenate([a_out.dot(bvec[c * HoWo:(c + 1) * HoWo]) for c in range(Cout)]).astype(np.float32) + beta_out
    lastcol = np.concatenate((col, np.ones(1, np.float32))).astype(np.float32)
    ent_coef = np.concatenate(ec)
    return dict(inshape=inshape, outshape=outshape, taps=np.stack(taps), ent_out=np.concatenate(eo), ent_in=np.concatenate(ei), ent_tap=np.concatenate(et),
                ent_coef=None if np.all(ent_coef == 1.0) else ent_coef, lastcol=lastcol)


def keyed_avgpool_csr(channels, inhw, ksize, stride, A, Ainv):
    """Keyed average pooling as one CSR:  I_C (x) (a_out S_pool a_in^-1) + last column + homogeneous row.
    S_pool = sum_t (1/k^2) S_t with the reference's value perturbation (the Toeplitz filter of keynet/sparse.py:206-212
    holds explicit zeros for the cross-channel pairs, so off = 1 when C > 1)."""
    (Hin, Win) = inhw
    (Hout, Wout) = (Hin // stride, Win // stride)
    (HoWo, HiWi) = (Hout * Wout, Hin * Win)
    vals = np.full(ksize * ksize + (1 if channels > 1 else 0), 1.0 / (ksize * ksize), dtype=np.float32)
    if channels > 1:
        vals[-1] = 0.0
    w = _perturb(vals)[0]
    S = None
    for (_, St) in shift_matrices((Hin, Win), ksize, stride):
        S = St if S is None else S + St
    S = (S * np.float32(w)).astype(np.float32).tocsr()
    (a_out, beta_out) = spatial_key(A, channels, HoWo) if A is not None else (scipy.sparse.eye(HoWo, dtype=np.float32, format='csr'), np.zeros(channels * HoWo, np.float32))
    (a_in, gamma) = spatial_key(Ainv, channels, HiWi)
    K = a_out.dot(S).dot(a_in).tocsr()
    K.eliminate_zeros()
    # last column: M_out (Wc gamma + 0) + beta_out ; exact zeros are not stored (scipy SpGEMM drops them)
    col = np.concatenate([a_out.dot(S.dot(gamma[c * HiWi:(c + 1) * HiWi])) for c in range(channels)]).astype(np.float32) + beta_out
    (rows_n, cols_n) = (channels * HoWo + 1, channels * HiWi + 1)
    per_row = np.diff(K.indptr)
    nzc = col != 0
    indptr = np.zeros(rows_n + 1, dtype=np.int64)
    counts = np.tile(per_row, channels) + nzc.astype(np.int64)
    indptr[1:rows_n] = np.cumsum(counts)
    indptr[rows_n] = indptr[rows_n - 1] + 1
    indices = np.empty(indptr[-1], dtype=np.int32)
    data = np.empty(indptr[-1], dtype=np.float32)
    # fill: spatial entries first (stored order of the SpGEMM result), then the bias entry, per row
    base = indptr[:-2].reshape(channels, HoWo)
    for c in range(channels):
        start = base[c]
        pos = (np.repeat(start, per_row) + (np.arange(K.nnz) - np.repeat(K.indptr[:-1], per_row))).astype(np.int64)
        indices[pos] = K.indices + c * HiWi
        data[pos] = K.data
        nz = np.flatnonzero(nzc[c * HoWo:(c + 1) * HoWo])
        if len(nz):
            bp = start[nz] + per_row[nz]
            indices[bp] = cols_n - 1
            data[bp] = col[c * HoWo + nz]
    indices[-1] = cols_n - 1
    data[-1] = 1.0
    return scipy.sparse.csr_matrix((data, indices, indptr.astype(np.int32)), shape=(rows_n, cols_n))


def toeplitz_entries(module_kind, inshape, outshape, ksize):
    """How many entries the reference's Toeplitz builder would emit (decides when the direct route is taken)."""
    (Cin, _, _) = inshape
    (Cout, Hout, Wout) = outshape
    return Hout * Wout * ksize * ksize * Cin * Cout
