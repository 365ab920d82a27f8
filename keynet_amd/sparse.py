"""Operator containers of the keyed forward, HIP-backed: the objects stored in KeyedLayer.W.

Mirror of the reference's operator interface (keynet/sparse.py:419-835): SparseMatrix, TiledMatrix, DiagonalTiledMatrix,
Conv2dTiledMatrix with torchdot / dot / nnz / shape / tocoo / tocsr / transpose / clone.  The host keeps the defining
arrays (scipy matrix or blocks+tiles) for the structural methods and pickling; `torchdot` -- the hot path -- always runs
in libkeynet_hip.so on an MI355X (there is no CPU route: a missing library or device raises).

Build-time (host, offline) constructors live here too, restated and vectorised: Toeplitz matrices of conv / avgpool
layers (keynet/sparse.py:122-212) and the tilers (keynet/sparse.py:519-571, 692-776).
"""
import copy
import numpy as np
import scipy.sparse
import torch

from . import _capi


# ------------------------------------------------------------------------------------------------------------------
# helpers
def is_scipy_sparse(A):
    return scipy.sparse.issparse(A)


def _stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def _compute_dtype(dt):
    """The dtype scipy's product of an operator of dtype `dt` with float32 activations runs and returns in (keynet/sparse.py:488-492:
    `x_torch.type(torch.FloatTensor)` against `self._matrix` as it is; numpy up-casts the pair)."""
    return np.dtype(np.float64) if np.result_type(dt, np.float32) == np.float64 else np.dtype(np.float32)


def _stored_order_csr(M):
    """(indptr, indices, data) whose per-row order equals the accumulation order of scipy's matvec for M's format:
    CSR as stored (csr_matvecs); COO in entry order (coo_matmat_dense); CSC column-major (csc_matvecs).  Nothing is
    sorted or summed -- keyed matrices are non-canonical and the order is part of the result.  The values keep the dtype scipy
    computes in against float32 activations (numpy's up-cast rule: float32 stays float32, float64 -- and integer types -- compute in float64)."""
    vdt = _compute_dtype(M.dtype)
    if M.format == 'csr':
        return (np.asarray(M.indptr, dtype=np.int32), np.asarray(M.indices, dtype=np.int32), np.asarray(M.data, dtype=vdt))
    if M.format == 'csc':
        cols = np.repeat(np.arange(M.shape[1], dtype=np.int64), np.diff(M.indptr))
        (rows, vals) = (np.asarray(M.indices, dtype=np.int64), M.data)
    else:
        C = M if M.format == 'coo' else M.tocoo()
        (rows, cols, vals) = (np.asarray(C.row, dtype=np.int64), np.asarray(C.col, dtype=np.int64), C.data)
    order = np.argsort(rows, kind='stable')
    indptr = np.zeros(M.shape[0] + 1, dtype=np.int64)
    np.add.at(indptr, rows + 1, 1)
    return (np.cumsum(indptr).astype(np.int32), cols[order].astype(np.int32), np.asarray(vals, dtype=vdt)[order])


def _on_device(W, attr, make, device=None):
    """Per-device cache of a container's operator handle: a kn_handle_t lives in ONE GPU's HBM, so it is created under the
    device of the activations it will multiply (not whatever device happens to be current) and looked up by device index.
    `W.<attr> = None` anywhere else drops every cached handle."""
    idx = torch.cuda.current_device() if (device is None or device.index is None) else device.index
    cache = getattr(W, attr, None)
    if not isinstance(cache, dict):
        cache = {}
        setattr(W, attr, cache)
    if idx not in cache:
        with torch.cuda.device(idx):
            cache[idx] = make()
    return cache[idx]


def _run_torchdot(get_op, shape, x, relu=False, exact=True, extra_flags=0, absmax=None, f64=False):
    """Y = W.X on the GPU.  x: torch tensor [cols, N] (any device / strides); get_op(device) -> the operator handle resident
    on that device.  Returns [rows, N] on x's device.  `f64`: the operator is float64 -- the result is the float64 block scipy returns for it
    (kn_spmm_f64); the activations are float32 either way (keynet/sparse.py:489-491)."""
    assert shape[1] == x.shape[0], 'Non-conformal shape for W=%s, x=%s' % (str(shape), str(tuple(x.shape)))
    if not torch.cuda.is_available():
        raise _capi.KeynetHipError('keynet_amd: no MI355X visible -- the keyed forward has no CPU fallback')
    src_device = x.device
    xd = x.detach()
    if xd.dtype != torch.float32:
        xd = xd.float()          # the reference coerces to FloatTensor silently (keynet/sparse.py:489-491)
    if not xd.is_cuda:
        xd = xd.cuda()
    if not xd.is_contiguous():
        xd = xd.contiguous()
    n = xd.shape[1]
    y = torch.empty((shape[0], n), dtype=torch.float64 if f64 else torch.float32, device=xd.device)
    flags = (_capi.KN_FLAG_RELU if relu else 0) | (_capi.KN_FLAG_EXACT if exact else 0) | int(extra_flags)
    with torch.cuda.device(xd.device):
        if f64:
            get_op(xd.device).spmm_f64(xd.data_ptr(), n, n, y.data_ptr(), n, flags, _stream_ptr())
            if absmax is not None:               # (a float64 layer is never on a re-ordering kernel; the slot still feeds the NEXT layer's screen)
                _capi.absmax(y.float().data_ptr(), shape[0], n, n, absmax.data_ptr(), _stream_ptr())
        else:
            get_op(xd.device).spmm(xd.data_ptr(), n, n, y.data_ptr(), n, flags, _stream_ptr(), absmax_ptr=None if absmax is None else absmax.data_ptr())
    return y if src_device.type == 'cuda' else y.to(src_device)


# ------------------------------------------------------------------------------------------------------------------
class SparseMatrix(object):
    """scipy-sparse (or dense ndarray) operator applied by order-preserving CSR SpMM kernels (keynet/sparse.py:419-514)."""

    def __init__(self, A=None):
        assert A is None or is_scipy_sparse(A) or isinstance(A, np.ndarray), 'Invalid input - %s' % (str(type(A)))
        self.shape = A.shape if A is not None else (0, 0)
        self._matrix = A
        self.dtype = A.dtype if A is not None else None
        self.ndim = 2
        self._op = None
        self._op_dense = None

    def __repr__(self):
        return str('<keynet_amd.SparseMatrix: H=%d, W=%d, backend=hip>' % (self.shape[0], self.shape[1]))

    def __getstate__(self):
        d = dict(self.__dict__)
        d['_op'] = None   # device handles do not pickle; rebuilt lazily
        d['_op_dense'] = None
        if '_op_split' in d:
            d['_op_split'] = None
        return d

    def __add__(self, other):
        assert isinstance(other, SparseMatrix) and self.shape == other.shape, 'Invalid input'
        self._matrix = self._matrix + other._matrix
        self._op = None
        return self

    # -- device side
    def _device_op(self, device=None):
        def make():
            M = self._matrix
            if isinstance(M, np.ndarray):
                M = scipy.sparse.csr_matrix((np.asarray(M, dtype=np.float32).ravel(), np.tile(np.arange(M.shape[1]), M.shape[0]),
                                             np.arange(0, M.size + 1, M.shape[1])), shape=M.shape)
            (ip, ix, dt) = _stored_order_csr(M)
            return _capi.Operator.csr(self.shape, ip, ix, dt)
        return _on_device(self, '_op', make, device)

    def is_float64(self):
        """Does scipy compute this operator's product with float32 activations in float64 (a float64 -- or integer -- scipy matrix)?
        Dense ndarray operators go through BLAS in the reference (no defined order) and stay float32 here."""
        M = getattr(self, '_matrix', None)
        return M is not None and is_scipy_sparse(M) and _compute_dtype(M.dtype) == np.float64

    DENSE_MIN_ELEMENTS = 1 << 20

    def _dense_device_op(self, device=None):
        """kn_dense_create handle for a (nearly) dense operator such as a keyed nn.Linear, or None when not eligible."""
        if getattr(self, '_matrix', None) is None:
            return None          # tiled containers have no single host matrix: never dense-eligible

        def make():
            M = self._matrix
            (r, c) = self.shape
            ok = (M is not None and not self.is_float64() and r * c >= self.DENSE_MIN_ELEMENTS and (c - 1) % 256 == 0 and
                  (M.nnz if is_scipy_sparse(M) else M.size) >= 0.5 * r * c)
            if ok:
                D = np.asarray(M.todense() if is_scipy_sparse(M) else M, dtype=np.float32)
                if bool(np.all(D[-1, :-1] == 0)):
                    return _capi.Operator.dense(D)
            return False
        return _on_device(self, '_op_dense', make, device) or None

    def torchdot(self, x_torch, relu=False, exact=True, absmax=None):
        """W . x for x of shape [W.shape[1], N]: the hot path (keynet/sparse.py:488-492).  exact=True (default): bit-exact
        with scipy (order-preserving CSR kernels).  exact=False: a large dense operator (keyed nn.Linear) may run as a
        split-K f32-MFMA GEMM instead (within 1e-5; used by the tiled key-nets whose conv layers are on MFMA anyway).
        `absmax`: one-element device f32 tensor raised to max |W . x| (kn_spmm_screen)."""
        if exact == 'bf16x3':
            exact = False            # a dense operator has no bf16x3 path (yet): f32 matrix cores
        if self.is_float64():
            # a float64 operator computes in float64 and returns float64, as scipy does for it (the next layer's coercion rounds to f32 once)
            return _run_torchdot(self._device_op, self.shape, x_torch, relu=relu, exact=True, absmax=absmax, f64=True)
        if not exact and torch.cuda.is_available() and self._dense_device_op(x_torch.device if x_torch.is_cuda else None) is not None:
            return _run_torchdot(self._dense_device_op, self.shape, x_torch, relu=relu, exact=False, absmax=absmax)
        return _run_torchdot(self._device_op, self.shape, x_torch, relu=relu, exact=True, absmax=absmax)

    def dot(self, x_numpy):
        assert isinstance(x_numpy, np.ndarray)
        return self.torchdot(torch.as_tensor(np.asarray(x_numpy))).cpu().numpy()

    def max_abs_rowsum(self):
        """max_i sum_j |W_ij| (host, float64): the operator-side factor of the float-key error bound (KeyedLayer._calibrate)."""
        M = self._matrix
        if is_scipy_sparse(M):
            return float(abs(M).sum(axis=1).max()) if M.nnz else 0.0
        return float(np.abs(np.asarray(M, dtype=np.float64)).sum(axis=1).max()) if M.size else 0.0

    # -- structure (host)
    def new(self):
        return SparseMatrix()

    def clone(self):
        return copy.deepcopy(self)

    def from_torch_dense(self, A):
        return SparseMatrix(A.detach().numpy())

    def from_scipy_sparse(self, A):
        assert is_scipy_sparse(A)
        return SparseMatrix(A)

    def matmul(self, A):
        """In-place sparse x sparse product on the host (keynet/sparse.py:472-480); not on the forward path."""
        assert isinstance(A, SparseMatrix) or is_scipy_sparse(A)
        self._matrix = self._matrix.dot(A._matrix if isinstance(A, SparseMatrix) else A)
        self.shape = self._matrix.shape
        (self._op, self._op_dense) = (None, None)
        return self

    def nnz(self):
        return self._matrix.nnz if is_scipy_sparse(self._matrix) else self._matrix.size

    def transpose(self):
        self._matrix = self._matrix.transpose()
        self.shape = self._matrix.shape
        (self._op, self._op_dense) = (None, None)
        return self

    def tocoo(self):
        return self._matrix.tocoo() if is_scipy_sparse(self._matrix) else scipy.sparse.coo_matrix(self._matrix)

    def tocsr(self):
        self._matrix = self._matrix.tocsr()
        self._op = None
        return self

    def tocsc(self):
        self._matrix = self._matrix.tocsc()
        self._op = None
        return self

    def from_torch_conv2d(self, inshape, w, b, stride):
        return SparseMatrix(sparse_toeplitz_conv2d(inshape, w.detach().numpy(), bias=b.detach().numpy(), stride=stride))


# ------------------------------------------------------------------------------------------------------------------
class FactoredSparseMatrix(SparseMatrix):
    """An UNTILED keyed conv layer (keynet/layer.py:24-41: a scipy CSR, `SparseMatrix`) whose stored CSR has been PROVEN, on the host, to be the
    ascending-column expansion of a factored conv operator (taps x spatial entries, keynet_amd/direct.py) with the exact zeros dropped.  That is
    what the keying SpGEMMs leave for identity / channel-replicated permutation keys on both sides -- every conv layer of PermutationKeynet
    AllConvNet but the first.  scipy's product in stored order is then, entry for entry, the factored operator's order-preserving product
    (KN_FLAG_EXACT: channel outer, the pixel's slots by ascending input pixel inner, bias last), so the device holds 0.3 MB of taps + the slot lists + a
    stored-column table instead of the CSR's hundreds of MB and runs the matrix-pipe grouped kernel from that table (narrow batches: the conv pipeline);
    kn_convtaps_drop_zero_entries covers the dropped zeros.  Everything on the
    host (`_matrix`, nnz(), tocoo(), pickling, the neutral file format) is the plain CSR container's."""

    def __init__(self, A, factored):
        super(FactoredSparseMatrix, self).__init__(A)
        self._factored = factored

    def __repr__(self):
        return str('<keynet_amd.SparseMatrix: H=%d, W=%d, backend=hip (factored on the device)>' % (self.shape[0], self.shape[1]))

    def _device_op(self, device=None):
        def make():
            t = self._factored._taps
            return _capi.Operator.convtaps(self._factored._inshape, self._factored._outshape, t['taps'], t['ent_out'], t['ent_in'], t['ent_tap'], t['ent_coef'],
                                           t['lastcol']).drop_zero_entries()
        return _on_device(self, '_op', make, device)

    def _dense_device_op(self, device=None):
        return None

    def torchdot(self, x_torch, relu=False, exact=True, absmax=None):
        return _run_torchdot(self._device_op, self.shape, x_torch, relu=relu, exact=True, absmax=absmax)       # always the reference's order

    @staticmethod
    def proven(M, factored, max_zero_fraction=0.01):
        """Is the stored CSR `M` (scipy) exactly -- indptr, indices AND values, in stored order -- the canonical expansion of `factored` with its
        zero-valued entries removed?  Also refuses operators with many exact zeros (a pruned filter): the device re-checks every such entry."""
        try:
            return FactoredSparseMatrix._proven(M, factored, max_zero_fraction)
        except Exception:                # any surprise in a crafted / damaged description: the plain CSR container it is (the comments promise a fallback)
            return False

    @staticmethod
    def _proven(M, factored, max_zero_fraction):
        t = factored._taps
        if t is None or M.format != 'csr' or tuple(M.shape) != tuple(factored.shape) or M.dtype != np.float32:
            return False
        taps = t['taps']
        live = taps[np.any(taps.reshape(taps.shape[0], -1) != 0, axis=1)]
        if live.size and float(np.count_nonzero(live == 0)) > max_zero_fraction * live.size:
            return False
        E = factored.rows_csr()
        keep = E.data != 0
        if M.nnz != int(np.count_nonzero(keep)):
            return False
        counts = np.diff(np.concatenate(([0], np.cumsum(keep, dtype=np.int64)))[E.indptr.astype(np.int64)])      # kept entries per row (empty rows, trailing ones included: 0)
        (ip, ix, dt) = _stored_order_csr(M)
        return bool(np.array_equal(np.concatenate(([0], np.cumsum(counts))).astype(np.int64), ip.astype(np.int64)) and
                    np.array_equal(E.indices[keep].astype(np.int32), ix) and np.array_equal(E.data[keep].view(np.uint32), dt.view(np.uint32)))


def _structure_preserving(vals64):
    """The reference stores tile values as ((v + off) - off) with off = |min v| + 1 computed in float64, then casts to
    float32 (keynet/sparse.py:562-566, 582): reproduced so tile contents match bit for bit."""
    off = np.abs(np.min(vals64)) + 1.0
    return ((vals64 + off) - off).astype(np.float32)


class TiledMatrix(SparseMatrix):
    """Sparse matrix stored as de-duplicated h x w tiles + a block list (keynet/sparse.py:517-653).

    The tile dictionary is storage compression; on the device the operator is expanded ONCE to canonical CSR (the
    reference re-expands it on every torchdot call, keynet/sparse.py:610) because per non-zero the 8 bytes of (col,val)
    are negligible next to the n_vecs*4-byte activation row it gathers."""

    def __init__(self, T, tileshape):
        assert is_scipy_sparse(T), 'input must be a scipy sparse matrix'
        assert isinstance(tileshape, tuple) and len(tileshape) == 2 and tileshape[0] > 0 and tileshape[1] > 0, 'tileshape must be tuple (tileheight, tilewidth) > 0'
        self._tileshape = tileshape
        self.dtype = T.dtype
        self.shape = (T.shape[0], T.shape[1])
        self.ndim = 2
        self._op = None
        (self._blocks, self._tiles) = _tile_sparse(T, tileshape)

    def __repr__(self):
        return str('<keynet_amd.TiledMatrix: H=%d, W=%d, tileshape=%s, tiles=%d>' % (self.shape[0], self.shape[1], str(self.tileshape()), len(self.tiles())))

    def __iter__(self):
        for (i, j, k) in self._blocks:
            yield (i, j, k)

    def tileshape(self):
        return self._tileshape

    def tiles(self):
        return self._tiles

    def blocks(self):
        return list(self.__iter__())

    def _tile_arrays(self):
        tiles = [t.tocoo() for t in self._tiles]
        ptr = np.cumsum([0] + [t.nnz for t in tiles]).astype(np.int64)
        cat = (lambda L, dt: np.concatenate(L).astype(dt) if len(L) else np.zeros(0, dt))
        return (ptr, cat([t.row for t in tiles], np.int32), cat([t.col for t in tiles], np.int32), cat([t.data for t in tiles], np.float32))

    def _device_op(self, device=None):
        def make():
            (ptr, tr, tc, tv) = self._tile_arrays()
            return _capi.Operator.tiled(self.shape, np.array(list(self), dtype=np.int64).reshape(-1, 3), ptr, tr, tc, tv)
        return _on_device(self, '_op', make, device)

    def torchdot(self, x, relu=False, exact=True, absmax=None):
        """[cols, N] -> [rows, N] (keynet/sparse.py:603-612); always the order-preserving path (bit-exact)."""
        if isinstance(x, np.ndarray):
            x = torch.as_tensor(x)
        return _run_torchdot(self._device_op, self.shape, x, relu=relu, exact=True, absmax=absmax)

    def dot(self, x):
        assert isinstance(x, np.ndarray)
        return self.torchdot(torch.as_tensor(x)).cpu().numpy()

    def max_abs_rowsum(self):
        M = self.tosparse('csr')
        return float(abs(M).sum(axis=1).max()) if M.nnz else 0.0

    def copy(self, blocks, tiles):
        (self._blocks, self._tiles, self._op) = (blocks, tiles, None)
        return self

    def transpose(self):
        self._blocks = [(j, i, k) for (i, j, k) in self._blocks] if self._blocks is not None else self._blocks
        self._tiles = [t.transpose() for t in self._tiles]
        self._tileshape = (self._tileshape[1], self._tileshape[0])
        self.shape = (self.shape[1], self.shape[0])
        self._op = None
        return self

    def tosparse(self, format='coo'):
        """Expanded operator as scipy csr/coo/csc (keynet/sparse.py:621-641); host side, small matrices / tests."""
        tiles = [t.tocoo() for t in self._tiles]
        (R, C, V) = ([], [], [])
        for (i, j, k) in self.__iter__():
            R.append(i + tiles[k].row.astype(np.int64))
            C.append(j + tiles[k].col.astype(np.int64))
            V.append(tiles[k].data)
        cat = (lambda L, dt: np.concatenate(L) if len(L) else np.zeros(0, dt))
        return _from_coo(format, cat(V, np.float32), cat(R, np.int64), cat(C, np.int64), self.shape)

    def tocsr(self):
        return self.tosparse(format='csr')

    def tocoo(self):
        return self.tosparse(format='coo')

    def nnz(self):
        return sum([t.nnz for t in self._tiles])


def _from_coo(format, vals, rows, cols, shape):
    if format == 'csr':
        return scipy.sparse.csr_matrix((vals, (rows, cols)), shape=shape)
    if format == 'coo':
        return scipy.sparse.coo_matrix((vals, (rows, cols)), shape=shape)
    if format == 'csc':
        return scipy.sparse.csc_matrix((vals, (rows, cols)), shape=shape)
    raise ValueError('Invalid format "%s" - must be ["coo", "csr", "csc"]' % format)


def _tile_sparse(T, tileshape):
    """Vectorised restatement of the TiledMatrix tiler (keynet/sparse.py:543-571).

    Returns (blocks, tiles): blocks = [(row0, col0, k)] row-major; tiles[k] = scipy COO (float32) of the k-th distinct
    tile.  Tile ids follow first occurrence in COO order; entries inside a tile are (row, col)-sorted (the reference's
    float64 -> float32 astype canonicalises the COO); two blocks share a tile iff their (i,j,v) sets and block shapes
    are identical (the reference hashes str(sorted(ijv))+str(shape))."""
    T = T.tocoo()
    (h, w) = tileshape
    (H, W) = T.shape
    if T.nnz == 0:
        return ([], [])
    (r, c) = (np.asarray(T.row, dtype=np.int64), np.asarray(T.col, dtype=np.int64))
    v = np.asarray(T.data, dtype=np.float64)
    (bi, bj) = (r // h, c // w)
    nbj = (W + w - 1) // w
    key = bi * nbj + bj
    order = np.argsort(key, kind='stable')
    ks = key[order]
    starts = np.flatnonzero(np.concatenate(([True], ks[1:] != ks[:-1])))
    ends = np.concatenate((starts[1:], [len(ks)]))
    first_seen = order[starts]                     # original COO index of each block's first entry
    tiles = []
    blocks = []
    seen = {}
    for b in np.argsort(first_seen, kind='stable'):
        idx = order[starts[b]:ends[b]]
        (kbi, kbj) = (int(ks[starts[b]] // nbj), int(ks[starts[b]] % nbj))
        (ii, jj, vv) = (r[idx] - kbi * h, c[idx] - kbj * w, v[idx])
        bshape = (h if (kbi * h + h) <= H else (H - kbi * h), w if (kbj * w + w) <= W else (W - kbj * w))
        s = np.lexsort((jj, ii))
        sig = (ii[s].tobytes(), jj[s].tobytes(), vv[s].tobytes(), bshape)
        k = seen.get(sig)
        if k is None:
            k = len(tiles)
            seen[sig] = k
            tiles.append(scipy.sparse.coo_matrix((_structure_preserving(vv)[s], (ii[s], jj[s])), shape=bshape))
        blocks.append((kbi * h, kbj * w, k))
    blocks.sort(key=lambda x: (x[0], x[1]))
    return (blocks, tiles)


class DiagonalTiledMatrix(TiledMatrix):
    """One block repeated down the main diagonal; where the block does not divide the matrix, the last (partial) diagonal
    position holds the matching corner of an identity instead (keynet/sparse.py:657-687)."""

    def __init__(self, B, shape):
        assert B.ndim == 2, 'Invalid block, must be 2D'
        assert isinstance(shape, tuple) and len(shape) == 2, 'invalid shape'
        (H, W) = shape
        if B.shape[0] > H or B.shape[1] > W:
            B = B.tocsr()[0:H, 0:W]                            # an oversized block is clipped to the matrix
        if not scipy.sparse.issparse(B):
            B = _dense_block_as_sparse(B)
        (h, w) = B.shape
        (self._tileshape, self.shape, self.dtype, self.ndim, self._op, self._blocks) = ((h, w), shape, B.dtype, 2, None, None)
        self._tiles = [B.astype(np.float32)]
        if H % h or W % w:
            self._tiles.append(scipy.sparse.eye(max(h, w)).tocsr()[0:H % h, 0:W % w].astype(np.float32))

    def __iter__(self):
        """(row0, col0, tile id) along the diagonal: tile 0 wherever the block fits strictly inside, else the last tile."""
        ((H, W), (h, w)) = (self.shape, self._tileshape)
        n = min(len(range(0, H, h)), len(range(0, W, w)))
        (i, j) = (np.arange(n) * h, np.arange(n) * w)
        k = np.where((i + h < H) & (j + w < W), 0, len(self._tiles) - 1)
        return iter([(int(a), int(b), int(c)) for (a, b, c) in zip(i, j, k)])


def _dense_block_as_sparse(B):
    """Dense block -> CSR keeping EVERY entry stored, zeros included (shift by |min| + 1 before the conversion, shift the stored
    values back afterwards): the tile then has the full h x w structure, as the reference's 'sparsity preserving' round trip."""
    off = np.abs(np.min(B)) + 1.0
    S = scipy.sparse.coo_matrix(B + off)
    S.data -= off
    return S.tocsr()


class Conv2dTiledMatrix(TiledMatrix):
    """Keyed conv operator as spatial blocks x dense channel matrices (keynet/sparse.py:690-835).

    _blocks = [(i, j, k)] over the channel-(0,0) plane (+ bias blocks at column Cin*Hin*Win);
    _tiles  = {(it, jt, k): float32[Cout, Cin]} (+ {(it, 0, k): [[b]]} for bias tiles).
    On the device the channel matrices are de-duplicated into `taps` and the forward is an implicit GEMM per output
    pixel on f32 MFMA (csrc/kn_conv.hip); `exact=True` in torchdot selects the order-preserving CSR path instead."""

    def __init__(self, T, inshape, outshape, tileshape, bias, sanitycheck=True):
        (Cin, Hin, Win) = inshape
        (Cout, Hout, Wout) = outshape
        self._inshape = inshape
        self._outshape = outshape
        self._tileshape = tileshape
        self.shape = T.shape
        self.dtype = T.dtype
        self.ndim = 2
        self._op = None
        self._taps = None
        assert tileshape[0] <= T.shape[0] and tileshape[1] <= T.shape[1]
        if bias:
            assert T.shape[0] == np.prod(outshape) + 1 and T.shape[1] == np.prod(inshape) + 1
            assert (self.shape[0] - 1) % tileshape[0] == 0 and (self.shape[1] - 1) % tileshape[1] == 0
        else:
            assert T.shape[0] == np.prod(outshape) and T.shape[1] == np.prod(inshape)
            assert self.shape[0] % tileshape[0] == 0 and self.shape[1] % tileshape[1] == 0
        T = T.tocsr()
        (HoWo, HiWi) = (Hout * Wout, Hin * Win)
        T_00 = T[0:HoWo, 0:HiWi]
        if sanitycheck and Cout > 1 and Cin > 1:
            T_10 = T[HoWo:2 * HoWo, 0:HiWi]
            T_01 = T[0:HoWo, HiWi:2 * HiWi]
            assert ((T_00 != 0) != (T_10 != 0)).nnz == 0 and ((T_00 != 0) != (T_01 != 0)).nnz == 0, 'channel-inconsistent sparsity'
        (blocks, _) = _tile_sparse(T_00, tileshape)
        T_lastcol = None
        if bias:
            T_lastcol = T[:, -1]
            T = T[0:-1, 0:-1]
        self._tiles = _conv_tiler(T.tocoo(), blocks, inshape, outshape, tileshape)
        self._blocks = list(blocks)
        if bias:
            (bblocks, btiles) = _tile_sparse(T_lastcol, (tileshape[0], 1))
            k_offset = len(self._tiles)
            for (kt, t) in enumerate(btiles):
                for (i, j, v) in zip(t.row, t.col, t.data):
                    self._tiles[(int(i), int(j), int(k_offset + kt))] = np.array(v).reshape(1, 1).astype(np.float32)
            self._blocks += [(i, Cin * HiWi, k_offset + k) for (i, j, k) in bblocks]
        self._blocks = sorted(self._blocks, key=lambda x: (x[0], x[1]))

    @classmethod
    def fromtaps(cls, inshape, outshape, taps, ent_out, ent_in, ent_tap, ent_coef=None, lastcol=None, tileshape=None):
        """Direct construction in factored form (never materialises the Toeplitz matrix): see kn_convtaps_create."""
        self = cls.__new__(cls)
        (self._inshape, self._outshape, self._tileshape) = (tuple(inshape), tuple(outshape), tileshape)
        has_last = lastcol is not None
        self.shape = (int(np.prod(outshape)) + (1 if has_last else 0), int(np.prod(inshape)) + (1 if has_last else 0))
        self.dtype = np.float32
        self.ndim = 2
        self._op = None
        (self._blocks, self._tiles) = (None, None)
        self._taps = dict(taps=np.ascontiguousarray(taps, dtype=np.float32), ent_out=np.asarray(ent_out, dtype=np.int32), ent_in=np.asarray(ent_in, dtype=np.int32),
                          ent_tap=np.asarray(ent_tap, dtype=np.int32), ent_coef=None if ent_coef is None else np.asarray(ent_coef, dtype=np.float32),
                          lastcol=None if lastcol is None else np.asarray(lastcol, dtype=np.float32))
        return self

    def __repr__(self):
        return str('<keynet_amd.Conv2dTiledMatrix: H=%d, W=%d, tileshape=%s, backend=hip>' % (self.shape[0], self.shape[1], str(self._tileshape)))

    def __iter__(self):
        assert self._blocks is not None, 'operator was built in factored form (fromtaps): no block list'
        for (i, j, k) in self._blocks:
            yield (i, j, k)

    def _golden_arrays(self):
        """(blocks, tile_keys, tile_isbias, tile_chan, tile_bias) in the layout of kn_conv2dtiled_create."""
        (Cout, Cin) = (self._outshape[0], self._inshape[0])
        lastcol_at = Cin * self._inshape[1] * self._inshape[2]
        biask = set(int(b[2]) for b in self._blocks if b[1] == lastcol_at and self.shape[1] == lastcol_at + 1)
        keys = list(self._tiles.keys())
        isbias = np.array([k[2] in biask for k in keys], dtype=np.uint8)
        chan = [np.asarray(self._tiles[k], dtype=np.float32) for (k, b) in zip(keys, isbias) if not b]
        bvals = [float(np.asarray(self._tiles[k]).reshape(())) for (k, b) in zip(keys, isbias) if b]
        return (np.array(self._blocks, dtype=np.int64).reshape(-1, 3), np.array(keys, dtype=np.int64).reshape(-1, 3), isbias,
                np.stack(chan) if len(chan) else np.zeros((0, Cout, Cin), np.float32), np.array(bvals, dtype=np.float32))

    def _device_op(self, device=None):
        def make():
            if self._taps is not None:
                t = self._taps
                return _capi.Operator.convtaps(self._inshape, self._outshape, t['taps'], t['ent_out'], t['ent_in'], t['ent_tap'], t['ent_coef'], t['lastcol'])
            (bl, tk, ib, ch, bs) = self._golden_arrays()
            return _capi.Operator.conv2dtiled(self.shape, self._inshape, self._outshape, bl, tk, ib, ch, bs)
        return _on_device(self, '_op', make, device)

    def torchdot(self, x, relu=False, exact=False, absmax=None):
        """[cols, N] -> [rows, N].  exact=False: f32 MFMA path (f32-input matrix instructions, exact f32 products); exact=True: the
        reference's accumulation order and rounding (order-preserving kernel on the factored operator); exact='split': a filled-in factored
        operator applied as spatial mixing per tap, then channel mixing (see _split_ops: another association of the sum, tolerance contract only); exact='bf16x3': f32 products
        emulated on the bf16 matrix pipe (three-way exact split, six of nine cross products, f32 accumulate: KN_FLAG_BF16X3) where the
        operator and batch qualify, else the f32 MFMA path."""
        if isinstance(x, np.ndarray):
            x = torch.as_tensor(x)
        if isinstance(exact, str):
            assert exact in ('bf16x3', 'split'), "exact must be True, False, 'bf16x3' or 'split'"
            if exact == 'split':
                return self._torchdot_split(x, relu=relu, absmax=absmax)
            return _run_torchdot(self._device_op, self.shape, x, relu=relu, exact=False, extra_flags=_capi.KN_FLAG_BF16X3, absmax=absmax)
        return _run_torchdot(self._device_op, self.shape, x, relu=relu, exact=exact, absmax=absmax)

    # ---- the SPLIT application of a filled-in operator (tolerance contract only) -----------------------------------------------------
    # A factored keyed conv is  sum_t F_t (x) K_t  with F_t the Cout x Cin matrix of tap t and K_t = a_out S_t a_in^-1 the HoWo x HiWi spatial
    # matrix of its entries (keynet_amd/direct.py).  The fused operator the reference stores costs  slots x Cin x Cout  multiply-adds per output
    # pixel and batch column; under a key whose inverse is dense inside its blocks (doubly-stochastic keys: 500 - 5 400 slots per pixel instead
    # of 9) that is 60x the un-keyed layer.  The same product in two steps --
    #     Z_t[ci] = K_t X[ci]                  (spatial mixing: one CSR of all taps' entries, applied to every input channel's plane)
    #     Y       = sum_t F_t Z_t              (channel mixing: an ordinary ntaps-slot conv-taps operator on Z, matrix cores, wave-uniform loaders)
    # -- costs  slots x Cin + ntaps x Cin x Cout.  It holds nothing the fused factored form does not hold (the same taps, the same entries), but it
    # is another association of the sum: NOT the reference's arithmetic, so it is a candidate of the float-key contract only (KeyedLayer._calibrate
    # measures it against the order-preserving kernel like the matrix-core kernel; layers that fail run in the reference's order as before).
    SPLIT_MIN_FILL = 2.0        # slots per (output pixel, tap) from which the split application is offered
    SPLIT_Z_BYTES = 32 << 30    # the intermediate Z is produced in column windows of at most this size (VGG-16 conv1_2 at 256 images: 29.6 GB in one window; 74-column windows
                                # under an 8 GB cap ran ragged tiles on the generic loader: 314 ms against 49 ms per 64 images)

    def fill_factor(self):
        """Entries per (output pixel, tap) of a factored operator: 1 for identity / permutation keys, 55 - 600 under doubly-stochastic keys."""
        t = self._taps
        if t is None or len(t['taps']) == 0:
            return 0.0
        return len(t['ent_out']) / float(len(t['taps']) * self._outshape[1] * self._outshape[2])

    def split_capable(self, n_vecs=None):
        """Is the split application worth offering to this operator (at this batch width)?  A filled-in factored operator whose split form is estimated at
        less than half the fused matrix-core launch: fused = entries x Cin x Cout multiply-adds at ~100 TFLOP/s (filled-in pixels run the slot-group / generic
        loaders); split = entries x Cin on the CSR kernels (~10 T MAC/s) + ntaps x HoWo x Cin x Cout on the matrix cores (~120 TFLOP/s) + the intermediate
        written and read once (~3 TB/s).  Givens-rotation keys (2 - 4 entries per pixel and tap) stay fused by this rule: their intermediate costs what the
        saved multiply-adds give back."""
        if self._taps is None or self.fill_factor() < self.SPLIT_MIN_FILL:
            return False
        if n_vecs is None:
            return True
        (Cin, Cout, HoWo) = (self._inshape[0], self._outshape[0], self._outshape[1] * self._outshape[2])
        (ent, nt, n) = (float(len(self._taps['ent_out'])), float(len(self._taps['taps'])), float(n_vecs))
        fused = 2.0 * ent * Cin * Cout * n / 100e12
        split = 2.0 * ent * Cin * n / 10e12 + 2.0 * nt * HoWo * Cin * Cout * n / 120e12 + 2.0 * (4.0 * Cin * nt * HoWo * n) / 3e12
        return split < 0.5 * fused

    def _split_arrays(self):
        """Host side of the split form: (K, second) with K the scipy CSR [ntaps * HoWo, HiWi] of all taps' entries (row t * HoWo + out, column in, value coef)
        and `second` the kwargs of the channel-mixing operator Conv2dTiledMatrix.fromtaps(**second) on Z [Cin, ntaps, HoWo]."""
        t = self._taps
        (Cin, Hin, Win) = self._inshape
        (Cout, Hout, Wout) = self._outshape
        (HoWo, HiWi, nt) = (Hout * Wout, Hin * Win, len(t['taps']))
        coef = t['ent_coef'] if t['ent_coef'] is not None else np.ones(len(t['ent_out']), np.float32)
        K = scipy.sparse.csr_matrix((coef.astype(np.float32), (t['ent_tap'].astype(np.int64) * HoWo + t['ent_out'], t['ent_in'].astype(np.int64))), shape=(nt * HoWo, HiWi))
        K.sort_indices()
        assert K.nnz == len(coef), 'an (output pixel, input pixel, tap) triple appears twice'
        eo = np.tile(np.arange(HoWo, dtype=np.int32), nt)
        et = np.repeat(np.arange(nt, dtype=np.int32), HoWo)
        second = dict(inshape=(Cin, nt, HoWo), outshape=self._outshape, taps=t['taps'], ent_out=eo, ent_in=(et.astype(np.int64) * HoWo + eo).astype(np.int32), ent_tap=et,
                      ent_coef=None, lastcol=t['lastcol'])
        return (K, second)

    def _split_ops(self, device=None):
        """(spatial CSR [ntaps * HoWo, HiWi], channel-mixing conv-taps operator on Z [Cin, ntaps, HoWo] -> [Cout, Hout, Wout]) resident on `device`."""
        def make():
            (K, f) = self._split_arrays()
            opK = _capi.Operator.csr(K.shape, K.indptr, K.indices, K.data)
            op2 = _capi.Operator.convtaps(f['inshape'], f['outshape'], f['taps'], f['ent_out'], f['ent_in'], f['ent_tap'], f['ent_coef'], f['lastcol'])
            return (opK, op2)
        return _on_device(self, '_op_split', make, device)

    def _torchdot_split(self, x, relu=False, absmax=None):
        assert self.shape[1] == x.shape[0], 'Non-conformal shape for W=%s, x=%s' % (str(self.shape), str(tuple(x.shape)))
        if not torch.cuda.is_available():
            raise _capi.KeynetHipError('keynet_amd: no MI355X visible -- the keyed forward has no CPU fallback')
        src_device = x.device
        xd = x.detach()
        xd = xd if xd.dtype == torch.float32 else xd.float()
        xd = xd if xd.is_cuda else xd.cuda()
        xd = xd if xd.is_contiguous() else xd.contiguous()
        (Cin, Hin, Win) = self._inshape
        (HoWo, HiWi, nt) = (self._outshape[1] * self._outshape[2], Hin * Win, len(self._taps['taps']))
        has_last = self._taps['lastcol'] is not None
        n = int(xd.shape[1])
        zrows = Cin * nt * HoWo + (1 if has_last else 0)
        cap = self.SPLIT_Z_BYTES
        with torch.cuda.device(xd.device):
            (free_b, _) = torch.cuda.mem_get_info()
            free_b += torch.cuda.memory_reserved() - torch.cuda.memory_allocated()       # what the caching allocator holds but does not use is available too
        cap = min(cap, max(free_b - 4 * self.shape[0] * n, 0) // 2)                      # the intermediate takes at most half of what is left beside the output block
        win = max(1, min(n, int(cap // (4 * zrows))))
        if win < n and win >= 64:
            win -= win % (128 if win >= 128 else 64)      # whole tiles of the matrix-core kernel per window
        y = torch.empty((self.shape[0], n), dtype=torch.float32, device=xd.device)
        flags = _capi.KN_FLAG_RELU if relu else 0
        with torch.cuda.device(xd.device):
            (opK, op2) = self._split_ops(xd.device)
            st = _stream_ptr()
            z = torch.empty((zrows, min(win, n)), dtype=torch.float32, device=xd.device)
            for c0 in range(0, n, win):
                w = min(win, n - c0)
                ldz = int(z.shape[1])
                # the spatial CSR on every input channel's plane (rows ci * HiWi ..) -> Z rows ci * ntaps * HoWo ..: ONE launch over all planes (kn_spmm_planes, round 6:
                # conv1_2 .. conv5_3 of the doubly-stochastic VGG-16 used to issue 64 .. 512 launches of 37 us each per layer and window), else plane by plane
                if not opK.spmm_planes(xd.data_ptr() + 4 * c0, n, HiWi * n, Cin, w, z.data_ptr(), ldz, nt * HoWo * ldz, _capi.KN_FLAG_EXACT, st):
                    for ci in range(Cin):
                        opK.spmm(xd.data_ptr() + 4 * (ci * HiWi * n + c0), n, w, z.data_ptr() + 4 * ci * nt * HoWo * ldz, ldz, _capi.KN_FLAG_EXACT, st)
                if has_last:
                    z[-1, :w].copy_(xd[-1, c0:c0 + w])
                op2.spmm(z.data_ptr(), ldz, w, y.data_ptr() + 4 * c0, n, flags, st, absmax_ptr=None if absmax is None else absmax.data_ptr())
        return y if src_device.type == 'cuda' else y.to(src_device)

    def _expand_taps_host(self, pixels=None, channels=None):
        """Canonical CSR of a factored operator -- or of its output rows (co, o) for o in `pixels` only, numbered
        co * len(pixels) + position of o -- on the host: the reference's expansion rule (keynet/sparse.py:802-812; kn_export_csr
        does the same on the C side; `channels` further restricts the rows to the first `channels` output channels).  Built
        directly in CSR order, one block copy + one table gather per output channel: row (co, o) holds,
        for ci ascending, the pixel's slots by ascending input pixel, then the bias entry -- exactly the column-sorted row scipy's
        csr_matrix((v,(r,c))) would produce, without ever holding a COO copy (conv5_1 of VGG-16: 420 M entries)."""
        t = self._taps
        (Cout, Hout, Wout) = self._outshape
        (Cin, Hin, Win) = self._inshape
        (HoWo, HiWi) = (Hout * Wout, Hin * Win)
        pixels = np.arange(HoWo, dtype=np.int64) if pixels is None else np.asarray(pixels, dtype=np.int64)
        npx = len(pixels)
        pos = -np.ones(HoWo, dtype=np.int64)
        pos[pixels] = np.arange(npx)
        sel = np.flatnonzero(pos[t['ent_out']] >= 0)
        order = sel[np.lexsort((t['ent_in'][sel], pos[t['ent_out'][sel]]))]          # by (pixel position, input pixel), stable
        (eo, ei, et) = (pos[t['ent_out'][order]], t['ent_in'][order].astype(np.int64), t['ent_tap'][order])
        coef = None if t['ent_coef'] is None else t['ent_coef'][order]
        if len(eo) > 1 and np.any((eo[1:] == eo[:-1]) & (ei[1:] == ei[:-1])):
            # several taps on one (output, input) pixel pair: ONE stored entry per channel pair, the float32 sum of its terms fl(coef * tap) in entry order.  The selected
            # pixels' pairs become the taps of an equivalent duplicate-free operator (one [Cout, Cin] matrix per pair, summed one term position at a time), which the
            # block-copy route below expands -- the per-entry COO route (_expand_taps_host_coo) holds 3 x 8 bytes per stored value and sorts them all.
            first = np.ones(len(eo), dtype=bool)
            first[1:] = (eo[1:] != eo[:-1]) | (ei[1:] != ei[:-1])
            seg = np.cumsum(first) - 1
            rank = np.arange(len(eo)) - np.flatnonzero(first)[seg]
            tp = t['taps']
            def term(idx):
                v = tp[et[idx]]
                if coef is None:
                    return v
                cf = coef[idx][:, None, None]
                return np.where(cf == 1.0, v, cf * v).astype(np.float32)
            V = term(np.flatnonzero(first)).astype(np.float32, copy=True)
            for k in range(1, int(rank.max()) + 1):
                idx = np.flatnonzero(rank == k)
                V[seg[idx]] = (V[seg[idx]] + term(idx)).astype(np.float32)
            W2 = Conv2dTiledMatrix.fromtaps(self._inshape, self._outshape, V, pixels[eo[first]], ei[first], np.arange(len(V), dtype=np.int32), None, t['lastcol'])
            return W2._expand_taps_host(pixels, channels)
        Cout = Cout if channels is None else int(channels)                                  # several taps on one (out, in) pair: scipy sums them
        ns = np.bincount(eo, minlength=npx)                                           # slots per selected pixel
        first = np.concatenate(([0], np.cumsum(ns)))[:-1]
        has_last = t['lastcol'] is not None
        rows_n = Cout * npx + (1 if (has_last and npx == HoWo and channels is None) else 0)
        counts = np.tile(ns * Cin, Cout)
        if has_last:
            lastv = t['lastcol'][(np.arange(Cout)[:, None] * HoWo + pixels[None, :]).ravel()]
            counts = counts + (lastv != 0)
        if rows_n > Cout * npx:
            counts = np.concatenate((counts, [1 if t['lastcol'][-1] != 0 else 0]))
        indptr = np.zeros(rows_n + 1, dtype=np.int64)
        np.cumsum(counts, out=indptr[1:])
        total = int(indptr[-1])
        idt = np.int32 if max(total, self.shape[1]) < 2 ** 31 - 1 else np.int64
        indices = np.empty(total, dtype=idt)
        data = np.empty(total, dtype=np.float32)
        taps = t['taps']
        ntaps = taps.shape[0]
        # The Cout rows of a pixel share one column sequence, and the rows of one output channel are contiguous in the result:
        # the channel's block is ONE copy of a precomputed column image and ONE gather from that channel's small [ntaps, Cin]
        # value table.  The image is rebuilt only when the bias column's zero pattern differs between channels.
        cached_mask = None
        for co in range(Cout):
            mask = (lastv[co * npx:(co + 1) * npx] != 0) if has_last else np.zeros(npx, dtype=bool)
            if cached_mask is None or not np.array_equal(mask, cached_mask):
                cached_mask = mask
                cnt = ns * Cin + mask
                loc = np.concatenate(([0], np.cumsum(cnt)))
                col_img = np.empty(int(loc[-1]), dtype=idt)
                src_img = np.zeros(int(loc[-1]), dtype=np.int64)                       # index into taps[:, co, :].ravel()
                ent_img = np.zeros(int(loc[-1]), dtype=np.int64)                       # entry id (for the coefficients)
                for k in np.unique(ns):
                    if k == 0:
                        continue
                    P = np.flatnonzero(ns == k)
                    g = first[P][:, None] + np.arange(k)[None, :]                      # [P, k] entry ids, input pixels ascending
                    dest = loc[P][:, None] + np.arange(Cin * k)[None, :]
                    col_img[dest] = ((np.arange(Cin) * HiWi)[None, :, None] + ei[g][:, None, :]).reshape(len(P), Cin * k)
                    src_img[dest] = (et[g].astype(np.int64)[:, None, :] * Cin + np.arange(Cin)[None, :, None]).reshape(len(P), Cin * k)
                    ent_img[dest] = np.broadcast_to(g[:, None, :], (len(P), Cin, k)).reshape(len(P), Cin * k)
                bias_at = loc[1:][mask] - 1
                col_img[bias_at] = self.shape[1] - 1
            (lo, hi) = (int(indptr[co * npx]), int(indptr[(co + 1) * npx]))
            indices[lo:hi] = col_img
            v = taps[:, co, :].ravel()[src_img]
            if coef is not None:
                cf = coef[ent_img]
                v = np.where(cf == 1.0, v, cf * v)
            if has_last:
                v[bias_at] = lastv[co * npx:(co + 1) * npx][mask]
            data[lo:hi] = v
        if rows_n > Cout * npx and t['lastcol'][-1] != 0:
            indices[-1] = self.shape[1] - 1
            data[-1] = t['lastcol'][-1]
        return scipy.sparse.csr_matrix((data, indices, indptr.astype(idt)), shape=(rows_n, self.shape[1]))

    def _expand_taps_host_coo(self, pixels):
        """General route of _expand_taps_host: duplicate (out, in) pairs become one stored entry, their terms summed in float32 in entry order."""
        t = self._taps
        (Cout, Hout, Wout) = self._outshape
        (Cin, Hin, Win) = self._inshape
        (HoWo, HiWi) = (Hout * Wout, Hin * Win)
        npx = len(pixels)
        pos = -np.ones(HoWo, dtype=np.int64)
        pos[pixels] = np.arange(npx)
        sel = np.flatnonzero(pos[t['ent_out']] >= 0)
        coef = t['ent_coef'][sel] if t['ent_coef'] is not None else np.ones(len(sel), np.float32)
        (ic, jc) = np.meshgrid(np.arange(Cout), np.arange(Cin), indexing='ij')
        rows = (pos[t['ent_out'][sel]][:, None, None] + (ic * npx)[None]).ravel()
        cols = (t['ent_in'][sel].astype(np.int64)[:, None, None] + (jc * HiWi)[None]).ravel()
        tv = t['taps'][t['ent_tap'][sel]]
        vals = np.where(coef[:, None, None] == 1.0, tv, coef[:, None, None] * tv).astype(np.float32).ravel()
        rows_n = Cout * npx
        if t['lastcol'] is not None:
            lastv = t['lastcol'][(np.arange(Cout)[:, None] * HoWo + pixels[None, :]).ravel()]
            if npx == HoWo:
                lastv = np.concatenate((lastv, t['lastcol'][-1:]))
                rows_n += 1
            nz = np.flatnonzero(lastv)
            rows = np.concatenate((rows, nz))
            cols = np.concatenate((cols, np.full(len(nz), self.shape[1] - 1, dtype=np.int64)))
            vals = np.concatenate((vals, lastv[nz]))
        # duplicates -- several taps on one (output, input) pixel pair -- are ONE stored entry: the float32 sum of their terms in ENTRY order (what the device kernels and
        # kn_export_csr compute).  scipy's own COO -> CSR conversion sums them behind an UNSTABLE sort, which defines no order for three or more terms: summed here.
        order = np.lexsort((np.arange(len(rows)), cols, rows))                           # by (row, column), entry order inside a pair
        (r, c, v) = (rows[order], cols[order], vals[order].astype(np.float32))
        first = np.ones(len(r), dtype=bool)
        first[1:] = (r[1:] != r[:-1]) | (c[1:] != c[:-1])
        seg = np.cumsum(first) - 1                                                      # stored entry of every term
        rank = np.arange(len(r)) - np.flatnonzero(first)[seg]                           # its position inside the entry
        acc = v[first].copy()
        for k in range(1, int(rank.max()) + 1 if len(rank) else 1):                     # one vectorised pass per term position: acc = fl(acc + term_k)
            sel_k = np.flatnonzero(rank == k)
            acc[seg[sel_k]] = (acc[seg[sel_k]] + v[sel_k]).astype(np.float32)
        indptr = np.zeros(rows_n + 1, dtype=np.int64)
        np.add.at(indptr, r[first] + 1, 1)
        np.cumsum(indptr, out=indptr)
        idt = np.int32 if max(len(acc), self.shape[1]) < 2 ** 31 - 1 else np.int64
        return scipy.sparse.csr_matrix((acc, c[first].astype(idt), indptr.astype(idt)), shape=(rows_n, self.shape[1]))

    def rows_csr(self, pixels=None, channels=None):
        """Canonical CSR of the output rows (co, o), o in `pixels` (channel-major: row = co * len(pixels) + index of o), of a
        factored operator: the reference's expansion rule (keynet/sparse.py:802-812) restricted to those rows.  Host side;
        lets a CPU checker or a CPU baseline work on a slice of an operator whose full CSR would be tens of GB.  With every
        pixel selected this is the whole operator incl. its homogeneous row."""
        assert self._taps is not None, 'rows_csr needs the factored form (fromtaps / direct keying)'
        return self._expand_taps_host(pixels, channels)

    def max_abs_rowsum(self):
        """max over output rows (co, o) of sum |W| along the row, from the factored form without expanding it: for row (co, o) it is
        sum over the pixel's entries of |coef| * sum_ci |taps[tap][co][ci]|  (+ |last column|) -- one [HoWo x ntaps] sparse product."""
        if self._taps is None:
            M = self.tosparse('csr')
            return float(abs(M).sum(axis=1).max()) if M.nnz else 0.0
        t = self._taps
        (Cout, Hout, Wout) = self._outshape
        HoWo = Hout * Wout
        tap_abs = np.abs(t['taps'].astype(np.float64)).sum(axis=2)                     # [ntaps, Cout]
        coef = np.ones(len(t['ent_out'])) if t['ent_coef'] is None else np.abs(t['ent_coef'].astype(np.float64))
        S = scipy.sparse.csr_matrix((coef, (t['ent_out'].astype(np.int64), t['ent_tap'].astype(np.int64))), shape=(HoWo, tap_abs.shape[0]))
        rs = S.dot(tap_abs)                                                            # [HoWo, Cout]
        if t['lastcol'] is not None:
            rs = rs + np.abs(t['lastcol'][:Cout * HoWo].astype(np.float64)).reshape(Cout, HoWo).T
        return float(rs.max()) if rs.size else 0.0

    def nnz(self):
        """Stored parameters: sum of tile sizes (keynet/sparse.py:778) -- or, for a factored operator, taps + entries +
        last column (what is actually stored)."""
        if self._tiles is None:
            t = self._taps
            return int(t['taps'].size + len(t['ent_out']) + (np.count_nonzero(t['lastcol']) if t['lastcol'] is not None else 0))
        return sum([v.size for v in self._tiles.values()])

    def transpose(self):
        raise NotImplementedError('transpose of a Conv2dTiledMatrix is not on the forward path')

    def tosparse(self, format='coo'):
        """Expansion rule of keynet/sparse.py:802-812, vectorised."""
        if self._tiles is None:
            return self._expand_taps_host().asformat(format)
        (Cout, Hout, Wout) = self._outshape
        (Cin, Hin, Win) = self._inshape
        bykey = {}
        for ((it, jt, k), m) in self._tiles.items():
            bykey.setdefault(k, []).append((it, jt, m))
        (R, C, V) = ([], [], [])
        for (i, j, k) in self._blocks:
            for (it, jt, m) in bykey.get(k, []):
                (ic, jc) = np.meshgrid(np.arange(m.shape[0]), np.arange(m.shape[1]), indexing='ij')
                R.append((i + it + ic * Hout * Wout).ravel())
                C.append((j + jt + jc * Hin * Win).ravel())
                V.append(np.asarray(m, dtype=np.float32).ravel())
        return _from_coo(format, np.concatenate(V), np.concatenate(R), np.concatenate(C), self.shape)


def _conv_tiler(T, blocks, inshape, outshape, tileshape):
    """Vectorised restatement of Conv2dTiledMatrix.__tiler__ (keynet/sparse.py:692-717): returns the ordered dict
    {(it, jt, k): float32[Cout, Cin]}; where several blocks share a tile id the LAST entry in COO order wins."""
    (Cin, Hin, Win) = inshape
    (Cout, Hout, Wout) = outshape
    (h, w) = tileshape
    (HoWo, HiWi) = (Hout * Wout, Hin * Win)
    tiles = {}
    loc = {}
    for (ib, jb, kt) in blocks:
        loc[(ib, jb)] = kt
        tiles[(0, 0, kt)] = np.zeros((Cout, Cin), dtype=np.float32)
    if T.nnz == 0:
        return tiles
    (i, j, v) = (np.asarray(T.row, dtype=np.int64), np.asarray(T.col, dtype=np.int64), np.asarray(T.data, dtype=np.float32))
    (ip, jp) = (i % HoWo, j % HiWi)
    (ib, jb) = (h * (ip // h), w * (jp // w))
    nbj = HiWi // w + 1
    lut = {}
    bkeys = ib // h * nbj + jb // w
    for ((bi_, bj_), kt) in loc.items():
        lut[bi_ // h * nbj + bj_ // w] = kt
    (ub, inv) = np.unique(bkeys, return_inverse=True)
    kt_of = np.array([lut.get(int(b), -1) for b in ub], dtype=np.int64)[inv]
    keep = kt_of >= 0
    (i, j, v, ip, jp, ib, jb, kt_of) = (i[keep], j[keep], v[keep], ip[keep], jp[keep], ib[keep], jb[keep], kt_of[keep])
    (it, jt, ic, jc) = (ip - ib, jp - jb, i // HoWo, j // HiWi)
    tkey = (kt_of * h + it) * w + jt                      # (it, jt, kt) flattened
    (utk, first_idx, tinv) = np.unique(tkey, return_index=True, return_inverse=True)
    for u in np.argsort(first_idx, kind='stable'):          # dict insertion order = first appearance in COO order
        k3 = (int(utk[u] // w % h), int(utk[u] % w), int(utk[u] // (w * h)))
        if k3 not in tiles:
            tiles[k3] = np.zeros((Cout, Cin), dtype=np.float32)
    # assign; later entries overwrite earlier ones (numpy fancy assignment keeps the last for repeated indices)
    stack = np.zeros((len(utk), Cout, Cin), dtype=np.float32)
    stack[tinv, ic, jc] = v
    for (u, tk) in enumerate(utk):
        k3 = (int(tk // w % h), int(tk % w), int(tk // (w * h)))
        tiles[k3] = stack[u]
    return tiles


# ------------------------------------------------------------------------------------------------------------------
# Build-time constructors (host, offline)
def sparse_toeplitz_conv2d(inshape, f, bias=None, as_correlation=True, stride=1, format='csr'):
    """Explicit sparse matrix of a 'same'-padded, odd, square conv layer acting on the flattened CxUxV image, with the
    bias column and homogeneous row when `bias` is given: conv2d(img, f) == W.dot(img.flatten()).

    Restates keynet/sparse.py:122-203 without the per-entry Python loop.  Rows are (cout, u/stride, v/stride), columns
    (cin, u+p, v+q); taps falling outside the image are omitted.  Values are fl32(fl32(w + off) - off) with
    off = |min w| + 1 exactly as the reference's sparsity-preserving round trip leaves them (SURVEY 8c iv)."""
    assert len(inshape) == 3 and f.ndim == 4
    assert f.shape[1] == inshape[0] and f.shape[2] == f.shape[3] and f.shape[2] % 2 == 1
    assert as_correlation, 'only the correlation form is on the keyed path'
    (C, U, V) = inshape
    (M, _, P, Q) = f.shape
    f = np.asarray(f, dtype=np.float32)
    (Us, Vs) = (U // stride, V // stride)
    (ku, kv) = np.meshgrid(np.arange(Us), np.arange(Vs), indexing='ij')
    (u, v) = (np.arange(0, U, stride)[ku], np.arange(0, V, stride)[kv])
    (R, Cc, D) = ([], [], [])
    (cin, cout) = np.meshgrid(np.arange(C), np.arange(M), indexing='ij')
    for (i, p) in enumerate(range(-((P - 1) // 2), ((P - 1) // 2) + 1)):
        for (j, q) in enumerate(range(-((Q - 1) // 2), ((Q - 1) // 2) + 1)):
            ok = ((u + p) >= 0) & ((u + p) < U) & ((v + q) >= 0) & ((v + q) < V)
            (opix, ipix) = ((ku * Vs + kv)[ok], ((u + p) * V + (v + q))[ok])
            R.append((cout.reshape(1, -1) * (Us * Vs) + opix.reshape(-1, 1)).ravel())
            Cc.append((cin.reshape(1, -1) * (U * V) + ipix.reshape(-1, 1)).ravel())
            D.append(np.broadcast_to(f[cout, cin, i, j].reshape(1, -1), (len(opix), C * M)).ravel())
    (rows, cols, data) = (np.concatenate(R), np.concatenate(Cc), np.concatenate(D).astype(np.float32))
    off = np.float32(np.abs(np.min(data)) + np.float32(1.0))
    data = (data + off) - off
    A = scipy.sparse.coo_matrix((data, (rows, cols)), shape=(M * Us * Vs, C * U * V))
    if bias is not None:
        bias = np.asarray(bias, dtype=np.float32)
        assert bias.ndim == 1 and bias.shape[0] == M
        offb = np.float32(np.abs(np.min(bias)) + np.float32(1.0))
        bcol = ((np.repeat(bias, Us * Vs) + offb) - offb).astype(np.float32)
        lastcol = scipy.sparse.coo_matrix((bcol, (np.arange(M * Us * Vs), np.zeros(M * Us * Vs, dtype=np.int64))), shape=(A.shape[0], 1))
        lastrow = scipy.sparse.coo_matrix(([1], ([0], [A.shape[1]])), shape=(1, A.shape[1] + 1), dtype=np.float32)
        A = scipy.sparse.vstack((scipy.sparse.hstack((A, lastcol)), lastrow))
    return A.tocsr() if format == 'csr' else A


def sparse_toeplitz_avgpool2d(inshape, filtershape, stride):
    """Average pooling as a conv with a (1/k^2)-filled diagonal filter and a zero bias (keynet/sparse.py:206-212):
    k x k window, `stride`, zero padding (k-1)/2 counted in the mean -- independent of the source module's padding."""
    (outchannel, inchannel, k, _) = filtershape
    F = np.zeros(filtershape, dtype=np.float32)
    F[np.arange(outchannel), np.arange(outchannel), :, :] = 1.0 / (k * k)
    return sparse_toeplitz_conv2d(inshape, F, bias=np.zeros(outchannel, dtype=np.float32), stride=stride)


def sparse_affine_to_linear(A, bias=None, dtype=np.float32):
    """The homogeneous form [[A, b], [0, 1]] of x -> A x + b as one sparse matrix (keynet/sparse.py:87-96); b = 0 when no bias."""
    assert is_scipy_sparse(A)
    (m, n) = A.shape
    if bias is not None:
        assert bias.shape[0] == m and bias.shape[1] == 1
    b = scipy.sparse.coo_matrix(bias) if bias is not None else scipy.sparse.coo_matrix((m, 1), dtype=dtype)
    one = scipy.sparse.coo_matrix(([1], ([0], [n])), shape=(1, n + 1), dtype=dtype)
    return scipy.sparse.vstack((scipy.sparse.hstack((A, b)), one))


def sparse_identity_matrix(n, dtype=np.float32):
    return scipy.sparse.eye(n, dtype=dtype)


def sparse_permutation_matrix(n, dtype=np.float32, withinverse=False):
    """Random n x n permutation matrix, row r holding its one in column perm[r]; ONE np.random.permutation(n) draw from numpy's
    global RNG (keynet/sparse.py:280-285).  The inverse of a permutation matrix is its transpose."""
    perm = np.random.permutation(n)
    P = scipy.sparse.csr_matrix((np.ones(n, dtype=dtype), perm, np.arange(n + 1)), shape=(n, n))
    return (P, P.transpose()) if withinverse else P
