"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's keyed forward path.

Parity status: PINNED.  Every function below is checked (tests/test_oracle_golden.py) against outputs produced by the
reference itself (visym/keynet imported from /root/reference in the build container; vectors committed under
tests/golden/, generator tests/golden/make_golden.py), bit-for-bit for f32 operators, and against the one literal
known answer in the reference repo (demo/challenge.ipynb cell 5).

What is restated (reference file:line -> here):
    keynet/sparse.py:488-492  SparseMatrix.torchdot      -> csr_matvecs() via oracle/kn_oracle.c (scipy csr_matvecs algorithm)
    keynet/sparse.py:621-641  TiledMatrix.tosparse       -> tiled_to_csr()
    keynet/sparse.py:683-687  DiagonalTiledMatrix.__iter__ (blocks are dumped by the generator; nothing to compute)
    keynet/sparse.py:781-835  Conv2dTiledMatrix._tosparse/tosparse -> conv2dtiled_to_csr()
    keynet/sparse.py:603-612  TiledMatrix.torchdot       -> expand to canonical CSR, then csr_matvecs()
    keynet/layer.py:88-93     KeyedLayer.forward         -> layer_forward()
    keynet/system.py:130-133  KeyedModel.forward         -> keynet_forward() (+ linear_to_affine, keynet/torch.py:71-77)
    keynet/torch.py:65-68     affine_to_linear           -> affine_to_linear()

The arithmetic itself lives in a third-party dependency absent from /root/reference: scipy.sparse._sparsetools
(scipy unpinned in the reference's setup.py:22-31; scipy 1.15.3 in this image).  Its published algorithm is restated
in C in oracle/kn_oracle.c.  There is no oracle/_ref: the reference is Python and cannot be compiled or shipped.
"""
import os
import ctypes
import subprocess
import numpy as np

__all__ = ['csr_matvecs', 'relu_', 'coo_to_canonical_csr', 'tiled_to_csr', 'conv2dtiled_to_csr', 'operator_from_golden',
           'layer_forward', 'keynet_forward', 'affine_to_linear', 'linear_to_affine', 'load_golden_layers', 'build', 'tiled_torchdot_loop']

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle/kn_oracle.c -> oracle/libkn_oracle.so with gcc (-ffp-contract=off)."""
    so = os.path.join(_HERE, 'libkn_oracle.so')
    src = os.path.join(_HERE, 'kn_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-ffp-contract=off', '-fno-fast-math', '-o', so, src])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(os.environ.get('KN_ORACLE_LIB') or build())     # KN_ORACLE_LIB: another BUILD of kn_oracle.c (tests/test_host_sanitize.py: ASan + UBSan)
        i64 = ctypes.c_int64
        p = ctypes.c_void_p
        L.kn_oracle_csr_matvecs_f32.argtypes = [i64, i64, i64, p, p, p, p, p]
        L.kn_oracle_csr_matvecs_f32.restype = None
        L.kn_oracle_csr_matvecs_f64.argtypes = [i64, i64, i64, p, p, p, p, p]
        L.kn_oracle_csr_matvecs_f64.restype = None
        L.kn_oracle_relu_f32.argtypes = [i64, p]
        L.kn_oracle_relu_f32.restype = None
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def csr_matvecs(shape, indptr, indices, data, X):
    """Y = W.dot(X) exactly as scipy's csr_matvecs computes it (stored order, mul then add, dtype upcast as numpy).

    X: [n_col, n_vecs] (any strides; scipy ravels to C order first).  Returns C-order [n_row, n_vecs]."""
    (n_row, n_col) = (int(shape[0]), int(shape[1]))
    assert X.shape[0] == n_col, 'dimension mismatch'
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    assert len(indptr) == n_row + 1 and indptr[-1] == len(indices) == len(data)
    rt = np.result_type(data.dtype, X.dtype)
    n_vecs = int(X.shape[1])
    if rt == np.float32:
        A = np.ascontiguousarray(data, dtype=np.float32)
        Xc = np.ascontiguousarray(X, dtype=np.float32)
        Y = np.zeros((n_row, n_vecs), dtype=np.float32)
        _lib().kn_oracle_csr_matvecs_f32(n_row, n_col, n_vecs, _ptr(indptr), _ptr(indices), _ptr(A), _ptr(Xc), _ptr(Y))
    else:
        A = np.ascontiguousarray(data, dtype=np.float64)
        Xc = np.ascontiguousarray(X, dtype=np.float64)
        Y = np.zeros((n_row, n_vecs), dtype=np.float64)
        _lib().kn_oracle_csr_matvecs_f64(n_row, n_col, n_vecs, _ptr(indptr), _ptr(indices), _ptr(A), _ptr(Xc), _ptr(Y))
    return Y


def relu_(Y):
    assert Y.dtype == np.float32 and Y.flags['C_CONTIGUOUS']
    _lib().kn_oracle_relu_f32(Y.size, _ptr(Y))
    return Y


def coo_to_canonical_csr(shape, rows, cols, vals):
    """scipy.sparse.csr_matrix((vals,(rows,cols)), shape): COO -> CSR, indices sorted per row, duplicates summed.

    The tile expansions never produce duplicates (each (row,col) is emitted once); asserted rather than summed so a
    summation-order question cannot arise."""
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    vals = np.asarray(vals)
    order = np.lexsort((cols, rows))
    (r, c, v) = (rows[order], cols[order], vals[order])
    if len(r) > 1:
        assert not np.any((r[1:] == r[:-1]) & (c[1:] == c[:-1])), 'duplicate entries in tile expansion'
    indptr = np.zeros(int(shape[0]) + 1, dtype=np.int64)
    np.add.at(indptr, r + 1, 1)
    indptr = np.cumsum(indptr).astype(np.int32)
    return (indptr, c.astype(np.int32), v)


def tiled_to_csr(shape, blocks, tile_ptr, tile_row, tile_col, tile_val):
    """keynet/sparse.py:621-641: for (i,j,k) in blocks: rows += i + tiles[k].row; cols += j + tiles[k].col."""
    (R, C, V) = ([], [], [])
    for (i, j, k) in np.asarray(blocks, dtype=np.int64):
        s = slice(int(tile_ptr[k]), int(tile_ptr[k + 1]))
        R.append(i + tile_row[s].astype(np.int64))
        C.append(j + tile_col[s].astype(np.int64))
        V.append(tile_val[s])
    cat = (lambda L, dt: np.concatenate(L) if len(L) else np.zeros(0, dt))
    return coo_to_canonical_csr(shape, cat(R, np.int64), cat(C, np.int64), cat(V, np.float32))


def conv2dtiled_to_csr(shape, inshape, outshape, blocks, tile_keys, tile_isbias, tile_chan, tile_bias):
    """keynet/sparse.py:781-814: W[i+it+ic*HoutWout, j+jt+jc*HinWin] = tiles[(it,jt,k)][ic,jc] for every block (i,j,k)."""
    (Cin, Hin, Win) = [int(v) for v in inshape]
    (Cout, Hout, Wout) = [int(v) for v in outshape]
    tile_keys = np.asarray(tile_keys, dtype=np.int64).reshape(-1, 3)
    bykey = {}
    (nc, nb) = (0, 0)
    for (e, (it, jt, k)) in enumerate(tile_keys):
        if tile_isbias[e]:
            m = np.asarray(tile_bias[nb], dtype=np.float32).reshape(1, 1)
            nb += 1
        else:
            m = tile_chan[nc]
            nc += 1
        bykey.setdefault(int(k), []).append((int(it), int(jt), m))
    (R, C, V) = ([], [], [])
    for (i, j, k) in np.asarray(blocks, dtype=np.int64):
        for (it, jt, m) in bykey.get(int(k), []):
            (ic, jc) = np.meshgrid(np.arange(m.shape[0]), np.arange(m.shape[1]), indexing='ij')
            R.append((i + it + ic * Hout * Wout).ravel())
            C.append((j + jt + jc * Hin * Win).ravel())
            V.append(np.asarray(m, dtype=np.float32).ravel())
    return coo_to_canonical_csr(shape, np.concatenate(R), np.concatenate(C), np.concatenate(V))


def operator_from_golden(z, prefix):
    """(shape, indptr, indices, data) of the operator the reference applies, rebuilt from the STRUCTURED dump where one
    exists (tiles/blocks), else the stored CSR triplet."""
    kind = str(z[prefix + 'kind'])
    shape = tuple(int(v) for v in z[prefix + 'shape'])
    if kind == 'csr':
        return (shape, z[prefix + 'indptr'], z[prefix + 'indices'], z[prefix + 'data'])
    if kind in ('tiled', 'diagtiled'):
        return (shape,) + tiled_to_csr(shape, z[prefix + 'blocks'], z[prefix + 'tile_ptr'], z[prefix + 'tile_row'], z[prefix + 'tile_col'], z[prefix + 'tile_val'])
    if kind == 'conv2dtiled':
        return (shape,) + conv2dtiled_to_csr(shape, z[prefix + 'inshape'], z[prefix + 'outshape'], z[prefix + 'blocks'], z[prefix + 'tile_keys'],
                                             z[prefix + 'tile_isbias'], z[prefix + 'tile_chan'], z[prefix + 'tile_bias'])
    raise ValueError(kind)


def layer_forward(op, x_affine, relu=False):
    """keynet/layer.py:88-93: y = W.torchdot(x_affine.t()).t(); F.relu for a keyed ReLU layer.  x_affine: [N, Din+1]."""
    (shape, indptr, indices, data) = op
    Y = csr_matvecs(shape, indptr, indices, data, x_affine.T)
    if relu:
        Y = relu_(np.ascontiguousarray(Y, dtype=np.float32))
    return Y.T


def affine_to_linear(x):
    """keynet/torch.py:65-68: NxCxHxW -> Nx(C*H*W+1), last column one."""
    x = np.asarray(x)
    x = x.reshape((1,) + x.shape) if x.ndim == 3 else x
    N = x.shape[0]
    return np.concatenate((x.reshape(N, -1), np.ones((N, 1), dtype=x.dtype)), axis=1)


def linear_to_affine(x, outshape=None):
    """keynet/torch.py:71-77."""
    assert x.ndim == 2
    if not np.allclose(x[:, -1], 1, atol=1e-3):
        raise ValueError('invalid affine vector')
    xa = x[:, :-1]
    return xa.reshape(outshape) if outshape is not None else xa


def load_golden_layers(z, structured=True):
    """[(name, 'relu'|op, keyed_relu)] in nn.Sequential order from a golden .npz."""
    layers = []
    for name in [str(n) for n in z['layer_names']]:
        p = 'L.%s.' % name
        if str(z[p + 'kind']) == 'relu':
            layers.append((name, 'relu', False))
        else:
            op = operator_from_golden(z, p) if structured else (tuple(int(v) for v in z[p + 'shape']), z[p + 'indptr'], z[p + 'indices'], z[p + 'data'])
            layers.append((name, op, 'ReLU' in str(z[p + 'layertype'])))
    return layers


def keynet_forward(layers, x_cipher, collect=False):
    """keynet/system.py:132: nn.Sequential of KeyedLayer / nn.ReLU on [N, D0+1]; float64 outputs are coerced back to
    float32 at the next layer's input (keynet/sparse.py:489-491)."""
    y = np.asarray(x_cipher)
    outs = {}
    for (name, op, keyed_relu) in layers:
        if isinstance(op, str):
            y = np.maximum(y, 0) if y.dtype != np.float32 else relu_(np.array(y, dtype=np.float32, order='C', copy=True))
        else:
            y = layer_forward(op, y.astype(np.float32) if y.dtype != np.float32 else y, relu=keyed_relu)
        if collect:
            outs[name] = y
    return (y, outs) if collect else y


def tiled_torchdot_loop(x, tileshape, shape, tiles, blocks):
    """Serial restatement of keynet/torch.py:173-184 (`keynet.torch.TiledMatrix._torchdot`): y[i+ii, :] += v * x[j+jj, :] over blocks, then over the
    entries of tile k, in f32 (the reference jit-compiles this loop with fastmath + parallel; serial, unfused f32 is the one well-defined reading)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.zeros((int(shape[0]), x.shape[1]), dtype=np.float32)
    for (i, j, k) in blocks:
        b = np.asarray(tiles[int(k)]).reshape(-1, 3)
        for u in range(b.shape[0]):
            (ii, jj, v) = (int(b[u, 0]), int(b[u, 1]), np.float32(b[u, 2]))
            y[int(i) + ii, :] = y[int(i) + ii, :] + v * x[int(j) + jj, :]
    return y
