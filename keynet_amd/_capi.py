"""ctypes binding of include/keynet_hip.h (the C ABI of libkeynet_hip.so).

The product path has NO CPU fallback: if the shared library is missing or no gfx950 device is visible, every compute
entry point raises.  (The CPU restatement lives in oracle/ and is test infrastructure only.)
"""
import os
import ctypes
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KEYNET_HIP_LIB: another BUILD of this same library (a diagnostic variant of the HIP kernels, e.g. tools/ablate_conv.sh's -DKN_ABLATION
# build); there is no other implementation behind this binding
LIBPATH = os.environ.get('KEYNET_HIP_LIB') or os.path.join(_HERE, 'libkeynet_hip.so')

KN_OK = 0
KN_FLAG_RELU = 1
KN_FLAG_EXACT = 2
KN_FLAG_BF16X3 = 4
KN_ABI_VERSION = 4

# every symbol include/keynet_hip.h declares (tests/test_capi.py checks the header against this list)
SYMBOLS = ['kn_abi_version', 'kn_last_error', 'kn_device_info', 'kn_csr_create', 'kn_csr_create_f64', 'kn_dtype_bits', 'kn_export_csr_f64', 'kn_spmm_f64', 'kn_tiled_create', 'kn_conv2dtiled_create',
           'kn_convtaps_create', 'kn_convtaps_drop_zero_entries', 'kn_dense_create', 'kn_chain_create', 'kn_destroy', 'kn_nnz', 'kn_nnz_expanded', 'kn_shape', 'kn_export_csr', 'kn_spmm', 'kn_spmm_planes', 'kn_spmm_screen', 'kn_absmax', 'kn_reserve_workspace', 'kn_release_side_tables', 'kn_spmm_plan', 'kn_relu',
           'kn_affine_to_linear', 'kn_linear_to_affine']


class KeynetHipError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded library; raises KeynetHipError loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise KeynetHipError('libkeynet_hip.so not found at %s: run `python -c "import __graft_entry__ as g; g.build()"` '
                                 '(hipcc --offload-arch=gfx950).  There is no CPU fallback.' % LIBPATH)
        # PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7, the same as /opt/rocm's).  It must be in the
        # process BEFORE this library is loaded so both bind to ONE HIP runtime; loaded the other way round the process
        # ends up with two runtimes and the second one sees no device.
        # (KEYNET_HIP_NO_TORCH=1: the host-only sanitizer build of tests/test_host_sanitize.py, which has no device side to share)
        if os.environ.get('KEYNET_HIP_NO_TORCH') != '1':
            import torch
            hip_rt = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
            if os.path.exists(hip_rt):
                ctypes.CDLL(hip_rt, mode=ctypes.RTLD_GLOBAL)
        L = ctypes.CDLL(LIBPATH)
        (i64, p, u32, ci) = (ctypes.c_int64, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int)
        L.kn_abi_version.restype = ci
        L.kn_last_error.restype = ctypes.c_char_p
        L.kn_device_info.argtypes = [p, p, i64]
        L.kn_csr_create.argtypes = [i64, i64, i64, p, p, p, p]
        L.kn_csr_create_f64.argtypes = [i64, i64, i64, p, p, p, p]
        L.kn_dtype_bits.argtypes = [p, p]
        L.kn_export_csr_f64.argtypes = [p, p, p, p]
        L.kn_spmm_f64.argtypes = [p, p, i64, i64, p, i64, u32, p]
        L.kn_tiled_create.argtypes = [i64, i64, i64, p, i64, p, p, p, p, p]
        L.kn_conv2dtiled_create.argtypes = [i64, i64, p, p, i64, p, i64, p, p, p, p, p]
        L.kn_convtaps_create.argtypes = [p, p, i64, p, i64, p, p, p, p, p, p]
        L.kn_dense_create.argtypes = [i64, i64, p, p]
        L.kn_convtaps_drop_zero_entries.argtypes = [p]
        L.kn_chain_create.argtypes = [i64, p, p, p]
        L.kn_destroy.argtypes = [p]
        L.kn_nnz.argtypes = [p, p]
        L.kn_nnz_expanded.argtypes = [p, p]
        L.kn_shape.argtypes = [p, p, p]
        L.kn_export_csr.argtypes = [p, p, p, p]
        L.kn_spmm.argtypes = [p, p, i64, i64, p, i64, u32, p]
        L.kn_spmm_plan.argtypes = [p, i64, i64, i64, u32, p, i64]
        L.kn_spmm_screen.argtypes = [p, p, i64, i64, p, i64, u32, p, p]
        L.kn_spmm_planes.argtypes = [p, p, i64, i64, i64, i64, p, i64, i64, u32, p]
        L.kn_release_side_tables.argtypes = [p]
        L.kn_absmax.argtypes = [p, i64, i64, i64, p, p]
        L.kn_reserve_workspace.argtypes = [p, i64, p]
        L.kn_relu.argtypes = [p, i64, i64, i64, p]
        L.kn_affine_to_linear.argtypes = [p, i64, i64, p, i64, p]
        L.kn_linear_to_affine.argtypes = [p, i64, i64, i64, p, p, p]
        for s in SYMBOLS:
            if s not in ('kn_last_error',):
                getattr(L, s).restype = ci
        L.kn_last_error.restype = ctypes.c_char_p
        if L.kn_abi_version() != KN_ABI_VERSION:
            raise KeynetHipError('libkeynet_hip.so ABI %d != binding ABI %d: rebuild' % (L.kn_abi_version(), KN_ABI_VERSION))
        _lib = L
    return _lib


def check(rc):
    if rc != KN_OK:
        msg = lib().kn_last_error()
        raise KeynetHipError('libkeynet_hip: error %d: %s' % (rc, msg.decode() if msg else ''))


def device_info():
    n = ctypes.c_int(0)
    buf = ctypes.create_string_buffer(64)
    check(lib().kn_device_info(ctypes.byref(n), buf, 64))
    return (n.value, buf.value.decode())


def _np(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return (a, a.ctypes.data_as(ctypes.c_void_p))


class Operator(object):
    """Owns one kn_handle_t (a keyed operator resident in HBM)."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        try:
            if getattr(self, '_h', None) is not None and _lib is not None:
                _lib.kn_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @staticmethod
    def csr(shape, indptr, indices, data):
        """kn_csr_create, or kn_csr_create_f64 when `data` is a float64 array (the operator then computes in float64, as scipy does for it)."""
        (ip, ipp) = _np(indptr, np.int32)
        (ix, ixp) = _np(indices, np.int32)
        f64 = np.asarray(data).dtype == np.float64
        (dt, dtp) = _np(data, np.float64 if f64 else np.float32)
        h = ctypes.c_void_p()
        check((lib().kn_csr_create_f64 if f64 else lib().kn_csr_create)(int(shape[0]), int(shape[1]), int(len(ix)), ipp, ixp, dtp, ctypes.byref(h)))
        return Operator(h)

    @staticmethod
    def tiled(shape, blocks, tile_ptr, tile_row, tile_col, tile_val):
        (bl, blp) = _np(np.asarray(blocks).reshape(-1, 3), np.int64)
        (tp, tpp) = _np(tile_ptr, np.int64)
        (tr, trp) = _np(tile_row, np.int32)
        (tc, tcp) = _np(tile_col, np.int32)
        (tv, tvp) = _np(tile_val, np.float32)
        h = ctypes.c_void_p()
        check(lib().kn_tiled_create(int(shape[0]), int(shape[1]), int(len(bl)), blp, int(len(tp) - 1), tpp, trp, tcp, tvp, ctypes.byref(h)))
        return Operator(h)

    @staticmethod
    def conv2dtiled(shape, inshape, outshape, blocks, tile_keys, tile_isbias, tile_chan, tile_bias):
        (bl, blp) = _np(np.asarray(blocks).reshape(-1, 3), np.int64)
        (tk, tkp) = _np(np.asarray(tile_keys).reshape(-1, 3), np.int64)
        (ib, ibp) = _np(tile_isbias, np.uint8)
        (ch, chp) = _np(tile_chan, np.float32)
        (bs, bsp) = _np(tile_bias, np.float32)
        (ins, insp) = _np(inshape, np.int64)
        (outs, outsp) = _np(outshape, np.int64)
        h = ctypes.c_void_p()
        check(lib().kn_conv2dtiled_create(int(shape[0]), int(shape[1]), insp, outsp, int(len(bl)), blp, int(len(tk)), tkp, ibp, chp, bsp, ctypes.byref(h)))
        return Operator(h)

    @staticmethod
    def convtaps(inshape, outshape, taps, ent_out, ent_in, ent_tap, ent_coef, lastcol):
        (ins, insp) = _np(inshape, np.int64)
        (outs, outsp) = _np(outshape, np.int64)
        (tp, tpp) = _np(taps, np.float32)
        assert tp.ndim == 3 and tp.shape[1] == outs[0] and tp.shape[2] == ins[0], 'taps must be [ntaps, Cout, Cin]'
        (eo, eop) = _np(ent_out, np.int32)
        (ei, eip) = _np(ent_in, np.int32)
        (et, etp) = _np(ent_tap, np.int32)
        if ent_coef is None:
            (ec, ecp) = (None, None)
        else:
            (ec, ecp) = _np(ent_coef, np.float32)
        if lastcol is None:
            (lc, lcp) = (None, None)
        else:
            (lc, lcp) = _np(lastcol, np.float32)
            assert len(lc) == int(outs[0] * outs[1] * outs[2]) + 1
        h = ctypes.c_void_p()
        check(lib().kn_convtaps_create(insp, outsp, int(tp.shape[0]), tpp, int(len(eo)), eop, eip, etp, ecp, lcp, ctypes.byref(h)))
        return Operator(h)

    def drop_zero_entries(self):
        """kn_convtaps_drop_zero_entries: zero-valued entries of the expansion are absent from the reference operator (an untiled keyed conv CSR)."""
        check(lib().kn_convtaps_drop_zero_entries(self._h))
        return self

    @staticmethod
    def dense(W):
        (w, wp) = _np(W, np.float32)
        assert w.ndim == 2
        h = ctypes.c_void_p()
        check(lib().kn_dense_create(int(w.shape[0]), int(w.shape[1]), wp, ctypes.byref(h)))
        return Operator(h)

    @staticmethod
    def chain(operators, flags):
        """kn_chain_create: the operators (CSR handles, in application order) as ONE launch; flags[l] & KN_FLAG_RELU = ReLU after operator l."""
        n = len(operators)
        hs = (ctypes.c_void_p * n)(*[op.handle for op in operators])
        fl = (ctypes.c_uint32 * n)(*[int(f) for f in flags])
        h = ctypes.c_void_p()
        check(lib().kn_chain_create(n, hs, fl, ctypes.byref(h)))
        return Operator(h)

    def nnz(self):
        n = ctypes.c_int64(0)
        check(lib().kn_nnz(self._h, ctypes.byref(n)))
        return n.value

    def nnz_expanded(self):
        n = ctypes.c_int64(0)
        check(lib().kn_nnz_expanded(self._h, ctypes.byref(n)))
        return n.value

    def shape(self):
        (r, c) = (ctypes.c_int64(0), ctypes.c_int64(0))
        check(lib().kn_shape(self._h, ctypes.byref(r), ctypes.byref(c)))
        return (r.value, c.value)

    def dtype_bits(self):
        b = ctypes.c_int(0)
        check(lib().kn_dtype_bits(self._h, ctypes.byref(b)))
        return b.value

    def export_csr(self):
        (rows, _) = self.shape()
        n = self.nnz_expanded()
        f64 = self.dtype_bits() == 64
        indptr = np.zeros(rows + 1, dtype=np.int32)
        indices = np.zeros(max(n, 1), dtype=np.int32)
        data = np.zeros(max(n, 1), dtype=np.float64 if f64 else np.float32)
        check((lib().kn_export_csr_f64 if f64 else lib().kn_export_csr)(self._h, indptr.ctypes.data_as(ctypes.c_void_p), indices.ctypes.data_as(ctypes.c_void_p),
                                                                        data.ctypes.data_as(ctypes.c_void_p)))
        return (indptr, indices[:n], data[:n])

    def spmm_f64(self, x_ptr, ldx, n_vecs, y_ptr, ldy, flags, stream):
        """kn_spmm_f64: a float64 operator's product as the float64 block the reference returns."""
        check(lib().kn_spmm_f64(self._h, x_ptr, int(ldx), int(n_vecs), y_ptr, int(ldy), int(flags), stream))

    def spmm(self, x_ptr, ldx, n_vecs, y_ptr, ldy, flags, stream, absmax_ptr=None):
        """kn_spmm, or kn_spmm_screen when `absmax_ptr` (device f32 raised to max |Y|) is given."""
        if absmax_ptr:
            check(lib().kn_spmm_screen(self._h, x_ptr, int(ldx), int(n_vecs), y_ptr, int(ldy), int(flags), absmax_ptr, stream))
        else:
            check(lib().kn_spmm(self._h, x_ptr, int(ldx), int(n_vecs), y_ptr, int(ldy), int(flags), stream))

    def reserve_workspace(self, n_vecs, stream):
        check(lib().kn_reserve_workspace(self._h, int(n_vecs), stream))

    def spmm_planes(self, x_ptr, ldx, x_plane_stride, n_planes, n_vecs, y_ptr, ldy, y_plane_stride, flags, stream):
        """kn_spmm_planes: this CSR operator on n_planes activation blocks in one launch per kernel; False when the operator needs the per-plane loop (KN_ERR_UNSUPPORTED)."""
        rc = lib().kn_spmm_planes(self._h, x_ptr, int(ldx), int(x_plane_stride), int(n_planes), int(n_vecs), y_ptr, int(ldy), int(y_plane_stride), int(flags), stream)
        if rc == 6:                                          # KN_ERR_UNSUPPORTED
            return False
        check(rc)
        return True

    def release_side_tables(self):
        """kn_release_side_tables: free what the operator built lazily (filled-in slot records, bf16 planes); rebuilt at next use."""
        check(lib().kn_release_side_tables(self._h))

    def plan(self, n_vecs, flags=0, ldx=None, ldy=None):
        """The kernels kn_spmm would launch for this batch width / flags (kn_spmm_plan): '; '-separated descriptions."""
        buf = ctypes.create_string_buffer(1024)
        check(lib().kn_spmm_plan(self._h, int(n_vecs), int(n_vecs if ldx is None else ldx), int(n_vecs if ldy is None else ldy), int(flags), buf, 1024))
        return buf.value.decode()


def absmax(x_ptr, rows, ld, n_vecs, out_ptr, stream):
    check(lib().kn_absmax(x_ptr, int(rows), int(ld), int(n_vecs), out_ptr, stream))


def relu(y_ptr, rows, ld, n_vecs, stream):
    check(lib().kn_relu(y_ptr, int(rows), int(ld), int(n_vecs), stream))


def affine_to_linear(x_ptr, n, d, out_ptr, ldo, stream):
    check(lib().kn_affine_to_linear(x_ptr, int(n), int(d), out_ptr, int(ldo), stream))


def linear_to_affine(y_ptr, ldy, n, d, out_ptr, maxdev_ptr, stream):
    check(lib().kn_linear_to_affine(y_ptr, int(ldy), int(n), int(d), out_ptr, maxdev_ptr, stream))
