"""KeyedLayer: one keyed linear layer  W_hat = A . W . A_prev^-1  of a key-net (mirror of keynet/layer.py:15-106).

Construction (host, offline) restates the reference's keying; `forward` -- the drop-in boundary -- hands the activation
block to the HIP operator stored in `self.W` (its torchdot), optionally fusing the ReLU that follows.
"""
import logging
import numpy as np
import scipy.sparse
import torch
from torch import nn
import torch.nn.functional as F

from .globals import verbose
from . import sparse as ksp
from . import direct as kdirect
from .sparse import SparseMatrix, sparse_toeplitz_conv2d, sparse_toeplitz_avgpool2d
from . import _capi
from .torch import affine_to_linear_matrix


def _square(v, what):
    """kernel_size / stride given as int or (a, a) -> a."""
    if isinstance(v, int):
        return v
    assert len(v) in (1, 2) and v[0] == v[-1], '%s must be square / isotropic, got %s' % (what, str(v))
    return v[0]


def _sandwich(A, W, Ainv):
    """A . W . Ainv with the reference's association ((A.W).Ainv, keynet/layer.py:35): scipy's SpGEMM leaves each row in
    an order that depends on it, and that stored order is part of the bit-exact contract.  A is None for the last
    layer of a key-net without output encryption."""
    return W.dot(Ainv) if A is None else A.dot(W).dot(Ainv)


def _conv_operator(m, inshape, outshape, A, Ainv, tileshape, direct):
    k = _square(m.kernel_size, 'kernel')
    stride = _square(m.stride, 'stride')
    assert len(inshape) == 3, 'inshape is the (C,H,W) of the tensor entering this layer'
    assert m.padding[0] == k // 2 and m.padding[-1] == k // 2, "only 'same' padding ((k-1)/2) is keyable"
    (w, b) = (m.weight.detach().numpy(), m.bias.detach().numpy())
    if direct is None:
        direct = tileshape is not None and kdirect.toeplitz_entries('conv', inshape, outshape, k) > KeyedLayer.DIRECT_THRESHOLD
    if direct:
        return ksp.Conv2dTiledMatrix.fromtaps(tileshape=tileshape, **kdirect.keyed_conv_taps(w, b, inshape, outshape, stride, A, Ainv))
    W = _sandwich(A, sparse_toeplitz_conv2d(inshape, w, bias=b, stride=stride), Ainv)
    if tileshape is None:
        # An untiled keyed conv whose stored CSR is, provably, the ascending-column expansion of its factored form (identity / channel-replicated
        # permutation keys on both sides: every conv layer of PermutationKeynet AllConvNet but the first) is handed to the device in factored form:
        # same bits (the proof is entry for entry, order included), 0.3 MB of taps instead of the CSR's hundreds of MB.  See FactoredSparseMatrix.
        if W.nnz >= KeyedLayer.FACTOR_UNTILED_MIN_NNZ:
            try:
                F = ksp.Conv2dTiledMatrix.fromtaps(tileshape=None, **kdirect.keyed_conv_taps(w, b, inshape, outshape, stride, A, Ainv))
                if ksp.FactoredSparseMatrix.proven(W.tocsr() if W.format != 'csr' else W, F):
                    return ksp.FactoredSparseMatrix(W, F)
            except (ValueError, AssertionError):
                pass                                    # keys that are not channel-replicated (a global permutation): the CSR it is
        return W
    return ksp.Conv2dTiledMatrix(W, inshape, outshape, tileshape, bias=True, sanitycheck=False)


def _pool_operator(m, inshape, outshape, A, Ainv, tileshape, direct):
    k = _square(m.kernel_size, 'kernel')
    stride = _square(m.stride, 'stride')
    assert len(inshape) == 3, 'inshape is the (C,H,W) of the tensor entering this layer'
    C = inshape[0]
    if direct is None:
        direct = tileshape is not None and kdirect.toeplitz_entries('pool', inshape, (C,) + tuple(outshape[1:]), k) > KeyedLayer.DIRECT_THRESHOLD
    if direct:
        W = kdirect.keyed_avgpool_csr(C, (inshape[1], inshape[2]), k, stride, A, Ainv)
    else:
        W = _sandwich(A, sparse_toeplitz_avgpool2d(inshape, (C, C, k, k), stride), Ainv)
    return W if tileshape is None else ksp.TiledMatrix(W, tileshape)


def _linear_operator(m, A, Ainv):
    dense = affine_to_linear_matrix(m.weight, m.bias).detach().numpy()      # (in+1, out+1), left-multiplying
    return _sandwich(A, scipy.sparse.coo_matrix(dense).transpose(), Ainv)


def _contract(exact, bit_exact_default):
    """Arithmetic contract of a layer: True (the reference's order and rounding), False (matrix cores, float-key tolerance) or 'auto'
    (decided at the first forward and re-screened on every later one, KeyedLayer._calibrate / rescreen).  None = the default: True for
    untiled layers and for layers keyed by permutations only (north_star: "bit-exact for the permutation-only key"), 'auto' for tiled
    layers whose keys carry float coefficients ("within 1e-5 for float keyed layers").  Strings 'exact' / 'mfma' name True / False."""
    if exact is None:
        return True if bit_exact_default else 'auto'
    if isinstance(exact, str):
        assert exact in CONTRACTS, "exact must be True, False, None or one of %s" % str(CONTRACTS)
        return {'exact': True, 'mfma': False}.get(exact, exact)
    return bool(exact)


CONTRACTS = ('exact', 'mfma', 'auto', 'bf16x3', 'split')


def contract_name(c):
    """True / False / 'auto' / 'bf16x3' / 'split' -> 'exact' / 'mfma' / 'auto' / 'bf16x3' / 'split' (the on-disk and reporting vocabulary)."""
    return c if isinstance(c, str) else ('exact' if c else 'mfma')


def _is_permutation_key(M):
    """Is this key (scipy sparse, or None = no key on that side) a permutation matrix: one entry per row and column, every value
    exactly 1?  Such keys only move entries around: the keyed operator holds the source weights themselves, nothing is scaled or mixed."""
    if M is None:
        return True
    if not scipy.sparse.issparse(M) or M.shape[0] != M.shape[1] or M.nnz != M.shape[0]:
        return False
    C = M.tocoo()
    n = M.shape[0]
    return bool(np.all(C.data == 1) and len(np.unique(C.row)) == n and len(np.unique(C.col)) == n)


FLOAT_KEY_TOL = 1e-5          # BASELINE north_star: "within 1e-5 for float keyed layers"
# ... in the form the reference's own tests assert it (test/test_keynet.py:33,196,218: np.allclose(a, b, atol=1e-5), rtol at numpy's default):
# ELEMENT-WISE |a - b| <= atol + rtol |b|.  A small element next to large ones gets no slack from them.
FLOAT_KEY_ATOL = 1e-5
FLOAT_KEY_RTOL = 1e-5
EPS32 = float(np.finfo(np.float32).eps)


def gate(y, ref):
    """The float-key criterion as a number: max over elements of |y - ref| / (atol + rtol |ref|); <= 1 is np.allclose(y, ref, atol=1e-5)
    (numpy's default rtol).  Returns (ratio, |y - ref| at the worst element, its tolerance, max |y - ref|); NaN when a difference is not
    finite.  Device tensors; one host read."""
    d = (y - ref).abs()
    t = FLOAT_KEY_ATOL + FLOAT_KEY_RTOL * ref.abs()
    r = d / t
    r = torch.where(torch.isnan(r), torch.full_like(r, float('inf')), r)        # NaN (Inf - Inf, NaN on one side): never inside any tolerance
    k = torch.argmax(r)
    (rk, dk, tk, dmax) = torch.stack((r.flatten()[k], d.flatten()[k], t.flatten()[k], d.max())).tolist()      # the four scalars in ONE device-to-host read
    return (rk if np.isfinite(rk) else float('nan'), dk, tk, dmax)


_log = logging.getLogger('keynet_amd')


def _absmax_into(yt, slot):
    """Raise the one-element device tensor `slot` to max |yt| (kn_absmax: one pass over a feature-major block)."""
    yt = yt if yt.is_contiguous() else yt.contiguous()
    with torch.cuda.device(yt.device):
        _capi.absmax(yt.data_ptr(), yt.shape[0], yt.shape[1], yt.shape[1], slot.data_ptr(), torch.cuda.current_stream().cuda_stream)


class KeyedLayer(nn.Module):
    """One keyed layer of a key-net: holds W_hat = A . W . A_prev^-1 as an HBM-resident operator and applies it."""

    DIRECT_THRESHOLD = 20000000   # Toeplitz entries above which tiled conv/pool layers are keyed in factored form
    FACTOR_UNTILED_MIN_NNZ = 1000000     # untiled conv layers at least this large are checked for a factored device form (FactoredSparseMatrix)

    def __init__(self, module, inshape, outshape, A, Ainv, tileshape=None, direct=None, exact=None):
        """module: the source nn.Conv2d / nn.AvgPool2d / nn.Linear / nn.ReLU; A: this layer's output key (None = leave the
        output unkeyed); Ainv: inverse of the key on its input; tileshape: store conv/pool operators tiled.
        `direct`: None = automatic (factored, Toeplitz-free keying for tiled conv/avgpool layers whose Toeplitz matrix
        would exceed DIRECT_THRESHOLD entries -- the reference route cannot build those at all); True / False force it.
        `exact`: True = every product in the reference's accumulation order and rounding (bit-exact with scipy);
        False = conv-taps and large dense operators run on the matrix cores, whatever the error; 'auto' = the matrix cores where a
        calibration on the first batch shows the result inside the float-key tolerance (element-wise 1e-5 + 1e-5 |ref|), else exact.
        None (default) = exact for untiled layers AND for tiled layers keyed by permutations only (north_star: "bit-exact for the
        permutation-only key"), 'auto' for tiled layers whose keys carry float coefficients.
        Raises ValueError for layer types that cannot be keyed (the reference's behaviour, keynet/layer.py:72-79)."""
        super(KeyedLayer, self).__init__()
        self._layertype = str(type(module))
        (self._inshape, self._outshape, self._tileshape) = (inshape, outshape, tileshape)
        # default contract: bit-exact unless the layer is tiled AND one of its keys carries float coefficients
        self._exact = self._exact_decl = _contract(exact, tileshape is None or exact is not None or (_is_permutation_key(A) and _is_permutation_key(Ainv)))
        if isinstance(module, nn.Conv2d):
            self._repr = 'Conv2d %d->%d, k=%s, s=%s' % (module.in_channels, module.out_channels, str(module.kernel_size), str(module.stride))
            W = _conv_operator(module, inshape, outshape, A, Ainv, tileshape, direct)
        elif isinstance(module, nn.AvgPool2d):
            self._repr = 'AvgPool2d k=%s, s=%s' % (str(module.kernel_size), str(module.stride))
            W = _pool_operator(module, inshape, outshape, A, Ainv, tileshape, direct)
        elif isinstance(module, nn.Linear):
            self._repr = 'Linear %d->%d' % (module.in_features, module.out_features)
            W = _linear_operator(module, A, Ainv)
        elif isinstance(module, nn.ReLU):
            self._repr = 'ReLU'                       # a keyed ReLU: the key change alone, ReLU applied in forward
            W = A.dot(Ainv)
        elif isinstance(module, nn.BatchNorm2d):
            raise ValueError('a BatchNorm2d is folded into the layer before it: name it "<layer>_bn" and place it right after "<layer>"')
        elif isinstance(module, nn.Dropout):
            raise ValueError('Dropout is the identity at inference: it is skipped during keying, never keyed')
        else:
            raise ValueError('unsupported layer type "%s"' % str(type(module)))
        self.W = W if isinstance(W, SparseMatrix) else SparseMatrix(W)

    @classmethod
    def fromoperator(cls, W, layertype, inshape=None, outshape=None, repr_=None, exact=None):
        """Wrap an already keyed operator (a public key-net loaded from a neutral file, a fixture, a direct build)."""
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        # default contract of a loaded operator: bit-exact unless it is a conv operator with float coefficients (a factored operator without
        # coefficient entries was keyed by permutations; a block/tile description does not say, so it gets the float-key contract)
        coef_free = isinstance(W, ksp.Conv2dTiledMatrix) and W._taps is not None and W._taps['ent_coef'] is None
        self._exact = self._exact_decl = _contract(exact, coef_free or not isinstance(W, ksp.Conv2dTiledMatrix))
        (self._layertype, self._tileshape, self._inshape, self._outshape) = (layertype, None, inshape, outshape)
        self._repr = repr_ if repr_ is not None else layertype
        self.W = W if isinstance(W, SparseMatrix) else SparseMatrix(W)
        return self

    def extra_repr(self):
        return str('<%s, backend=hip, shape=%s, nnz=%d>' % (self._repr, str(self.W.shape), self.nnz()))

    def iskeyedrelu(self):
        return 'ReLU' in self._layertype

    def forward(self, x_affine, fuse_relu=False, absmax=None):
        """[N, Din+1] -> [N, Dout+1] (keynet/layer.py:88-93).  The result is a transposed view of the feature-major
        [Dout+1, N] block the kernel wrote, so the next layer's x.t() is free.  `fuse_relu` folds the unkeyed nn.ReLU
        that follows this layer in the key-net (keynet/system.py:92) into the kernel epilogue.  `absmax`: a one-element device
        f32 tensor raised to max |y| of this call (kn_spmm_screen; KeyedModel.forward_linear re-screens the next layer's contract with it).
        A layer whose contract is still 'auto' decides it here, on this batch (blocking host reads, an extra order-preserving launch:
        not capturable into a HIP graph -- KeyedModel.capture runs an eager forward first)."""
        if verbose():
            print('[keynet_amd.layer]: forward %s' % str(self))
        exact = getattr(self, '_exact', True)
        if exact == 'auto':
            if x_affine.is_cuda and torch.cuda.is_current_stream_capturing():
                raise _capi.KeynetHipError('keynet_amd: %s has not decided its arithmetic contract yet (exact=\'auto\' calibrates on the first batch, with host reads): '
                                           'run one eager forward before capturing a HIP graph (KeyedModel.capture does), or declare exact=True / False' % self._repr)
            y = self._calibrate(x_affine, fuse_relu or self.iskeyedrelu())
            if absmax is not None:
                _absmax_into(y.t(), absmax)
            return y
        if exact == 'split' and not (isinstance(self.W, ksp.Conv2dTiledMatrix) and self.W._taps is not None):
            exact = False                                         # (forced with exact_mode('split') on an operator that has no split form: the matrix cores)
        y = self.W.torchdot(x_affine.t(), relu=(fuse_relu or self.iskeyedrelu()), exact=exact, absmax=absmax).t()
        return y

    # -- float-key contract: decided by calibration, re-screened on every forward --------------------------------------
    RESCREEN_FACTOR = 2.0        # a layer is re-calibrated when max |x| exceeds the calibrated value by more than this factor
    ALLOW_SPLIT = True           # calibration offers the split application to filled-in conv operators (set False to put the fused kernels side by side with it)

    def screened(self):
        """Runs this layer on a re-ordering kernel (matrix cores) BY A CALIBRATION DECISION?  Then every forward must check that the
        decision still covers its input (a layer forced there with exact_mode(False) / 'bf16x3' is the caller's responsibility)."""
        rec = getattr(self, '_contract_record', None)
        return getattr(self, '_exact', True) in (False, 'bf16x3', 'split') and rec is not None and rec.get('max_abs_x') is not None

    def rescreen(self, xmax):
        """max |x| of a later batch against the calibrated one: True = the decision does not cover this batch (re-calibrate).  The measured
        difference of a re-ordered f32 sum scales with the activations while the tolerance 1e-5 max(1, |y|) has a floor, so a decision
        taken with 2x headroom (accepted at <= 0.5 tol) is kept for inputs up to RESCREEN_FACTOR x the calibrated magnitude."""
        cal = float(self._contract_record['max_abs_x'])
        if not np.isfinite(cal):
            return False                                          # calibrated on non-finite activations: there is no larger batch to learn from
        return not (xmax <= self.RESCREEN_FACTOR * cal)          # also True for NaN

    def mfma_capable(self, device=None):
        """Does tolerance mode run this layer on the matrix cores at all (conv-taps operator, or a large dense nn.Linear)?"""
        W = self.W
        if isinstance(W, ksp.Conv2dTiledMatrix):
            return True
        return type(W) is SparseMatrix and torch.cuda.is_available() and W._dense_device_op(device) is not None

    def _calibrate(self, x_affine, relu):
        """First forward of a layer whose contract is 'auto': decide ONCE, on this batch, between the matrix cores and the
        order-preserving kernels, so that the float-key tolerance holds in the reference's own form -- element-wise
        |y_mfma - y_reference| <= 1e-5 + 1e-5 |y_reference| (np.allclose(atol=1e-5): test/test_keynet.py:33,196,218), unconditioned.

        An f32 evaluation of sum_j a_j x_j in another order than the reference's differs from it by about 2 eps32 sum|a_j x_j|.  With keys
        that carry large coefficients (TiledOrthogonalKeynet: gamma = 100 bias keys) that is 1e-4 on unit-scale outputs -- the
        reference's own f32 result is that far from the exact sum too -- so such a layer cannot meet 1e-5 against scipy on ANY
        re-ordered arithmetic and must run in the reference's order.  Screen: bound = 2 eps32 (max_row sum|a|) max|x| against the
        tolerance's floor 1e-5 (operator factor from the host description, activation factor reduced on the device).  Check: the
        order-preserving kernel on up to 256 batch columns of this very input -- the window that holds the batch's largest |x| --
        compared with the matrix-core result element by element (always for conv operators -- one launch; for a dense nn.Linear only when
        the screen does not show 2x headroom, because its CSR twin has to be uploaded first).  The layer switches to exact when the worst
        element uses more than half its tolerance, or when the screen fails and the measurement does not show 4x headroom.  (The 2x
        headroom of every accepted decision is what rescreen() relies on: inputs up to RESCREEN_FACTOR x the calibrated magnitude.)"""
        W = self.W
        xt = x_affine.t()
        dev = xt.device if xt.is_cuda else None
        if xt.shape[1] == 0:
            # an empty batch (a rank whose shard of a small batch is empty) decides nothing: the layer stays 'auto' and calibrates on the first batch that holds an image
            return W.torchdot(xt, relu=relu, exact=True).t()
        rec = dict(layer=self._repr, decided='exact', reason='no matrix-core path for this operator')
        if not self.mfma_capable(dev):
            self._exact = True
            self._contract_record = rec
            return W.torchdot(xt, relu=relu, exact=True).t()
        # the measured window: up to 256 batch columns (a multiple of 128 wide when the batch allows), placed over the column with the largest |x|
        n = int(xt.shape[1])
        cols = min(n, 256)
        colmax = xt.detach().abs().amax(dim=0)
        xmax = float(colmax.max()) if n else 0.0
        c0 = 0
        if n > cols:
            c0 = min((int(torch.argmax(colmax)) // cols) * cols, n - cols)
        win = slice(c0, c0 + cols)
        # opt-in first candidate (KeyedModel.exact_mode('auto-bf16x3')): f32 products emulated on the bf16 matrix pipe (KN_FLAG_BF16X3).
        # It is taken only with 4x headroom under the tolerance on this batch; otherwise the decision below is made as usual.
        if getattr(self, '_allow_bf16x3', False) and isinstance(W, ksp.Conv2dTiledMatrix):
            xs = xt[:, win] if cols % 128 == 0 else None
            with torch.cuda.device(xt.device):
                eligible = xs is not None and 'bf16x3' in W._device_op(dev).plan(cols, _capi.KN_FLAG_BF16X3 | (_capi.KN_FLAG_RELU if relu else 0))
            if eligible:
                yb = W.torchdot(xs, relu=relu, exact='bf16x3')
                ye = W.torchdot(xs, relu=relu, exact=True)
                (ratio, meas, tol_b, dmax) = gate(yb, ye)
                ymax_b = float(ye.abs().max())
                del yb, ye
                if ratio <= 0.25:
                    self._exact = 'bf16x3'
                    self._contract_record = dict(layer=self._repr, decided='bf16x3', measured_bf16x3_vs_exact=meas, tol=tol_b, gate_ratio=ratio, max_abs_diff=dmax,
                                                 max_abs_y=ymax_b, measured_on_columns=cols, measured_from_column=c0, max_abs_x=xmax)
                    return W.torchdot(xt, relu=relu, exact='bf16x3').t()
        asum = getattr(self, '_abs_rowsum', None)
        if asum is None:
            asum = self._abs_rowsum = W.max_abs_rowsum()
        bound = 2.0 * EPS32 * asum * xmax
        # first candidate for a FILLED-IN factored conv (a key whose inverse is dense inside its blocks: 500 - 5 400 slots per output pixel): the split
        # application -- spatial mixing per tap, then channel mixing (Conv2dTiledMatrix._split_ops: 1/10 - 1/30 of the fused operator's multiply-adds).
        # Another association of the same sum: measured against the order-preserving kernel and accepted by the rule the matrix-core kernel is held to below.
        ye_win = None
        if isinstance(W, ksp.Conv2dTiledMatrix) and self.ALLOW_SPLIT and W.split_capable(n):
            ye_win = W.torchdot(xt[:, win], relu=relu, exact=True)
            try:
                ys = W.torchdot(xt, relu=relu, exact='split')
                (ratio_s, meas_s, tol_s, dmax_s) = gate(ys[:, win], ye_win)
            except (torch.cuda.OutOfMemoryError, _capi.KeynetHipError) as e:
                # the candidate needs a resident spatial CSR, a second operator and the intermediate (several ranks on one device, a smaller device): not offered then --
                # the fused kernels below need nothing beyond the operator that is already resident (round-5 advisor finding: this used to abort the first forward)
                if isinstance(e, _capi.KeynetHipError) and 'memory' not in str(e).lower():
                    raise
                _log.warning('keynet_amd: %s: the split application does not fit this device now (%s); deciding between the fused kernels', self._repr, str(e)[:120])
                W._op_split = None
                torch.cuda.empty_cache()
                (ys, ratio_s) = (None, float('inf'))
            if ratio_s <= 0.5 and (bound <= FLOAT_KEY_ATOL or ratio_s <= 0.25):
                self._exact = 'split'
                self._contract_record = dict(layer=self._repr, decided='split', max_abs_rowsum=asum, max_abs_x=xmax, max_abs_y=float(ys.abs().max()), tol=tol_s, bound=bound,
                                             measured_split_vs_exact=meas_s, gate_ratio=ratio_s, max_abs_diff=dmax_s, measured_on_columns=cols, measured_from_column=c0,
                                             fill_factor=W.fill_factor())
                # the order-preserving measurement above built the operator's slot records (16 bytes per slot: 0.4 - 4 GB per layer of the doubly-stochastic VGG-16); a layer that
                # runs split does not use them again until it is re-calibrated (round-5 advisor finding)
                with torch.cuda.device(xt.device):
                    W._device_op(dev).release_side_tables()
                return ys.t()
            del ys
        y = W.torchdot(xt, relu=relu, exact=False)
        ymax = float(y.abs().max())
        (ratio, measured, tol, dmax) = (None, None, FLOAT_KEY_ATOL, None)
        if isinstance(W, ksp.Conv2dTiledMatrix) or not (bound <= FLOAT_KEY_ATOL / self.RESCREEN_FACTOR):
            ye = ye_win if ye_win is not None else W.torchdot(xt[:, win], relu=relu, exact=True)
            (ratio, measured, tol, dmax) = gate(y[:, win].to(ye.device), ye)
            del ye
        # (a difference that is not finite -- Inf / NaN activations -- cannot be bounded: the reference's order it is)
        switch = ratio is not None and not (ratio <= 0.5 and (bound <= FLOAT_KEY_ATOL or ratio <= 0.25))
        rec = dict(layer=self._repr, decided='exact' if switch else 'mfma', max_abs_rowsum=asum, max_abs_x=xmax, max_abs_y=ymax, tol=tol, bound=bound,
                   measured_mfma_vs_exact=measured, gate_ratio=ratio, max_abs_diff=dmax, measured_on_columns=None if ratio is None else cols,
                   measured_from_column=None if ratio is None else c0)
        self._exact = bool(switch)
        self._contract_record = rec
        if switch:
            _log.warning('keynet_amd: %s runs in the reference\'s accumulation order from now on: matrix-core result off by %.3g where the tolerance is %.3g '
                         '(1e-5 + 1e-5 |y|, element-wise; bound 2 eps sum|a| max|x| = %.3g); KeyedModel.exact_mode(False) forces the matrix cores', self._repr, measured, tol, bound)
            y = W.torchdot(xt, relu=relu, exact=True)
        return y.t()

    def decrypt(self, Ainv, x_affine):
        """Apply a decryption key to this layer's output (keynet/layer.py:95-99)."""
        if scipy.sparse.issparse(Ainv):
            Ainv = SparseMatrix(Ainv)
        return Ainv.torchdot(x_affine.t()).t()

    def nnz(self):
        assert self.W is not None, 'Layer not keyed'
        return self.W.nnz()
