"""Neutral on-disk container for keyed networks (.npz of plain arrays), replacing the reference's whole-object pickles
(test/test_keynet.py:106, demo/challenge.ipynb cell 1), which need the reference's classes to load.

Schema (also the schema of tests/golden/*.npz, written by tests/golden/make_golden.py from the reference's objects):
    layer_names                      nn.Sequential order
    L.<name>.kind                    'relu' | 'csr' (+ optional fact_* arrays: the factored device form of an untiled conv) | 'tiled' | 'diagtiled' | 'conv2dtiled' | 'convtaps'
    L.<name>.layertype               str(type(module)) of the source layer ('ReLU' in it => keyed ReLU)
    csr:          shape, indptr, indices, data           (STORED order; data float32, or float64 for an operator the reference computes in float64)
    tiled:        shape, tileshape, blocks, tile_shapes, tile_ptr, tile_row, tile_col, tile_val
    conv2dtiled:  shape, inshape, outshape, tileshape, blocks, tile_keys, tile_isbias, tile_chan, tile_bias
    convtaps:     inshape, outshape, taps, ent_out, ent_in, ent_tap, ent_coef, lastcol [, tileshape]
    L.<name>.exact                   (save_keynet only) the layer's arithmetic contract IN FORCE when it was saved, as a string: 'exact' = the
                                     reference's accumulation order and rounding, 'mfma' = f32 matrix cores, 'bf16x3' = f32 products emulated
                                     on the bf16 matrix pipe, 'split' = a filled-in conv applied as spatial mixing then channel mixing, 'auto' = not decided yet (float-key tolerance 1e-5, decided at the first forward);
                                     absent (golden files) = the default.  (Older archives hold a bool, or the string 'auto'.)
    L.<name>.exact_decl              (save_keynet only) the contract the layer was DECLARED with (what exact_mode(None) returns to)
    L.<name>.contract_record         (save_keynet only) JSON of the calibration record behind a decided 'auto' layer (measured difference,
                                     tolerance, the max |x| the decision covers): a loaded key-net keeps re-screening against it
    outshape                         (C,1,1) of the logits
    sensor.*                         (optional) the image key pair as stored-order CSR + 'sensor.inshape'

tests/golden/import_pickle.py converts a reference pickle (a keyed model saved with vipy.util.save / pickle, e.g.
demo/keynet_challenge_lenet_10AUG20.pkl) into this container; it needs the reference importable and therefore runs in the build
container only.
"""
from collections import OrderedDict
import json
import numpy as np
import scipy.sparse
from torch import nn

from . import sparse as ksp
from .layer import KeyedLayer, CONTRACTS, contract_name, _contract
from .system import KeyedModel, KeyedSensor


def operator_from_arrays(z, p):
    kind = str(z[p + 'kind'])
    if kind == 'convtaps':
        g = (lambda k: z[p + k] if (p + k) in z.files and z[p + k].size > 0 else None)
        ts = tuple(int(v) for v in z[p + 'tileshape']) if (p + 'tileshape') in z.files else None
        return ksp.Conv2dTiledMatrix.fromtaps(tuple(int(v) for v in z[p + 'inshape']), tuple(int(v) for v in z[p + 'outshape']), z[p + 'taps'],
                                              z[p + 'ent_out'], z[p + 'ent_in'], z[p + 'ent_tap'], g('ent_coef'), g('lastcol'), tileshape=ts)
    shape = tuple(int(v) for v in z[p + 'shape'])
    if kind == 'csr':
        data = z[p + 'data']
        # the values keep the dtype the reference computes in: a float64 operator (the public challenge key-net) stays float64
        M = scipy.sparse.csr_matrix((data.astype(ksp._compute_dtype(data.dtype), copy=False), z[p + 'indices'], z[p + 'indptr']), shape=shape)
        if (p + 'fact_taps') in z.files:
            # saved from a FactoredSparseMatrix: the factored device form travels with the CSR and is PROVEN again against it on load (a file edited in
            # between simply loads as the plain CSR container)
            g = (lambda k: z[p + 'fact_' + k] if (p + 'fact_' + k) in z.files and z[p + 'fact_' + k].size > 0 else None)
            F = ksp.Conv2dTiledMatrix.fromtaps(tuple(int(v) for v in z[p + 'fact_inshape']), tuple(int(v) for v in z[p + 'fact_outshape']), z[p + 'fact_taps'],
                                               z[p + 'fact_ent_out'], z[p + 'fact_ent_in'], z[p + 'fact_ent_tap'], g('ent_coef'), g('lastcol'))
            if ksp.FactoredSparseMatrix.proven(M, F):
                return ksp.FactoredSparseMatrix(M, F)
        return ksp.SparseMatrix(M)
    if kind in ('tiled', 'diagtiled'):
        W = ksp.TiledMatrix.__new__(ksp.TiledMatrix)
        (W.shape, W.dtype, W.ndim, W._op) = (shape, np.float32, 2, None)
        W._tileshape = tuple(int(v) for v in z[p + 'tileshape'])
        W._blocks = [tuple(int(v) for v in b) for b in z[p + 'blocks']]
        ptr = z[p + 'tile_ptr']
        W._tiles = [scipy.sparse.coo_matrix((z[p + 'tile_val'][ptr[k]:ptr[k + 1]], (z[p + 'tile_row'][ptr[k]:ptr[k + 1]], z[p + 'tile_col'][ptr[k]:ptr[k + 1]])),
                                            shape=tuple(int(v) for v in z[p + 'tile_shapes'][k])) for k in range(len(ptr) - 1)]
        return W
    if kind == 'conv2dtiled':
        W = ksp.Conv2dTiledMatrix.__new__(ksp.Conv2dTiledMatrix)
        (W.shape, W.dtype, W.ndim, W._op, W._taps) = (shape, np.float32, 2, None, None)
        W._inshape = tuple(int(v) for v in z[p + 'inshape'])
        W._outshape = tuple(int(v) for v in z[p + 'outshape'])
        W._tileshape = tuple(int(v) for v in z[p + 'tileshape'])
        W._blocks = [tuple(int(v) for v in b) for b in z[p + 'blocks']]
        W._tiles = OrderedDict()
        (nc, nb) = (0, 0)
        for (key, isb) in zip(z[p + 'tile_keys'], z[p + 'tile_isbias']):
            if isb:
                W._tiles[tuple(int(v) for v in key)] = np.array(z[p + 'tile_bias'][nb], dtype=np.float32).reshape(1, 1)
                nb += 1
            else:
                W._tiles[tuple(int(v) for v in key)] = np.asarray(z[p + 'tile_chan'][nc], dtype=np.float32)
                nc += 1
        return W
    raise ValueError('unknown operator kind "%s"' % kind)


def operator_to_arrays(W, p, out):
    if isinstance(W, ksp.Conv2dTiledMatrix):
        if W._taps is not None:
            out[p + 'kind'] = np.array('convtaps')
            (out[p + 'inshape'], out[p + 'outshape']) = (np.array(W._inshape, dtype=np.int64), np.array(W._outshape, dtype=np.int64))
            for (k, v) in W._taps.items():
                out[p + k] = v if v is not None else np.zeros(0, np.float32)
            if W._tileshape is not None:
                out[p + 'tileshape'] = np.array(W._tileshape, dtype=np.int64)
            return
        out[p + 'kind'] = np.array('conv2dtiled')
        (bl, tk, ib, ch, bs) = W._golden_arrays()
        for (k, v) in (('shape', np.array(W.shape, dtype=np.int64)), ('inshape', np.array(W._inshape, dtype=np.int64)), ('outshape', np.array(W._outshape, dtype=np.int64)),
                       ('tileshape', np.array(W._tileshape, dtype=np.int64)), ('blocks', bl), ('tile_keys', tk), ('tile_isbias', ib.astype(bool)), ('tile_chan', ch), ('tile_bias', bs)):
            out[p + k] = v
    elif isinstance(W, ksp.TiledMatrix):
        out[p + 'kind'] = np.array('tiled')
        (ptr, tr, tc, tv) = W._tile_arrays()
        for (k, v) in (('shape', np.array(W.shape, dtype=np.int64)), ('tileshape', np.array(W._tileshape, dtype=np.int64)), ('blocks', np.array(list(W), dtype=np.int64).reshape(-1, 3)),
                       ('tile_shapes', np.array([t.shape for t in W._tiles], dtype=np.int64).reshape(-1, 2)), ('tile_ptr', ptr), ('tile_row', tr), ('tile_col', tc), ('tile_val', tv)):
            out[p + k] = v
    else:
        out[p + 'kind'] = np.array('csr')
        (ip, ix, dt) = ksp._stored_order_csr(W._matrix if ksp.is_scipy_sparse(W._matrix) else scipy.sparse.csr_matrix(W._matrix))
        for (k, v) in (('shape', np.array(W.shape, dtype=np.int64)), ('indptr', ip), ('indices', ix), ('data', dt)):
            out[p + k] = v
        if isinstance(W, ksp.FactoredSparseMatrix):                  # + the factored device form (a few MB next to the CSR's hundreds)
            F = W._factored
            (out[p + 'fact_inshape'], out[p + 'fact_outshape']) = (np.array(F._inshape, dtype=np.int64), np.array(F._outshape, dtype=np.int64))
            for (k, v) in F._taps.items():
                out[p + 'fact_' + k] = v if v is not None else np.zeros(0, np.float32)


def _contract_from_array(a, what):
    """'exact' / 'mfma' / 'auto' / 'bf16x3' / 'split' (or a bool of an older archive) -> True / False / 'auto' / 'bf16x3' / 'split'; anything else is refused."""
    if a.dtype.kind in 'US':
        v = str(a)
        if v not in CONTRACTS:
            raise ValueError('%s: unknown arithmetic contract "%s" (expected one of %s)' % (what, v, ', '.join(CONTRACTS)))
        return _contract(v, True)
    if a.dtype.kind == 'b':
        return bool(a)
    raise ValueError('%s: arithmetic contract must be a string or a bool, got dtype %s' % (what, a.dtype))


def keynet_from_arrays(z, recalibrate=False):
    """KeyedModel (public: no keys) from a neutral archive / golden file.  A layer saved with a DECIDED float-key contract comes back with
    that decision and its calibration record (so replicas that load one file run the same kernels, and every forward keeps re-screening
    against the recorded max |x|); `recalibrate=True` returns such layers to their declared contract instead ('auto': decided again on the
    first batch seen here)."""
    layers = OrderedDict()
    for name in [str(n) for n in z['layer_names']]:
        p = 'L.%s.' % name
        if str(z[p + 'kind']) == 'relu':
            layers[name] = nn.ReLU()
        else:
            exact = _contract_from_array(z[p + 'exact'], p + 'exact') if (p + 'exact') in z.files else None
            decl = _contract_from_array(z[p + 'exact_decl'], p + 'exact_decl') if (p + 'exact_decl') in z.files else exact
            c = KeyedLayer.fromoperator(operator_from_arrays(z, p), str(z[p + 'layertype']), exact=decl if recalibrate else exact)
            if (p + 'exact') in z.files:
                c._exact_decl = decl
                if not recalibrate and (p + 'contract_record') in z.files:
                    c._contract_record = json.loads(str(z[p + 'contract_record']))
            layers[name] = c
    last = [l for l in layers.values() if isinstance(l, KeyedLayer)][-1]
    outshape = tuple(int(v) for v in z['outshape']) if 'outshape' in z.files else (last.W.shape[0] - 1, 1, 1)
    return KeyedModel.fromlayers(layers, outshape)


def sensor_from_arrays(z, inshape=None):
    """KeyedSensor from the 'sensor.*' arrays of an archive (inshape from 'sensor.inshape' unless given)."""
    inshape = tuple(int(v) for v in z['sensor.inshape']) if inshape is None else inshape
    shape = tuple(int(v) for v in z['sensor.shape'])
    enc = scipy.sparse.csr_matrix((z['sensor.enc.data'], z['sensor.enc.indices'], z['sensor.enc.indptr']), shape=shape)
    dec = scipy.sparse.csr_matrix((z['sensor.dec.data'], z['sensor.dec.indices'], z['sensor.dec.indptr']), shape=shape)
    return KeyedSensor(tuple(inshape), (enc, dec))


def save_keynet(knet, filename, sensor=None, compress=True):
    """`compress=False`: a plain (stored) archive -- seconds instead of minutes for a GB-sized key-net handed to other processes on one node."""
    out = {'layer_names': np.array([n for (n, _) in knet._keynet.named_children()]), 'outshape': np.array(knet._outshape, dtype=np.int64)}
    for (name, c) in knet._keynet.named_children():
        p = 'L.%s.' % name
        if isinstance(c, KeyedLayer):
            operator_to_arrays(c.W, p, out)
            out[p + 'layertype'] = np.array(c._layertype)
            # the contract in force (a decided 'auto' layer is saved WITH its decision and the evidence: every loader then runs the same
            # kernels -- load_keynet(recalibrate=True) decides again instead) and the declared one
            out[p + 'exact'] = np.array(contract_name(getattr(c, '_exact', True)))
            out[p + 'exact_decl'] = np.array(contract_name(getattr(c, '_exact_decl', getattr(c, '_exact', True))))
            rec = getattr(c, '_contract_record', None)
            if rec is not None:
                out[p + 'contract_record'] = np.array(json.dumps(rec))
        else:
            out[p + 'kind'] = np.array('relu')
    if sensor is not None:
        for (tag, M) in (('enc', sensor._encryptkey), ('dec', sensor._decryptkey)):
            (ip, ix, dt) = ksp._stored_order_csr(M.tocsr() if M.format not in ('csr', 'coo', 'csc') else M)
            (out['sensor.%s.indptr' % tag], out['sensor.%s.indices' % tag], out['sensor.%s.data' % tag]) = (ip, ix, dt)
        out['sensor.shape'] = np.array(sensor._encryptkey.shape, dtype=np.int64)
        out['sensor.inshape'] = np.array(sensor._inshape[1:], dtype=np.int64)
    (np.savez_compressed if compress else np.savez)(filename, **out)
    return filename


def load_keynet(filename, with_sensor=False, recalibrate=False):
    """KeyedModel from an archive written by save_keynet (or by tests/golden/import_pickle.py); `with_sensor` also returns the
    KeyedSensor when the archive holds the image keys: (sensor, model) like the reference's factories.  `recalibrate`: see
    keynet_from_arrays."""
    z = np.load(filename, allow_pickle=False)
    knet = keynet_from_arrays(z, recalibrate=recalibrate)
    if not with_sensor:
        return knet
    return (sensor_from_arrays(z) if 'sensor.shape' in z.files and 'sensor.inshape' in z.files else None, knet)
