"""Checker legs: the device results of a bench run against the CPU oracle (oracle/: scipy csr_matvecs restated) -- test infrastructure used as the checker only,
after the timed region, never as the thing measured."""
import time

import numpy as np
import torch

from keynet_amd import sparse as ksp
from keynet_amd import system as ksys
from keynet_amd.layer import KeyedLayer
from keynet_amd.models import VGG16


def exact_parity(knet, x_cipher, n_img=8, n_pix=4, layers=('conv1_1', 'conv1_2', 'pool3_3', 'conv4_2', 'conv5_2', 'fc6')):
    """Checker for the exact leg: the order-preserving kernels AS TIMED -- launched on the whole batch -- on one real operator of each kernel family
    (first-layer conv, 64- and 512-channel conv pipelines, a keyed pooling layer = loose CSR rows, a keyed Linear = one big pattern group) against
    the CPU oracle (oracle/: scipy csr_matvecs restated) on sampled output rows, the first `n_img` batch columns, bit for bit, chained layer to
    layer with the key-net's own activations as input."""
    import oracle
    import scipy.sparse
    rng = np.random.RandomState(1)
    y = x_cipher
    checked = []
    children = list(knet._keynet.named_children())
    for (i, (name, c)) in enumerate(children):
        if not isinstance(c, KeyedLayer):
            continue
        fuse = (i + 1 < len(children)) and isinstance(children[i + 1][1], torch.nn.ReLU)
        out = c.forward(y, fuse_relu=fuse)
        if name in layers:
            W = c.W
            xh = y.t()[:, :n_img].contiguous().cpu().numpy()
            if isinstance(W, ksp.Conv2dTiledMatrix) and W._taps is not None:
                (Cout, Hout, Wout) = W._outshape
                pix = np.sort(rng.choice(Hout * Wout, size=n_pix, replace=False))
                M = W.rows_csr(pix)
                rows = (np.arange(Cout)[:, None] * Hout * Wout + pix[None, :]).ravel()
            else:
                full = W.tocsr() if isinstance(W, ksp.TiledMatrix) else W._matrix.tocsr()
                rows = np.unique(np.concatenate((rng.choice(full.shape[0] - 1, size=min(300, full.shape[0] - 1), replace=False), [full.shape[0] - 1])))
                if isinstance(W, ksp.TiledMatrix):
                    M = full[rows]
                else:                                             # stored (unsorted) order of the keyed Linear's rows, untouched
                    (ip, ix, dt) = (full.indptr, full.indices, full.data)
                    sel = np.concatenate([np.arange(ip[r], ip[r + 1]) for r in rows])
                    M = scipy.sparse.csr_matrix((dt[sel], ix[sel], np.concatenate(([0], np.cumsum(ip[rows + 1] - ip[rows])))), shape=(len(rows), full.shape[1]))
            ref = oracle.csr_matvecs(M.shape, M.indptr, M.indices, M.data.astype(np.float32), xh)
            if fuse:
                ref = np.maximum(ref, 0)
            got = out.t()[torch.as_tensor(rows, device=out.device)][:, :n_img].cpu().numpy()
            with torch.cuda.device(out.device):
                plan = W._device_op(out.device).plan(int(y.shape[0]), 2 | (1 if fuse else 0)).split(' grid=')[0] if hasattr(W, '_device_op') else ''
            checked.append({'layer': name, 'rows': int(len(rows)), 'images': n_img, 'batch_columns_launched': int(y.shape[0]), 'kernel': plan[:80],
                            'bit_equal': bool(np.array_equal(got, ref))})
        y = out
        if name == layers[-1]:
            break
    return {'check': 'exact-mode kernels, launched on the whole batch, vs the CPU oracle (scipy csr_matvecs restated) on sampled output rows of real layers', 'layers': checked,
            'ok': bool(checked) and all(r['bit_equal'] for r in checked)}


def float_key_parity(dev, batch=256):
    """Float-key family on a VGG-16 slice (the same 21-layer topology at width 8 on 32x32 inputs, keyed by TiledOrthogonalKeynet:
    hierarchical permutation + block Givens rotations + affine photometric keys, gamma = 100).  The order-preserving path is bit-exact
    with the reference's scipy arithmetic (tests/test_parity_gpu.py), so it stands in for the reference here.  Two records:
      contract   the key-net under its DEFAULT contract ('auto'): per conv layer, the shipped forward's output against the exact path on
                 the same input -- `ok` = every layer within 1e-5 * max(1, |y|), unconditioned; `layers_switched_to_exact` = the layers the
                 calibration moved off the matrix cores to get there;
      forced_mfma  the same layers forced onto the matrix cores (exact_mode(False)): how far a re-ordered f32 evaluation lands."""
    import warnings
    t0 = time.time()
    torch.manual_seed(0)
    net = VGG16(num_classes=10, width=8, fc_width=64, insize=32).eval()
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        (sensor, knet) = ksys.TiledOrthogonalKeynet((3, 32, 32), net, 8)
    g = torch.Generator(device=dev).manual_seed(77)
    x = torch.randn((batch, 3, 32, 32), generator=g, device=dev)
    xc = sensor.fromtensor(x).encrypt().astensor()
    import logging
    logging.getLogger('keynet_amd').setLevel(logging.ERROR)          # the switches are reported below, not as log lines
    la = knet.forward_linear(xc)[:, :-1]                              # calibrates every layer
    logging.getLogger('keynet_amd').setLevel(logging.WARNING)
    rep = knet.contract_report()

    def per_layer(force_mfma):
        rows = []
        y = xc
        children = list(knet._keynet.named_children())
        for (i, (name, c)) in enumerate(children):
            if not isinstance(c, KeyedLayer):
                continue
            fuse = (i + 1 < len(children)) and isinstance(children[i + 1][1], torch.nn.ReLU)
            if isinstance(c.W, ksp.Conv2dTiledMatrix):
                xt = y.t()
                ye = c.W.torchdot(xt, relu=fuse, exact=True)
                ys = c.W.torchdot(xt, relu=fuse, exact=False) if force_mfma else c.forward(y, fuse_relu=fuse).t()
                (d, m) = (float((ye - ys).abs().max()), float(ye.abs().max()))
                rows.append({'layer': name, 'max_abs_diff': d, 'max_abs_out': m, 'within_1e-5': bool(d <= 1e-5 * max(1.0, m)), 'ran': 'mfma' if (force_mfma or c._exact is False) else 'exact'})
                y = ye.t()
            else:
                y = c.forward(y, fuse_relu=fuse)
        return rows
    rows_auto = per_layer(False)
    rows_mfma = per_layer(True)
    knet.exact_mode(True)
    le = knet.forward_linear(xc)[:, :-1]
    knet.exact_mode(False)
    lm = knet.forward_linear(xc)[:, :-1]
    with torch.no_grad():
        lp = net(x.cpu()).reshape(batch, -1)
    return {'net': 'TiledOrthogonalKeynet VGG16 slice (width 8, 3x32x32, tile 8), %d images' % batch,
            'contract': {'tolerance': 1e-5, 'layers': rows_auto, 'ok': bool(all(r['within_1e-5'] for r in rows_auto)), 'layers_switched_to_exact': rep['switched'],
                         'worst_layer_abs_diff': max(r['max_abs_diff'] for r in rows_auto),
                         'logits_max_abs_diff_vs_exact': float((le - la).abs().max())},
            'ok': bool(all(r['within_1e-5'] for r in rows_auto)), 'layers_switched_to_exact': rep['switched'],
            'forced_mfma': {'layers': rows_mfma, 'worst_layer_abs_diff_mfma_vs_exact': max(r['max_abs_diff'] for r in rows_mfma),
                            'logits_max_abs_diff_mfma_vs_exact': float((le - lm).abs().max())},
            'logits_max_abs': float(le.abs().max()),
            'logits_max_abs_err_exact_vs_source_network': float((le.cpu() - lp).abs().max()),
            'logits_max_abs_err_mfma_vs_source_network': float((lm.cpu() - lp).abs().max()), 'seconds': time.time() - t0}


def oracle_parity_csr(knet, x_cipher, logits, n_img=8):
    """Checker for the untiled (permutation) key-nets: the CPU oracle (oracle/: scipy csr_matvecs restated) recomputes the first images
    through EVERY layer of the same stored-order operators; the device logits of the timed batch must equal them bit for bit."""
    import oracle
    t0 = time.time()
    yo = np.ascontiguousarray(x_cipher[:n_img].cpu().numpy().T)                     # [D0+1, n] feature-major
    children = list(knet._keynet.named_children())
    i = 0
    while i < len(children):
        (name, c) = children[i]
        if not isinstance(c, KeyedLayer) or not isinstance(c.W, ksp.SparseMatrix) or isinstance(c.W, ksp.TiledMatrix):
            return {'check': 'CPU oracle on every layer', 'ok': None, 'skipped': 'layer %s is not a plain stored-order CSR operator' % name}
        fuse = (i + 1 < len(children)) and isinstance(children[i + 1][1], torch.nn.ReLU)
        (ip, ix, dt) = ksp._stored_order_csr(c.W._matrix)
        yo = oracle.csr_matvecs(c.W.shape, ip, ix, dt, yo)
        if fuse or c.iskeyedrelu():
            yo = np.maximum(yo, 0)
        i += 2 if fuse else 1
    got = logits[:n_img].contiguous().cpu().numpy()
    eq = bool(np.array_equal(got, yo.T[:, :-1]))
    return {'check': 'logits of the timed batch vs the CPU oracle (scipy csr_matvecs restated) run through every layer on the first %d images' % n_img,
            'bit_equal': eq, 'ok': eq, 'images': n_img, 'seconds': time.time() - t0}
