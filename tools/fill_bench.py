#!/usr/bin/env python3
"""The filled-in order-preserving kernel (convtaps_exact_fill_kernel) on ONE synthetic operator shaped like a layer of the reference's doubly-stochastic VGG-16 (hundreds of slots
per output pixel, several taps per pixel pair), per kernel form: T stored MAC/s against the no-FMA roof.  Forms are forced with KN_FILL_FORM (diagnostic build, read at create):
    KEYNET_HIP_LIB=/tmp/libkn_abl.so python3 tools/fill_bench.py [n_vecs] [once]      ("once": one form given by the environment, a few launches -- for rocprofv3 --pmc passes)
The operator: H x H output pixels, every pixel reads `fill` input pixels of its 14 x 14 key block through each of the nine taps (entries of one pixel pair under several taps)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keynet_amd import sparse as ksp, _capi        # noqa: E402


def operator(H=28, Cin=32, Cout=256, fill=60, seed=0):
    rng = np.random.RandomState(seed)
    HW = H * H
    taps = (rng.randn(9, Cout, Cin) / np.sqrt(9 * Cin)).astype(np.float32)
    blk = 14
    (eo, ei, et, ec) = ([], [], [], [])
    for o in range(HW):
        (r, c) = divmod(o, H)
        (r0, c0) = (r // blk * blk, c // blk * blk)
        cand = ((r0 + np.arange(blk))[:, None] * H + (c0 + np.arange(blk))[None, :]).ravel()          # the 196 pixels of the key block
        for t in range(9):
            ins = rng.choice(cand, size=fill, replace=False)
            eo.append(np.full(fill, o)); ei.append(ins); et.append(np.full(fill, t)); ec.append((rng.rand(fill) / fill).astype(np.float32))
    lastcol = np.concatenate((rng.randn(Cout * HW), [1.0])).astype(np.float32)
    return ksp.Conv2dTiledMatrix.fromtaps((Cin, H, H), (Cout, H, H), taps, np.concatenate(eo), np.concatenate(ei), np.concatenate(et), np.concatenate(ec), lastcol)


_X = {}


def run(W, n_vecs, form, reps=5):
    import copy
    dev = torch.device('cuda:0')
    if form:
        os.environ['KN_FILL_FORM'] = str(form)
    else:
        os.environ.pop('KN_FILL_FORM', None)
    Wv = copy.copy(W)
    Wv._op = None
    if n_vecs not in _X:                                        # (ONE input for every form: their results are compared bit for bit)
        _X[n_vecs] = torch.randn(W.shape[1], n_vecs, device=dev)
        _X[n_vecs][-1] = 1.0
    x = _X[n_vecs]
    with torch.cuda.device(dev):
        plan = Wv._device_op(dev).plan(n_vecs, _capi.KN_FLAG_EXACT).split(' (')[0]
        nnz = Wv._device_op(dev).nnz_expanded()
    y = Wv.torchdot(x, exact=True)
    torch.cuda.synchronize()
    (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    e0.record()
    for _ in range(reps):
        y = Wv.torchdot(x, exact=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return (plan, nnz, ms, y)


if __name__ == '__main__':
    n_vecs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    once = len(sys.argv) > 2 and sys.argv[2] == 'once'
    W = operator()
    t = W._taps
    pairs = len(np.unique(t['ent_out'].astype(np.int64) * (28 * 28) + t['ent_in']))
    print('operator: 28 x 28 pixels, 32 -> 256 channels, %d entries = %.0f slots per pixel, %.2f terms per stored entry; %d batch columns' % (len(t['ent_out']), len(t['ent_out']) / 784.0, len(t['ent_out']) / pairs, n_vecs))
    if once:
        (plan, nnz, ms, y) = run(W, n_vecs, int(os.environ.get('KN_FILL_FORM', '0')), reps=3)
        print('%-110s %8.3f ms  %6.2f T MAC/s' % (plan, ms, nnz * n_vecs / ms / 1e9))
        sys.exit(0)
    ref = None
    for form in (0, 1, 2, 3, 4):
        (plan, nnz, ms, y) = run(W, n_vecs, form)
        ref = y if ref is None else ref
        print('form %d  %-100s %8.3f ms  %6.2f T MAC/s = %.3f of the no-FMA roof   bit-equal to form 0: %s' % (form, plan[:100], ms, nnz * n_vecs / ms / 1e9, nnz * n_vecs / ms / 1e9 / 39.3, bool(torch.equal(y, ref))))
