#!/usr/bin/env python3
"""Single-layer microbench of the conv-taps MFMA kernel (kernel iteration / rocprofv3 PMC runs; not the headline bench).

    python3 tools/conv_bench.py --cin 256 --cout 256 --hw 56 --batch 256 --iters 5 [--perm]
Builds a 3x3 'same' conv operator in factored form (identity or block-permutation spatial key), runs it, reports
TFLOP/s (algorithmic: 2 * nnz_expanded * batch) and checks one output pixel against a float64 host computation.
"""
import argparse
import os
import sys
import time
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keynet_amd import sparse as ksp          # noqa: E402
from keynet_amd import direct as kdirect      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cin', type=int, default=256)
    ap.add_argument('--cout', type=int, default=256)
    ap.add_argument('--hw', type=int, default=56)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--iters', type=int, default=7)
    ap.add_argument('--perm', action='store_true', help='block-permutation spatial key (tile = hw or 56)')
    ap.add_argument('--gain', action='store_true', help='photometric (per-pixel gain) keys on top: non-unit per-entry coefficients')
    ap.add_argument('--exact', action='store_true', help='time the order-preserving path (KN_FLAG_EXACT) instead of the MFMA path')
    args = ap.parse_args()
    rng = np.random.RandomState(0)
    (Cin, Cout, H) = (args.cin, args.cout, args.hw)
    HW = H * H
    w = (rng.randn(Cout, Cin, 3, 3) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rng.randn(Cout).astype(np.float32)
    (eo, ei, et, ec) = ([], [], [], [])
    (g_out, g_in) = ((rng.rand(HW) + 0.5).astype(np.float32), (rng.rand(HW) + 0.5).astype(np.float32))
    if args.perm:
        blk = min(H, 56) ** 2
        pi = np.concatenate([rng.permutation(blk) + k for k in range(0, HW, blk)])
        po = np.concatenate([rng.permutation(blk) + k for k in range(0, HW, blk)])
    else:
        (pi, po) = (np.arange(HW), np.arange(HW))
    for (t, ((i, j), S)) in enumerate(kdirect.shift_matrices((H, H), 3, 1)):
        S = S.tocoo()
        eo.append(po[S.row]); ei.append(pi[S.col]); et.append(np.full(S.nnz, t))
        ec.append((g_out[po[S.row]] / g_in[pi[S.col]]).astype(np.float32))
    taps = np.stack([w[:, :, i, j] for i in range(3) for j in range(3)])
    lastcol = np.concatenate((np.repeat(b, HW), [1.0])).astype(np.float32)
    W = ksp.Conv2dTiledMatrix.fromtaps((Cin, H, H), (Cout, H, H), taps, np.concatenate(eo), np.concatenate(ei), np.concatenate(et), np.concatenate(ec) if args.gain else None, lastcol)
    dev = torch.device('cuda:0')
    x = torch.randn((Cin * HW + 1, args.batch), device=dev)
    x[-1] = 1.0
    y = W.torchdot(x, relu=True, exact=args.exact)
    torch.cuda.synchronize()
    nnz = W._device_op().nnz_expanded()
    for _ in range(3):
        y = W.torchdot(x, relu=True, exact=args.exact)
    torch.cuda.synchronize()
    times = []
    for _ in range(args.iters):
        (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        e0.record()
        y = W.torchdot(x, relu=True, exact=args.exact)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    ms = float(np.median(times))
    ms_min = float(np.min(times))
    # spot check: output pixel o=po[H+1] (interior), all channels, vs float64
    o = int(po[H + 1])
    sel = np.concatenate(eo) == o
    (ins, tps) = (np.concatenate(ei)[sel], np.concatenate(et)[sel])
    xh = x.cpu().numpy().astype(np.float64)
    ref = np.zeros((Cout, args.batch))
    for (ii, tt) in zip(ins, tps):
        cf = float(g_out[o] / g_in[ii]) if args.gain else 1.0
        ref += cf * (taps[tt].astype(np.float64) @ xh[np.arange(Cin) * HW + ii])
    ref = np.maximum(ref + b[:, None].astype(np.float64), 0)
    got = y.cpu().numpy()[np.arange(Cout) * HW + o]
    err = float(np.abs(got - ref).max())
    if args.exact:
        print('EXACT path: %.2f T MAC/s (median), %.2f (min time)' % (nnz * args.batch / ms / 1e9, nnz * args.batch / ms_min / 1e9))
    print('cin=%d cout=%d hw=%d batch=%d perm=%d: median %.3f ms %.2f TFLOP/s | min %.3f ms %.2f TFLOP/s | max|err| vs f64 = %.2e' %
          (Cin, Cout, H, args.batch, int(args.perm), ms, 2.0 * nnz * args.batch / ms / 1e9, ms_min, 2.0 * nnz * args.batch / ms_min / 1e9, err))


if __name__ == '__main__':
    main()
