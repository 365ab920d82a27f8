#!/usr/bin/env python3
"""convtaps_bf16x3_kernel against the order-preserving kernel (= the reference's arithmetic) and the f32 MFMA kernel on VGG-16-shaped
permutation-keyed conv layers: max abs difference and time.   gpurun -- 'python3 tools/bf16x3_bench.py'"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keynet_amd import sparse as ksp, _capi     # noqa: E402

dev = torch.device('cuda:0')


def make(cin, cout, hw, seed=0, coef=False):
    rng = np.random.RandomState(seed)
    n = hw * hw
    taps = (rng.rand(9, cout, cin).astype(np.float32) - 0.5) * (2.0 / np.sqrt(9 * cin))
    perm_in = rng.permutation(n)
    perm_out = rng.permutation(n)
    (eo, ei, et) = ([], [], [])
    for t, (dy, dx) in enumerate([(a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)]):
        yy, xx = np.meshgrid(np.arange(hw), np.arange(hw), indexing='ij')
        ok = (yy + dy >= 0) & (yy + dy < hw) & (xx + dx >= 0) & (xx + dx < hw)
        eo.append(perm_out[(yy * hw + xx)[ok]])
        ei.append(perm_in[((yy + dy) * hw + (xx + dx))[ok]])
        et.append(np.full(int(ok.sum()), t))
    (eo, ei, et) = (np.concatenate(eo), np.concatenate(ei), np.concatenate(et))
    ec = (0.5 + rng.rand(len(eo))).astype(np.float32) if coef else None
    lastcol = np.concatenate((rng.randn(cout * n).astype(np.float32) * 0.1, [1.0])).astype(np.float32)
    return ksp.Conv2dTiledMatrix.fromtaps((cin, hw, hw), (cout, hw, hw), taps, eo, ei, et, ec, lastcol)


def timeit(f, reps=5):
    f()
    torch.cuda.synchronize()
    (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
  for (cin, cout, hw, B, coef) in [(64, 64, 56, 256, False), (64, 128, 28, 256, False), (256, 256, 56, 256, False), (512, 512, 28, 256, False), (512, 512, 14, 256, False), (128, 128, 28, 256, True)]:
      W = make(cin, cout, hw, coef=coef)
      x = torch.randn(cin * hw * hw + 1, B, device=dev)
      x[-1] = 1.0
      x[:-1].clamp_(min=0)                    # post-ReLU activations
      ye = W.torchdot(x, relu=True, exact=True)
      ym = W.torchdot(x, relu=True, exact=False)
      yb = W.torchdot(x, relu=True, exact='bf16x3')
      plan = W._device_op().plan(B, 1 | 4)
      flops = 2.0 * W._device_op().nnz_expanded() * B
      tm = timeit(lambda: W.torchdot(x, relu=True, exact=False))
      tb = timeit(lambda: W.torchdot(x, relu=True, exact='bf16x3'))
      print('Cin %3d Cout %3d %2dx%2d B %d coef %d | max|y| %.3g | mfma-exact %.3g  bf16x3-exact %.3g  bf16x3-mfma %.3g | f32 mfma %.3f ms %.1f TF | bf16x3 %.3f ms %.1f TF-equiv (x%.2f) | %s' %
            (cin, cout, hw, hw, B, int(coef), float(ye.abs().max()), float((ym - ye).abs().max()), float((yb - ye).abs().max()), float((yb - ym).abs().max()), tm, flops / tm / 1e9,
             tb, flops / tb / 1e9, tm / tb, plan.split(';')[0][:70]))
      del W, x, ye, ym, yb


if __name__ == '__main__':
    main()
