"""A/B: what the per-forward contract screen costs on the headline key-net (same process, interleaved): KN_NO_RESCREEN=1 vs default.
Also prints which kernels the screen adds (per forward) from the plan strings."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    (sensor, knet, inshape, batch, desc, net) = bench.build_workload('vgg16', 0, exact='auto')
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(1234)
    x = torch.randn((batch,) + tuple(inshape), generator=g, device=dev)
    xc = sensor.fromtensor(x).encrypt().astensor()
    knet.forward_linear(xc)
    knet.forward_linear(xc)
    torch.cuda.synchronize()

    def timed(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            knet.forward_linear(xc)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n
    res = {'screen': [], 'noscreen': [], 'gather_only': []}
    for rnd in range(4):
        for mode in ('noscreen', 'gather_only', 'screen'):
            os.environ.pop('KN_NO_RESCREEN', None)
            os.environ.pop('KN_RESCREEN_NOREAD', None)
            if mode == 'noscreen':
                os.environ['KN_NO_RESCREEN'] = '1'
            elif mode == 'gather_only':
                os.environ['KN_RESCREEN_NOREAD'] = '1'
            timed(2)
            res[mode].append(timed(10))
    os.environ.pop('KN_NO_RESCREEN', None)
    os.environ.pop('KN_RESCREEN_NOREAD', None)
    for (k, v) in res.items():
        print('%-9s %s  median %.3f ms' % (k, ' '.join('%.3f' % t for t in v), float(np.median(v))))
    print('screen cost: %.3f ms per forward' % (float(np.median(res['screen'])) - float(np.median(res['noscreen']))))
    print(knet.contract_report()['rescreen'], [r['name'] for r in knet.contract_report()['layers'] if r['screened']])


if __name__ == '__main__':
    main()
