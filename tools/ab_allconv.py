"""Same-process, interleaved A/B of the grouped-row kernels on the whole PermutationKeynet AllConvNet forward (B = 4096): matrix-pipe products
(default) against the vector-ALU pipeline (KN_NO_GROUP_MFMA=1, read per call); whole-forward time and per-layer times under each."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    (sensor, knet, inshape, batch, desc, net) = bench.build_workload('allconv', 0)
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(1234)
    x = torch.randn((batch,) + tuple(inshape), generator=g, device=dev)
    xc = sensor.fromtensor(x).encrypt().astensor()
    knet.forward_linear(xc)
    torch.cuda.synchronize()

    def setmode(m):
        if m == 'valu':
            os.environ['KN_NO_GROUP_MFMA'] = '1'
        else:
            os.environ.pop('KN_NO_GROUP_MFMA', None)

    def timed(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            knet.forward_linear(xc)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n
    res = {'mfma': [], 'valu': []}
    for rnd in range(5):
        for m in ('valu', 'mfma'):
            setmode(m)
            timed(3)
            res[m].append(timed(15))
    for (k, v) in res.items():
        print('%-5s forward %s  median %.3f ms  %.0f images/s' % (k, ' '.join('%.2f' % t for t in v), float(np.median(v)), batch / float(np.median(v)) * 1e3))
    for m in ('valu', 'mfma'):
        setmode(m)
        table = bench.time_layers(xc, bench.layer_table(knet, batch), 3)
        print(m, 'per layer:', ' '.join('%s %.3f' % (r['name'], r['ms']) for r in table), ' sum %.3f' % sum(r['ms'] for r in table))
    os.environ.pop('KN_NO_GROUP_MFMA', None)


if __name__ == '__main__':
    main()
