#!/usr/bin/env python3
"""How a caller's tensor LAYOUT changes the keyed VGG-16 forward: feature-major memory (what bench.py holds: x.t() contiguous) against the row-major [N, D] batch a user of the
reference would pass, on the matrix cores (exact='auto') and in the stored order.   python3 tools/layout_time.py [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlegs.workloads import build_workload      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device('cuda:0')
for exact in ('auto', None):
    (sensor, knet, inshape, _, desc, net) = build_workload('vgg16', 0, exact=exact)
    x = torch.randn((n,) + inshape, device=dev)
    xc = sensor.fromtensor(x).encrypt().astensor()            # feature-major memory
    xr = xc.contiguous()                                      # row-major [N, D]
    assert xc.t().is_contiguous() and xr.is_contiguous()
    for (name, t) in (('feature-major', xc), ('row-major', xr)):
        for _ in range(2):
            y = knet.forward_linear(t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            y = knet.forward_linear(t)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 5
        print('contract %-5s %-14s %8.2f ms per forward of %d images   logits equal to the feature-major run: %s' % (
            'auto' if exact == 'auto' else 'exact', name, ms, n, bool(torch.equal(y, knet.forward_linear(xc)))), flush=True)
    del knet, sensor
    torch.cuda.empty_cache()
