#!/usr/bin/env python3
"""LeNet whole-net kernel: microseconds per forward (HIP events over 5 x 200 back-to-back launches behind 2 000 warm-up launches; min / median).
    KEYNET_HIP_LIB=/tmp/variant.so python3 tools/chain_time.py [n_images] [label]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keynet_amd import system as ksys       # noqa: E402
from keynet_amd.models import LeNet_AvgPool  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(os.environ.get('KEYNET_HIP_LIB', 'product'))
torch.manual_seed(0)
net = LeNet_AvgPool().eval()
np.random.seed(0)
(sensor, knet) = ksys.PermutationKeynet((1, 28, 28), net)
x = sensor.fromtensor(torch.randn(n, 1, 28, 28, device='cuda:0')).encrypt().astensor()
for _ in range(2000):
    y = knet.forward_linear(x)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    e0.record()
    for _ in range(200):
        y = knet.forward_linear(x)
    e1.record()
    torch.cuda.synchronize()
    ts.append(1e3 * e0.elapsed_time(e1) / 200)
print('  %s | us per forward: min %.2f median %.2f' % (label, min(ts), float(np.median(ts))), flush=True)
