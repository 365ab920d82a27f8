#!/usr/bin/env python3
"""Where the whole-net kernel (csrc/kn_chain.hip) spends its time on LeNet_AvgPool: chains of the first k operators, and single
operators, timed at several batch widths (4 columns = ONE workgroup: no contention between CUs).
    gpurun -- 'python3 tools/chain_bench.py'"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keynet_amd import system as ksys, _capi          # noqa: E402
from keynet_amd.models import LeNet_AvgPool            # noqa: E402
from keynet_amd.layer import KeyedLayer                # noqa: E402

torch.manual_seed(0)
net = LeNet_AvgPool().eval()
np.random.seed(0)
(sensor, knet) = ksys.PermutationKeynet((1, 28, 28), net)
dev = torch.device('cuda:0')
children = list(knet._keynet.children())
steps = []
i = 0
while i < len(children):
    c = children[i]
    fuse = (i + 1 < len(children)) and isinstance(children[i + 1], torch.nn.ReLU)
    steps.append((c.W, 1 if fuse else 0))
    i += 2 if fuse else 1
ops = [W._device_op(dev) for (W, _) in steps]
flags = [f for (_, f) in steps]
s = torch.cuda.current_stream().cuda_stream


def timeit(op, rows, cols, n, reps=50):
    x = torch.randn(cols, n, device=dev)
    y = torch.empty(rows, n, device=dev)
    for _ in range(5):
        op.spmm(x.data_ptr(), n, n, y.data_ptr(), n, 2, s)
    torch.cuda.synchronize()
    (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    e0.record()
    for _ in range(reps):
        op.spmm(x.data_ptr(), n, n, y.data_ptr(), n, 2, s)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


names = ['conv1', 'pool1', 'conv2', 'pool2', 'fc1', 'fc2', 'fc3']
print('prefix chains (us):  n=4 / 64 / 256 / 1024')
for k in range(1, len(ops) + 1):
    ch = _capi.Operator.chain(ops[:k], flags[:k])
    (r, c) = ch.shape()
    print('  first %d (..%s): ' % (k, names[k - 1]) + ' / '.join('%7.1f' % timeit(ch, r, c, n) for n in (4, 64, 256, 1024)))
print('single-operator chains (us):  n=4 / 64 / 256 / 1024')
for k in range(len(ops)):
    ch = _capi.Operator.chain(ops[k:k + 1], flags[k:k + 1])
    (r, c) = ch.shape()
    print('  %s [%d x %d]: ' % (names[k], r, c) + ' / '.join('%7.1f' % timeit(ch, r, c, n) for n in (4, 64, 256, 1024)))
print('launch-per-layer kernels (us), n = 1024:')
for k in range(len(ops)):
    (r, c) = ops[k].shape()
    print('  %s: %7.1f' % (names[k], timeit(ops[k], r, c, 1024)))
