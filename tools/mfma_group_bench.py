"""A/B of the grouped CSR kernels on one AllConvNet-shaped layer (un-permuted 3x3 conv, C channels, HxW pixels, batch B): the matrix-pipe
kernel (kn_csr_mfma.hip) against the vector-ALU pipeline (KN_NO_GROUP_MFMA=1), same process, interleaved; T MAC/s and bit-equality."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keynet_amd import sparse as ksp          # noqa: E402
from keynet_amd.sparse import sparse_toeplitz_conv2d    # noqa: E402


def main():
    (C, H, B) = (int(sys.argv[1]) if len(sys.argv) > 1 else 96, int(sys.argv[2]) if len(sys.argv) > 2 else 32, int(sys.argv[3]) if len(sys.argv) > 3 else 4096)
    rng = np.random.RandomState(0)
    w = (rng.randn(C, C, 3, 3) / np.sqrt(9 * C)).astype(np.float32)
    b = rng.randn(C).astype(np.float32)
    t0 = time.time()
    M = sparse_toeplitz_conv2d((C, H, H), w, bias=b, stride=1)
    W = ksp.SparseMatrix(M.tocsr())
    print('operator %s nnz %d built in %.1f s' % (str(M.shape), M.nnz, time.time() - t0), flush=True)
    dev = torch.device('cuda:0')
    x = torch.randn(M.shape[1], B, device=dev)
    x[-1] = 1
    op = W._device_op(dev)
    print(op.plan(B, 2))
    res = {}
    outs = {}
    for rnd in range(3):
        for mode in ('valu', 'mfma'):
            if mode == 'valu':
                os.environ['KN_NO_GROUP_MFMA'] = '1'
            else:
                os.environ.pop('KN_NO_GROUP_MFMA', None)
            y = W.torchdot(x, relu=True)
            torch.cuda.synchronize()
            (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            e0.record()
            for _ in range(5):
                y = W.torchdot(x, relu=True)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) / 5)
            outs[mode] = y
    os.environ.pop('KN_NO_GROUP_MFMA', None)
    macs = float(M.nnz) * B
    for (k, v) in res.items():
        ms = float(np.median(v))
        print('%s: %s ms  median %.3f ms  %.2f T MAC/s' % (k, ' '.join('%.3f' % t for t in v), ms, macs / ms / 1e9))
    print('bit-equal:', bool(torch.equal(outs['valu'], outs['mfma'])))


if __name__ == '__main__':
    main()
