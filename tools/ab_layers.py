#!/usr/bin/env python3
"""Same-process A/B of kernel options on the REAL keyed VGG-16 layers.  The library reads its KN_* options when an operator is CREATED and
records them in the handle (kn_internal.h: Tuning), so every variant gets its own resident operator; the diagnostic options (KN_OCC, KN_EXACT_XD, ...)
exist in the -DKN_ABLATION build only (tools/ablate_conv.sh builds it on demand under /tmp and points KEYNET_HIP_LIB at it).

    python3 tools/ab_layers.py --layers conv1_1,conv4_2,conv5_1 --variants "base;KN_OCC=3;KN_OCC=2" --rounds 5
Each round times every variant once per layer (interleaved: A B C A B C ...), HIP events on the launch stream; prints median and
min per (layer, variant).  Inputs of a layer are the key-net's own activations at batch 256."""
import argparse
import os
import sys
import time
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from keynet_amd.layer import KeyedLayer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', default='conv1_1,conv1_2,conv2_2,conv3_2,conv4_2,conv5_1')
    ap.add_argument('--variants', default='base')
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--exact', action='store_true')
    ap.add_argument('--warm', type=int, default=3)
    ap.add_argument('--reps', type=int, default=4)
    args = ap.parse_args()
    variants = []
    for v in args.variants.split(';'):
        v = v.strip()
        variants.append((v, {} if v == 'base' else dict(kv.split('=') for kv in v.split(','))))
    knobs = sorted({k for (_, d) in variants for k in d})
    # matrix cores forced (exact=False) unless --exact: a permutation-only key-net is bit-exact by default since round 4, and 'auto' would calibrate inside the timing
    (sensor, knet, inshape, batch, desc, net) = bench.build_workload('vgg16', 0, exact=bool(args.exact))
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(1234)
    x = torch.randn((args.batch,) + tuple(inshape), generator=g, device=dev)
    y = sensor.fromtensor(x).encrypt().astensor()
    del x
    want = args.layers.split(',')
    children = list(knet._keynet.named_children())
    res = {}
    for (i, (name, c)) in enumerate(children):
        if not isinstance(c, KeyedLayer):
            continue
        fuse = (i + 1 < len(children)) and isinstance(children[i + 1][1], torch.nn.ReLU)
        if args.exact:
            c._exact = True
        if name in want:
            import copy
            nnz = bench.host_nnz(c)
            # one resident operator per variant (options are read when the device operator is built)
            layer_of = {}
            for (v, d) in variants:
                for k in knobs:
                    os.environ.pop(k, None)
                os.environ.update(d)
                if d:
                    lc = copy.copy(c)
                    lc.W = copy.copy(c.W)
                    (lc.W._op, lc.W._op_dense) = (None, None)
                else:
                    lc = c
                lc.forward(y, fuse_relu=fuse)
                layer_of[v] = lc
            torch.cuda.synchronize()
            times = {v: [] for (v, _) in variants}
            for r in range(args.rounds):
                for (v, d) in variants:
                    lc = layer_of[v]
                    for _ in range(args.warm):                       # back-to-back launches: the clock settles under THIS load
                        out = lc.forward(y, fuse_relu=fuse)
                    (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    e0.record()
                    for _ in range(args.reps):
                        out = lc.forward(y, fuse_relu=fuse)
                    e1.record()
                    torch.cuda.synchronize()
                    times[v].append(e0.elapsed_time(e1) / args.reps)
                    del out
            for k in knobs:
                os.environ.pop(k, None)
            for (v, _) in variants:
                t = np.array(times[v])
                print('%-8s %-36s median %8.4f ms  min %8.4f ms  %7.2f TFLOP/s' % (name, v, np.median(t), t.min(), 2.0 * nnz * args.batch / np.median(t) / 1e9), flush=True)
            del layer_of
        y = c.forward(y, fuse_relu=fuse)
        if name == want[-1]:
            break


if __name__ == '__main__':
    main()
