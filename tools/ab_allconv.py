"""Same-process, interleaved A/B of per-call kernel switches on the whole PermutationKeynet AllConvNet forward (B = 4096): whole-forward time, per-layer
times and bit-equality of the logits under each variant.
    python3 tools/ab_allconv.py                                        # row blocks per workgroup of the tap-table kernel: default, 3, 2, 1
    python3 tools/ab_allconv.py base KN_NO_EXACT_TABLE=1 KN_TABLE_NRB=3,KN_MF_PF=6
A variant is a comma-separated list of NAME=value environment settings that the library reads per call ('base' = none)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    variants = sys.argv[1:] or ['base', 'KN_TABLE_NRB=3', 'KN_TABLE_NRB=2', 'KN_TABLE_NRB=1']
    names = sorted({kv.split('=')[0] for v in variants if v != 'base' for kv in v.split(',')})
    (sensor, knet, inshape, batch, desc, net) = bench.build_workload('allconv', 0)
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(1234)
    x = torch.randn((batch,) + tuple(inshape), generator=g, device=dev)
    xc = sensor.fromtensor(x).encrypt().astensor()
    for k in names:
        os.environ.pop(k, None)
    y0 = knet.forward_linear(xc).clone()
    torch.cuda.synchronize()
    for rnd in range(3):
        for v in variants:
            for k in names:
                os.environ.pop(k, None)
            if v != 'base':
                for kv in v.split(','):
                    (k, val) = kv.split('=')
                    os.environ[k] = val
            y = knet.forward_linear(xc)
            torch.cuda.synchronize()
            eq = torch.equal(y, y0)
            t0 = time.perf_counter()
            for _ in range(10):
                knet.forward_linear(xc)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 100
            table = bench.time_layers(xc, bench.layer_table(knet, batch), 3)
            print('%-28s forward %.3f ms (%.1f k images/s) bit-equal %s | ' % (v, ms, batch / ms, eq) +
                  ' '.join('%s %.3f' % (r['name'], r['ms']) for r in table if r['name'].startswith('conv')), flush=True)


if __name__ == '__main__':
    main()
