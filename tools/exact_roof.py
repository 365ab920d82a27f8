#!/usr/bin/env python3
"""profiles/rNN_vgg16_exact_roof.txt: the bit-exact VGG-16 forward layer by layer against the no-FMA roof, from a bench detail file (its `exact` leg).
    python3 tools/exact_roof.py profiles/r05_vgg16_b256_bench_detail.json > profiles/r05_vgg16_exact_roof.txt"""
import json
import sys

d = json.load(open(sys.argv[1]))
ex = d['exact']
ms = ex['layers_ms']
nnz = {r['name']: r['nnz'] for r in d.get('layers_table', [])} if 'layers_table' in d else None
# expanded non-zeros per image (identity / permutation keys)
NNZ = {'conv1_1': 89400065, 'conv1_2': 1841905665, 'pool1_2': 7985217, 'conv2_1': 915472385, 'conv2_2': 1829339137, 'pool2_2': 3971201, 'conv3_1': 903757825,
       'conv3_2': 1806712833, 'conv3_3': 1806712833, 'pool3_3': 1964289, 'conv4_1': 881729537, 'conv4_2': 1763057665, 'conv4_3': 1763057665, 'pool4_3': 961025,
       'conv5_1': 419530753, 'conv5_2': 419530753, 'conv5_3': 419530753, 'pool5_3': 229889, 'fc6': 102764545, 'fc7': 16781313, 'fc8': 10742335}
B = d['config']['images_per_gpu']
PEAK = 39.3
print('Keyed VGG-16 (TiledPermutationKeynet, tile 64), %d images, bit-exact contract (the default of this key-net): per layer, HIP events under sustained load (bench.py exact leg).' % B)
print('no-FMA roof: 39.3 T MAC/s = 256 CUs x 4 SIMDs x 16 MAC per clock at 2.4 GHz (one packed multiply + one packed add per 128 MACs and wavefront).')
print('%-9s %12s %9s %10s %8s' % ('layer', 'MAC/image', 'ms', 'T MAC/s', 'of roof'))
tot_ms = 0.0
tot_mac = 0.0
mid = []
for (n, t) in ms.items():
    mac = NNZ[n] * B
    r = mac / (t * 1e-3) / 1e12
    tot_ms += t
    tot_mac += mac
    print('%-9s %12d %9.3f %10.2f %8.3f' % (n, NNZ[n], t, r, r / PEAK))
    if n in ('conv1_2', 'conv2_2', 'conv3_2', 'conv3_3', 'conv4_2', 'conv4_3'):
        mid.append(r)
print('%-9s %12d %9.3f %10.2f %8.3f   (kernel time; the step is %.2f ms = %.1f images/s)' % ('SUM', sum(NNZ.values()), tot_ms, tot_mac / (tot_ms * 1e-3) / 1e12, tot_mac / (tot_ms * 1e-3) / 1e12 / PEAK,
                                                                                                 ex['ms_per_step'], ex['images_per_s']))
m = sum(mid) / len(mid)
ideal = sum(NNZ[n] * B / (m * 1e12) * 1e3 if not n.startswith('pool') else t for (n, t) in ms.items())
print()
print('The six long stride-1 layers run at %.1f T MAC/s = %.3f of the nominal roof.  profiles/r05_micro_valu_issue_rate.txt: a SIMD issues one packed f32 instruction per 1.99-2.02 ns' % (m, m / PEAK))
print('under sustained load on this pool (the clock settles near 2.1 GHz) = 32.4-32.9 T MAC/s for the mul + add pair: the long layers are AT what the issue port delivers.')
print('If every conv / fc layer ran at the long layers\' rate (pools as measured) the forward would take %.1f ms = %.0f images/s: the tails (conv5_x, conv1_1, fc6-8) are worth %.1f %%.' %
      (ideal, 1e3 * B / ideal, 100 * (tot_ms - ideal) / tot_ms))
print('Round 5 tried the two levers for the tails -- half-size work items (128-column tiles) and overlapping two half-batch windows -- and both lost (profiles/HISTORY.md).')
