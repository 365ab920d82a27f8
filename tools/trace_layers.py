#!/usr/bin/env python3
"""Per-LAYER kernel table from a rocprofv3 --kernel-trace of `bench.py --trace-layers FILE` (every layer launched 8 times, a torch
elementwise marker kernel between layers): layer, kernel, launches, avg_us (+ algorithmic TFLOP/s and GB/s of the layer), so that the
roofline fraction can be recomputed from the committed trace alone.
    python3 tools/trace_layers.py <rocprof dir> <layers.json> > profiles/rNN_<workload>_per_layer_trace.csv"""
import collections
import csv
import glob
import json
import os
import re
import sys

csv.field_size_limit(1 << 30)


def main(d, layers_json):
    meta = json.load(open(layers_json))
    f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    segs = []
    cur = None
    for r in rows:
        name = r['Kernel_Name']
        if 'elementwise' in name and 'kn::' not in name:
            if cur is not None:
                segs.append(cur)
            cur = []
            continue
        if cur is not None and 'kn::' in name:
            cur.append(r)
    # the markers of the layer loop are the LAST len(layers) + 1 separators
    segs = segs[-len(meta['layers']):]
    w = csv.writer(sys.stdout)
    w.writerow(['layer', 'kind', 'kernel', 'launches', 'avg_us', 'layer_us_per_forward', 'layer_TFLOPs', 'layer_alg_GBs', 'flops', 'alg_bytes'])
    tot = collections.defaultdict(float)
    for (L, seg) in zip(meta['layers'], segs):
        by = collections.OrderedDict()
        for r in seg:
            m = re.search(r'kn::(\w+)(<[^>]*>)?', r['Kernel_Name'])
            k = (m.group(1) + (m.group(2) or '')) if m else r['Kernel_Name'][:60]
            by.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
        per_fwd = sum(sum(v) for v in by.values()) / meta['launches_per_layer']
        for (k, v) in by.items():
            w.writerow([L['name'], L['kind'], k, len(v), '%.2f' % (sum(v) / len(v)), '%.2f' % per_fwd, '%.2f' % (L['flops'] / per_fwd / 1e6), '%.1f' % (L['bytes'] / per_fwd / 1e3),
                        '%.6g' % L['flops'], '%.6g' % L['bytes']])
        tot[L['kind']] += per_fwd
        tot['flops_' + L['kind']] += L['flops']
    for k in [k for k in tot if not k.startswith('flops_')]:
        w.writerow(['TOTAL', k, '', '', '', '%.2f' % tot[k], '%.2f' % (tot['flops_' + k] / tot[k] / 1e6), '', '%.6g' % tot['flops_' + k], ''])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
