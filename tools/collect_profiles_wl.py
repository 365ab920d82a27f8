#!/usr/bin/env python3
"""Copy the condensed artefacts of tools/run_profiles_wl.sh runs into profiles/:  python3 tools/collect_profiles_wl.py r03 gpurun_out/p_lenet:lenet_b1024 ..."""
import csv
import glob
import os
import shutil
import sys

csv.field_size_limit(1 << 30)
tag = sys.argv[1]
for spec in sys.argv[2:]:
    (d, name) = spec.split(':')
    pre = 'profiles/%s_%s_' % (tag, name)
    shutil.copy(d + '/bench.json', pre + 'bench.json')
    shutil.copy(d + '/per_layer_trace.csv', pre + 'per_layer_trace.csv')
    open(pre + 'layers.log', 'w').write(''.join(l for l in open(d + '/bench.log') if 'bench' in l))
    # PMC table: keep kn:: compute kernels, at most the first 3 occurrences of each (kernel, grid) -- the passes repeat every layer several times
    rows = list(csv.reader(open(d + '/pmc.csv')))
    with open(pre + 'pmc.csv', 'w') as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if int(r[2]) < 3:
                w.writerow(r)
    st = glob.glob(d + '/stats/**/*kernel_stats.csv', recursive=True)
    if st:
        with open(pre + 'kernel_stats.csv', 'w') as f:
            w = csv.writer(f)
            for (i, r) in enumerate(csv.reader(open(st[0]))):
                w.writerow([c if len(c) < 150 else c[:147] + '...' for c in r])
    print(pre, [os.path.getsize(p) for p in sorted(glob.glob(pre + '*'))])
