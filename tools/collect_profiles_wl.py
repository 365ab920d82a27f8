#!/usr/bin/env python3
"""Copy the condensed artefacts of tools/run_profiles_wl.sh runs into profiles/:  python3 tools/collect_profiles_wl.py r03 gpurun_out/p_lenet:lenet_b1024 ...
A third field names the dominant kernel (substring): `gpurun_out/p_allconv:allconv_b4096:csr_group_mfma_kernel` also writes <prefix>traffic.json = the
HBM bytes of those launches of ONE marked forward (tools/pmc_forward.py: FETCH_SIZE doubled per the guide's gfx950 note + WRITE_SIZE) with the sha256
of the kernel sources, which bench.py quotes as roofline.traffic when its own sources are byte-identical."""
import csv
import glob
import os
import shutil
import sys

csv.field_size_limit(1 << 30)
tag = sys.argv[1]
for spec in sys.argv[2:]:
    (d, name) = spec.split(':')[:2]
    dominant = spec.split(':')[2] if spec.count(':') >= 2 else None
    pre = 'profiles/%s_%s_' % (tag, name)
    shutil.copy(d + '/bench.json', pre + 'bench.json')
    shutil.copy(d + '/per_layer_trace.csv', pre + 'per_layer_trace.csv')
    if os.path.exists(d + '/bench_detail.json'):
        shutil.copy(d + '/bench_detail.json', pre + 'bench_detail.json')
    open(pre + 'layers.log', 'w').write(''.join(l for l in open(d + '/bench.log') if 'bench' in l))
    # PMC table: keep kn:: compute kernels, at most the first 3 occurrences of each (kernel, grid) -- the passes repeat every layer several times
    rows = list(csv.reader(open(d + '/pmc.csv')))
    with open(pre + 'pmc.csv', 'w') as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if int(r[2]) < 3:
                w.writerow(r)
    if dominant:
        # roofline.traffic: the launches of ONE marked forward (bench.py --pmc-forward in passes 1 / 2 of tools/run_profiles_wl.sh), condensed by tools/pmc_forward.py; a fourth
        # field names the layer kinds (bench's per-layer table) whose algorithmic bytes the dominant kernel is priced against
        import subprocess
        kinds = spec.split(':')[3] if spec.count(':') >= 3 else 'convexact,csr'
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        out = subprocess.check_output([sys.executable, os.path.join(root, 'tools', 'pmc_forward.py'), d + '/p1', d + '/p2', d + '/forward1.json', dominant, kinds])
        open(pre + 'traffic.json', 'wb').write(out)
    st = sorted(glob.glob(d + '/stats/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime, reverse=True)      # newest first: gpurun merges runs into one directory
    if st:
        with open(pre + 'kernel_stats.csv', 'w') as f:
            w = csv.writer(f)
            for (i, r) in enumerate(csv.reader(open(st[0]))):
                w.writerow([c if len(c) < 150 else c[:147] + '...' for c in r])
    print(pre, [os.path.getsize(p) for p in sorted(glob.glob(pre + '*'))])
