#!/usr/bin/env python3
"""Copy the condensed artefacts of tools/run_profiles_wl.sh runs into profiles/:  python3 tools/collect_profiles_wl.py r03 gpurun_out/p_lenet:lenet_b1024 ...
A third field names the dominant kernel (substring): `gpurun_out/p_allconv:allconv_b4096:csr_group_mfma_kernel` also writes <prefix>traffic.json = the
HBM bytes per forward of those launches (FETCH_SIZE doubled per the guide's gfx950 note + WRITE_SIZE, first forward of the PMC passes) with the sha256
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
        import hashlib
        import json
        h = hashlib.sha256()
        cs = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'keynet_amd', 'csrc')
        for f in sorted(os.listdir(cs)):
            if f.endswith(('.hip', '.h')):
                h.update(open(os.path.join(cs, f), 'rb').read())
        (ik, io, ifx, iw) = (rows[0].index('kernel'), rows[0].index('occ'), rows[0].index('fetch_GB_x2'), rows[0].index('write_GB'))
        first = [r for r in rows[1:] if dominant in r[ik]]
        # occurrence numbers count per (kernel, grid): the first forward = the lowest occurrence of each grid, repeated grids (two layers of one shape) in launch order
        n_first = {}
        for r in first:
            n_first.setdefault(r[1], []).append(int(r[io]))
        bench = json.loads([l for l in open(d + '/bench.json') if l.startswith('{')][0])
        mode = 'exact' if str(bench.get('config', {}).get('mode', '')).startswith('exact') else 'tolerance'
        per_fwd = json.load(open(d + '/layers.json')) if os.path.exists(d + '/layers.json') else {}
        n_launch = sum(1 for l in per_fwd.get('layers', []) if dominant in str(l.get('plan', '')))
        use = first[:n_launch] if n_launch else first
        json.dump({'workload': name, 'mode': mode, 'csrc_sha256': h.hexdigest(), 'dominant_kernel': dominant, 'launches_per_forward': len(use),
                   'dominant_hbm_bytes_per_forward': sum((float(r[ifx]) + float(r[iw])) * 1e9 for r in use),
                   'fetch_bytes_x2': sum(float(r[ifx]) * 1e9 for r in use), 'write_bytes': sum(float(r[iw]) * 1e9 for r in use),
                   'source': 'separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/run_profiles_wl.sh), rows of %spmc.csv' % pre}, open(pre + 'traffic.json', 'w'), indent=1)
    st = sorted(glob.glob(d + '/stats/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime, reverse=True)      # newest first: gpurun merges runs into one directory
    if st:
        with open(pre + 'kernel_stats.csv', 'w') as f:
            w = csv.writer(f)
            for (i, r) in enumerate(csv.reader(open(st[0]))):
                w.writerow([c if len(c) < 150 else c[:147] + '...' for c in r])
    print(pre, [os.path.getsize(p) for p in sorted(glob.glob(pre + '*'))])
