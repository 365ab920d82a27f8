#!/usr/bin/env python3
"""`roofline.traffic`, ONE definition: HBM bytes of the launches of exactly ONE forward_linear (the timed step), from two separate rocprofv3 passes of
`bench.py --pmc-forward FILE` (--kernel-trace --pmc FETCH_SIZE, and --pmc WRITE_SIZE; never combined with other trace domains).  The forward is cut out of the
dispatch stream by the two torch elementwise marker kernels bench.py launches around it -- no "forwards in the run" estimate, no dispatch-order heuristics.

    python3 tools/pmc_forward.py <fetch dir> <write dir> <FILE written by bench.py> [kernel substring = convtaps_mfma_kernel] [layer kinds whose algorithmic bytes it is priced
                                 against, comma-separated = convtaps,dense] > profiles/rNN_<workload>_traffic.json

Units: FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads: MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import collections
import csv
import glob
import json
import os
import re
import sys

csv.field_size_limit(1 << 30)


def between_markers(d, counter):
    f = max(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    marks = [i for (i, r) in enumerate(rows) if 'elementwise' in r['Kernel_Name'] and 'kn::' not in r['Kernel_Name']]
    assert len(marks) >= 2, 'no marker pair in %s' % f
    (a, b) = (marks[-2], marks[-1])                 # the LAST two markers: the pair bench.py --pmc-forward puts around its one forward
    return [r for r in rows[a + 1:b] if 'kn::' in r['Kernel_Name']]


def short(name):
    m = re.search(r'kn::(\w+)(<[^>]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:60]


def main(fetch_dir, write_dir, meta_file, dominant='convtaps_mfma_kernel', kinds='convtaps,dense'):
    meta = json.load(open(meta_file))
    fetch = between_markers(fetch_dir, 'FETCH_SIZE')
    write = between_markers(write_dir, 'WRITE_SIZE')
    assert [short(r['Kernel_Name']) for r in fetch] == [short(r['Kernel_Name']) for r in write], 'the two passes did not launch the same kernels'
    launches = []
    per = collections.OrderedDict()
    for (rf, rw) in zip(fetch, write):
        k = short(rf['Kernel_Name'])
        (fb, wb) = (float(rf['Counter_Value']) * 1024.0, float(rw['Counter_Value']) * 1024.0)
        launches.append({'kernel': k, 'fetch_x2': 2.0 * fb, 'write': wb})
        e = per.setdefault(k, {'launches': 0, 'fetch_x2': 0.0, 'write': 0.0})
        e['launches'] += 1
        e['fetch_x2'] += 2.0 * fb
        e['write'] += wb
    dom = [l for l in launches if dominant in l['kernel']]
    alg = meta.get('algorithmic_bytes_per_forward', {})
    alg_dom = float(sum(alg.get(k, 0.0) for k in kinds.split(','))) or None
    out = {'definition': 'HBM bytes of the launches of ONE forward_linear between two marker kernels (bench.py --pmc-forward), separate --pmc FETCH_SIZE / WRITE_SIZE passes; '
                         'FETCH_SIZE x 2 (gfx950), KiB -> bytes', 'forwards_counted': 1, 'mode': meta.get('mode'), 'batch': meta.get('batch'), 'workload': meta.get('workload'),
           'csrc_sha256': meta.get('csrc_sha256'), 'dominant_kernel': dominant, 'dominant_launches': len(dom),
           'fetch_x2': sum(l['fetch_x2'] for l in dom), 'write': sum(l['write'] for l in dom),
           'dominant_hbm_bytes_per_forward': sum(l['fetch_x2'] + l['write'] for l in dom),
           'algorithmic_bytes_dominant': alg_dom, 'traffic_ratio': (sum(l['fetch_x2'] + l['write'] for l in dom) / alg_dom) if alg_dom else None,
           'whole_forward': {'launches': len(launches), 'fetch_x2': sum(l['fetch_x2'] for l in launches), 'write': sum(l['write'] for l in launches),
                             'algorithmic_bytes': float(sum(alg.values())) if alg else None},
           'per_kernel': per, 'launch_list': launches}
    if dominant == 'convtaps_mfma_kernel':
        out['convtaps_hbm_bytes_per_forward'] = out['dominant_hbm_bytes_per_forward']
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:6])
