#!/usr/bin/env python3
"""Per-dispatch table of several rocprofv3 --pmc passes (separate runs of the same program): one row per launch of a kn:: kernel with its
duration and every collected counter; FETCH_SIZE / WRITE_SIZE converted to GB (FETCH doubled: MI355X_MICROARCH.md HBM section), L2 hit rate
from TCC_HIT_sum / TCC_MISS_sum.  Rows are matched across passes by (kernel short name, grid, occurrence index).
    python3 tools/pmc_table.py <dir with p*/ subdirectories> > profiles/rNN_<workload>_pmc.csv"""
import collections
import csv
import glob
import os
import re
import sys

csv.field_size_limit(1 << 30)


def read_pass(d):
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        return {}
    per = collections.OrderedDict()
    for r in csv.DictReader(open(files[0])):
        m = re.search(r'kn::(\w+)(<[^>]*>)?', r['Kernel_Name'])
        if not m:
            continue
        e = per.setdefault(int(r['Dispatch_Id']), {'kernel': m.group(1) + (m.group(2) or ''), 'grid': int(r['Grid_Size']), 'vgpr': r['VGPR_Count'], 'lds': r['LDS_Block_Size'],
                                                   'us': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3})
        e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    occ = collections.Counter()
    out = collections.OrderedDict()
    for (k, e) in sorted(per.items()):
        key = (e['kernel'], e['grid'], occ[(e['kernel'], e['grid'])])
        occ[(e['kernel'], e['grid'])] += 1
        out[key] = e
    return out


def main(root):
    merged = collections.OrderedDict()
    counters = []
    for p in sorted(glob.glob(os.path.join(root, 'p*'))):
        if not os.path.isdir(p):
            continue
        for (k, e) in read_pass(p).items():
            m = merged.setdefault(k, {'us': []})
            for (kk, vv) in e.items():
                if kk == 'us':
                    m['us'].append(vv)
                elif kk in ('kernel', 'grid', 'vgpr', 'lds'):
                    m.setdefault(kk, vv)
                else:
                    m[kk] = vv
                    if kk not in counters:
                        counters.append(kk)
    w = csv.writer(sys.stdout)
    w.writerow(['kernel', 'grid', 'occ', 'vgpr', 'lds', 'us_min', 'fetch_GB_x2', 'write_GB', 'l2_hit'] + counters)
    for ((kern, grid, i), m) in merged.items():
        f = m.get('FETCH_SIZE')
        wr = m.get('WRITE_SIZE')
        (h, mi) = (m.get('TCC_HIT_sum'), m.get('TCC_MISS_sum'))
        w.writerow([kern, grid, i, m.get('vgpr'), m.get('lds'), '%.2f' % min(m['us']), '' if f is None else '%.4f' % (2 * f * 1024 / 1e9), '' if wr is None else '%.4f' % (wr * 1024 / 1e9),
                    '' if not h else '%.3f' % (h / (h + mi))] + ['%.6g' % m[c] if c in m else '' for c in counters])


if __name__ == '__main__':
    main(sys.argv[1])
