#!/usr/bin/env python3
"""Per-dispatch condensation of rocprofv3 --pmc passes over bench.py: one row per launch of a kn:: kernel
(dispatch order = layer order inside a forward), so traffic / L2 hit rate / clock can be read PER LAYER.

    python3 tools/pmc_layers.py <dir with pmc_*/runc/*counter_collection.csv> > per_layer.csv
Each pmc_<tag> directory is one rocprofv3 pass (separate counter sets); rows are matched across passes by
(kernel short name, occurrence index).  FETCH_SIZE / WRITE_SIZE are KiB; FETCH is doubled (MI355X_MICROARCH.md HBM section)."""
import collections
import csv
import glob
import os
import re
import sys

csv.field_size_limit(1 << 30)


def short(name):
    m = re.search(r'kn::(\w+)(<[^>]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else None


def read_pass(d):
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        return {}
    per = collections.OrderedDict()
    for r in csv.DictReader(open(files[0])):
        s = short(r['Kernel_Name'])
        if s is None:
            continue
        key = int(r['Dispatch_Id'])
        e = per.setdefault(key, {'kernel': s, 'grid': int(r['Grid_Size']), 'vgpr': r['VGPR_Count'], 'agpr': r['Accum_VGPR_Count'], 'sgpr': r['SGPR_Count'],
                                 'lds': r['LDS_Block_Size'], 'ns': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
        e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    occ = collections.Counter()
    out = collections.OrderedDict()
    for (k, e) in sorted(per.items()):
        i = occ[e['kernel']]
        occ[e['kernel']] += 1
        out[(e['kernel'], e['grid'], i)] = e
    return out


def main(root):
    passes = sorted(glob.glob(os.path.join(root, 'pmc_*')))
    merged = collections.OrderedDict()
    for p in passes:
        if not os.path.isdir(p):
            continue
        for (k, e) in read_pass(p).items():
            m = merged.setdefault(k, {})
            for (kk, vv) in e.items():
                if kk == 'ns':
                    m.setdefault('ns_' + os.path.basename(p), vv)
                else:
                    m.setdefault(kk, vv)
    cols = ['kernel', 'grid', 'occ', 'vgpr', 'agpr', 'sgpr', 'lds', 'ms', 'fetch_GB_x2', 'write_GB', 'l2_hit', 'clock_GHz', 'mfma_busy', 'lds_conflict']
    w = csv.writer(sys.stdout)
    w.writerow(cols)
    for ((kern, grid, i), m) in merged.items():
        ns = [v for (k, v) in m.items() if k.startswith('ns_')]
        ms = min(ns) * 1e-6 if ns else 0.0
        f = m.get('FETCH_SIZE')
        wr = m.get('WRITE_SIZE')
        (h, mi) = (m.get('TCC_HIT_sum'), m.get('TCC_MISS_sum'))
        gui = m.get('GRBM_GUI_ACTIVE')
        ns_m = m.get('ns_pmc_mfma')
        clock = (gui / 8 / (ns_m * 1e-9) / 1e9) if (gui and ns_m) else None
        busy = (m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * gui / 8)) if (gui and 'SQ_VALU_MFMA_BUSY_CYCLES' in m) else None
        fmt = (lambda v, p='%.3f': '' if v is None else p % v)
        w.writerow([kern, grid, i, m.get('vgpr'), m.get('agpr'), m.get('sgpr'), m.get('lds'), '%.3f' % ms, fmt(None if f is None else 2 * f * 1024 / 1e9),
                    fmt(None if wr is None else wr * 1024 / 1e9), fmt(None if not h else h / (h + mi)), fmt(clock), fmt(busy), fmt(m.get('SQ_LDS_BANK_CONFLICT'), '%.0f')])


if __name__ == '__main__':
    main(sys.argv[1])
