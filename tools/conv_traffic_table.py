#!/usr/bin/env python3
"""Table behind profiles/r05_conv_traffic_ablation.txt: per variant directory pair <tag>-FETCH_SIZE / <tag>-WRITE_SIZE written by tools/conv_traffic_ablation.sh, the HBM bytes
and the kernel time of the conv-taps matrix-core launches of ONE forward (the third of the run: warm), plus the images/s the same run printed."""
import collections
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)
R = sys.argv[1]


def last_forward(d, counter):
    f = max(glob.glob(d + '/*/*counter_collection.csv'), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter and 'convtaps_mfma_kernel' in r['Kernel_Name']]
    t = max(glob.glob(d + '/*/*kernel_trace.csv'), key=os.path.getmtime)
    dur = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(t))}
    # A forward starts with conv1_2's launch (the only 64-channel layer of 16-channel chunks: the 64 x 256 tile); behind it the other eleven conv layers run as two
    # half-batch windows each (the overlapped forward) and fc6-8 as split-K launches of the same kernel.  Forward 0 calibrates the contract, 1 warms up, forward 2
    # is the first timed step; behind the timed steps bench.py launches single layers for its per-layer table (not forwards: they follow the last marker).
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    marks = [i for (i, r) in enumerate(rows) if '<64, 256, 16' in r['Kernel_Name']]
    rows = rows[marks[2]:marks[3]]
    return (sum(float(r['Counter_Value']) for r in rows) * 1024, sum(dur.get(r['Dispatch_Id'], 0) for r in rows) * 1e-6, len(rows))


print('%-18s %10s %10s %10s %12s %12s %10s' % ('variant', 'fetch GB', 'write GB', 'total GB', 'conv ms(pmc)', 'images/s', 'launches'))
tags = sorted({os.path.basename(p).rsplit('-', 1)[0] for p in glob.glob(R + '/*-FETCH_SIZE')})
for t in (['base'] if 'base' in tags else []) + [x for x in tags if x != 'base']:
    try:
        (fb, ms, n) = last_forward('%s/%s-FETCH_SIZE' % (R, t), 'FETCH_SIZE')
        (wb, _, _) = last_forward('%s/%s-WRITE_SIZE' % (R, t), 'WRITE_SIZE')
        line = [l for l in open('%s/%s-FETCH_SIZE.json' % (R, t)) if l.startswith('{')]
        v = json.loads(line[-1])['value'] if line else float('nan')
        print('%-18s %10.2f %10.2f %10.2f %12.2f %12.1f %10d' % (t, 2 * fb / 1e9, wb / 1e9, (2 * fb + wb) / 1e9, ms, v, n))
    except Exception as e:
        print('%-18s failed: %s' % (t, e))
print('(fetch = FETCH_SIZE x 2: gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md; algorithmic bytes of the forward: 19.83 GB; images/s = the run under the profiler, not a bench figure)')
