#!/usr/bin/env python3
"""Per-dispatch dump of a rocprofv3 --pmc pass: kernel (kn:: short name), grid, ms and every collected counter, one line per launch.
    python3 tools/pmc_dump.py <dir containing *counter_collection.csv>"""
import collections
import csv
import glob
import os
import re
import sys

csv.field_size_limit(1 << 30)


def main(d):
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            m = re.search(r'kn::(\w+)(<[^>]*>)?', r['Kernel_Name'])
            if not m:
                continue
            e = per.setdefault(int(r['Dispatch_Id']), {'kernel': m.group(1) + (m.group(2) or ''), 'grid': r['Grid_Size'], 'ms': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6})
            e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        for (k, e) in sorted(per.items()):
            print(' '.join('%s=%s' % (a, ('%.4g' % b) if isinstance(b, float) else b) for (a, b) in e.items()))


if __name__ == '__main__':
    main(sys.argv[1])
