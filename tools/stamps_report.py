#!/usr/bin/env python3
"""Report of a KN_STAMPS diagnostic file (per-workgroup s_memrealtime start/end, 100 MHz): where a conv launch loses time.
    python3 tools/stamps_report.py gpurun_out/stamps.bin"""
import sys
import numpy as np


def main(path):
    raw = np.fromfile(path, dtype=np.int64)
    i = 0
    seen = {}
    while i + 4 <= len(raw):
        assert raw[i] == 0x7374616d70, 'bad header'
        (seq, grid, meta) = (int(raw[i + 1]), int(raw[i + 2]), int(raw[i + 3]))
        d = raw[i + 4:i + 4 + 4 * grid].reshape(grid, 4)
        i += 4 + 4 * grid
        key = (grid, meta)
        seen[key] = (seq, d)           # keep the LAST launch of each shape (warm)
    for ((grid, meta), (seq, d)) in seen.items():
        ok = d[:, 1] > 0
        (st, en, xcc, quad) = (d[ok, 0], d[ok, 1], d[ok, 2], d[ok, 3])
        t0 = st.min()
        (st, en) = ((st - t0) / 100.0, (en - t0) / 100.0)       # microseconds
        span = en.max()
        life = en - st
        full = quad < 0
        print('launch seq %d  grid %d  n_pix %d  cin*1000+n_mt %d: %d workgroups ran (%d full tiles, %d quarter tiles), span %.1f us' %
              (seq, grid, meta >> 32, meta & 0xffffffff, ok.sum(), full.sum(), (~full).sum(), span))
        print('   full-tile lifetime: median %.1f us  p10 %.1f  p90 %.1f   | quarter: median %.1f us' %
              (np.median(life[full]), np.percentile(life[full], 10), np.percentile(life[full], 90), np.median(life[~full]) if (~full).any() else 0))
        # work-weighted occupancy over time: each full tile = 1 unit, quarter = 0.25; resident workgroups over time in 20 bins
        edges = np.linspace(0, span, 21)
        for (a, b) in zip(edges[:-1], edges[1:]):
            resident = np.sum(np.clip(np.minimum(en, b) - np.maximum(st, a), 0, None)) / (b - a)
            finished = np.sum(np.where(full, 1.0, 0.25)[(en > a) & (en <= b)])
            print('   %7.0f..%7.0f us   resident workgroups %7.1f   tile-units finished %7.1f (%.3f per us)' % (a, b, resident, finished, finished / (b - a)))
        first_done = np.sort(en)[0]
        print('   first workgroup finished at %.1f us; last started at %.1f us; per-XCC workgroups: %s' %
              (first_done, st.max(), np.bincount(xcc.astype(np.int64) & 15)[:8].tolist()))
        # per-XCC finish times
        for x in range(8):
            m = (xcc & 15) == x
            if m.any():
                print('     xcc %d: last end %.1f us' % (x, en[m].max()))


if __name__ == '__main__':
    main(sys.argv[1])
