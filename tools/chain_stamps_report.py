#!/usr/bin/env python3
"""Reads the [workgroup][16] 100 MHz timestamps written by the -DKN_ABLATION build of the whole-net kernel (tools/chain_stamps.sh)
and prints, per phase, the duration inside a workgroup (median / p90 / max over workgroups) and when the phase ended relative
to the earliest workgroup start (median / max): start skew, input load, each operator, output store."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).astype(np.int64)
grid, has_w = int(raw[0]), int(raw[1])
t = raw[2:2 + grid * 16].reshape(-1, 16)
w = raw[2 + grid * 16:].reshape(grid, 12, 16, 40) if has_w else None
live = t[:, 0] > 0
t = t[live]                                   # idle workgroups return before the first stamp
nph = int((t[0] > 0).sum())
t = t[:, :nph]
t0 = t[:, 0].min()
names = ['start', 'input'] + ['op%d' % i for i in range(nph - 3)] + ['output']
print('%d workgroups, %d stamps; times in us (10 ns clock)' % (t.shape[0], nph))
print('%-8s %28s   %22s' % ('phase', 'duration med / p90 / max', 'ended at med / max'))
for k in range(nph):
    end = (t[:, k] - t0) / 100.0
    if k == 0:
        print('%-8s %28s   %10.2f / %8.2f' % ('start', '', np.median(end), end.max()))
        continue
    dur = (t[:, k] - t[:, k - 1]) / 100.0
    print('%-8s %10.2f / %6.2f / %6.2f   %10.2f / %8.2f' % (names[k], np.median(dur), np.percentile(dur, 90), dur.max(), np.median(end), end.max()))

if w is not None:
    # inside the walks of ONE workgroup (the first live one): per layer and wavefront, microseconds since the layer's phase began
    # (stamps with a forced wait: 1 = slice / lane records landed, 2 = first ring landed; 3 = first slice done, 6 = walk done, 7 = past the barrier)
    g = int(np.nonzero(live)[0][0])
    print('walks of workgroup %d: us since the phase began, per wavefront: records landed / first ring landed / first slice done / walk done / past barrier' % g)
    for l in range(nph - 3):
        base = t[0, 1 + l]
        print(' op%d' % l)
        for wv in range(16):
            x = w[g, l, wv]
            if x[0] == 0 and x[6] == 0:
                continue
            f = lambda k: ('%6.2f' % ((x[k] - base) / 100.0)) if x[k] else '     -'
            print('   wave %2d: enter %s | %s %s %s | done %s barrier %s' % (wv, f(0), f(1), f(2), f(3), f(6), f(7)))

    print('slices of the column-patterns-in-LDS walk (workgroup %d): per wavefront and slice, us since the phase began: start / first activations landed / arithmetic issued' % g)
    for l in range(nph - 3):
        base = t[0, 1 + l]
        if not w[g, l, :, 8:].any():
            continue
        print(' op%d' % l)
        for wv in range(16):
            x = w[g, l, wv, 8:].reshape(8, 4)
            cells = ['%5.2f/%5.2f/%5.2f' % tuple((x[i, k] - base) / 100.0 for k in range(3)) for i in range(8) if x[i, 0]]
            print('   wave %2d: %s' % (wv, '  '.join(cells)))
