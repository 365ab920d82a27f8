#!/usr/bin/env python3
"""Start-up cost of an N-rank run, measured where it will run: N concurrent processes each key the workload on the host (what every rank
of `bench.py --gpus N` does before it touches its GPU), wall time per rank and for the slowest.

    python3 tools/time_startup.py --ranks 8 --workload vgg16
Host only (no GPU call).  DESIGN section 7 quotes its output; if the slowest rank needs more than ~60 s, key once and hand the arrays to
the ranks through the neutral archive (keynet_amd.io.save_keynet / load_keynet) instead."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
t0 = time.time()
import torch
t1 = time.time()
import bench
(sensor, knet, inshape, batch, desc, net) = bench.build_workload(%r, %d, exact='auto' if %r.startswith('vgg16') else None)
t2 = time.time()
print(json.dumps({'rank': %d, 'import_s': round(t1 - t0, 2), 'keying_s': round(t2 - t1, 2), 'threads': torch.get_num_threads()}))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ranks', type=int, default=8)
    ap.add_argument('--workload', default='vgg16')
    args = ap.parse_args()
    for n in sorted({1, args.ranks}):
        t0 = time.time()
        procs = [subprocess.Popen([sys.executable, '-c', CHILD % (ROOT, args.workload, r, args.workload, r)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                 for r in range(n)]
        outs = [json.loads([l for l in p.communicate()[0].splitlines() if l.startswith('{')][-1]) for p in procs]
        wall = time.time() - t0
        print(json.dumps({'concurrent_ranks': n, 'workload': args.workload, 'host_cores': os.cpu_count(), 'wall_s': round(wall, 1),
                          'keying_s_max': max(o['keying_s'] for o in outs), 'keying_s_min': min(o['keying_s'] for o in outs),
                          'import_s_max': max(o['import_s'] for o in outs), 'threads_per_rank': outs[0]['threads']}), flush=True)


if __name__ == '__main__':
    main()
