#!/usr/bin/env python3
"""Experiment: the keyed VGG-16 forward as TWO half-batch column windows on two HIP streams (the drain of one launch overlaps the
body of the other stream's launch) against the plain single-stream forward.  Same buffers (a half is a column window: ldx = 256,
n_vecs = 128), same kernels, bit-identical logits expected.

    python3 tools/ab_streams.py --rounds 5 [--split-from pool1_2]"""
import argparse
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from keynet_amd import _capi  # noqa: E402
from keynet_amd import sparse as ksp  # noqa: E402
from keynet_amd.layer import KeyedLayer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--split-from', default='pool1_2')
    args = ap.parse_args()
    (sensor, knet, inshape, batch, desc, net) = bench.build_workload('vgg16', 0)
    B = args.batch
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(1234)
    x = torch.randn((B,) + tuple(inshape), generator=g, device=dev)
    xc = sensor.fromtensor(x).encrypt().astensor()
    del x
    ref = knet.forward_linear(xc)
    torch.cuda.synchronize()
    children = list(knet._keynet.named_children())
    plan = []
    for (i, (name, c)) in enumerate(children):
        if isinstance(c, KeyedLayer):
            fuse = (i + 1 < len(children)) and isinstance(children[i + 1][1], torch.nn.ReLU)
            exact = bool(getattr(c, '_exact', True))
            if type(c.W) is ksp.SparseMatrix and not exact and c.W._dense_device_op(dev) is not None:
                op = c.W._dense_device_op(dev)
                ex = False
            else:
                op = c.W._device_op(dev)
                ex = exact or not isinstance(c.W, ksp.Conv2dTiledMatrix)
            flags = (_capi.KN_FLAG_RELU if (fuse or c.iskeyedrelu()) else 0) | (_capi.KN_FLAG_EXACT if ex else 0)
            plan.append((name, op, c.W.shape[0], flags))
    x0 = xc.t().contiguous()                                # [D0+1, B]
    bufs = [torch.empty((rows, B), device=dev) for (_, _, rows, _) in plan]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    split_at = [n for (n, _, _, _) in plan].index(args.split_from)
    H = B // 2

    def forward_single():
        s = torch.cuda.current_stream().cuda_stream
        src = x0
        for (k, (name, op, rows, flags)) in enumerate(plan):
            op.spmm(src.data_ptr(), B, B, bufs[k].data_ptr(), B, flags, s)
            src = bufs[k]
        return bufs[-1]

    def forward_two_streams():
        main_s = torch.cuda.current_stream()
        src = x0
        for k in range(split_at):                            # whole batch (kernels that need 256-wide tiles)
            (name, op, rows, flags) = plan[k]
            op.spmm(src.data_ptr(), B, B, bufs[k].data_ptr(), B, flags, main_s.cuda_stream)
            src = bufs[k]
        for st in streams:
            st.wait_stream(main_s)
        for (h, st) in enumerate(streams):
            s_in = src
            for k in range(split_at, len(plan)):
                (name, op, rows, flags) = plan[k]
                op.spmm(s_in.data_ptr() + 4 * H * h, B, H, bufs[k].data_ptr() + 4 * H * h, B, flags, st.cuda_stream)
                s_in = bufs[k]
        for st in streams:
            main_s.wait_stream(st)
        return bufs[-1]

    def forward_interleaved():
        """Same two streams, launches issued layer by layer (half 0 then half 1) so the host enqueues them alternately."""
        main_s = torch.cuda.current_stream()
        src = x0
        for k in range(split_at):
            (name, op, rows, flags) = plan[k]
            op.spmm(src.data_ptr(), B, B, bufs[k].data_ptr(), B, flags, main_s.cuda_stream)
            src = bufs[k]
        for st in streams:
            st.wait_stream(main_s)
        for k in range(split_at, len(plan)):
            (name, op, rows, flags) = plan[k]
            s_in = src if k == split_at else bufs[k - 1]
            for (h, st) in enumerate(streams):
                op.spmm(s_in.data_ptr() + 4 * H * h, B, H, bufs[k].data_ptr() + 4 * H * h, B, flags, st.cuda_stream)
        for st in streams:
            main_s.wait_stream(st)
        return bufs[-1]

    def forward_offset(delay_layers):
        """Stream B starts `delay_layers` kernels behind stream A, so that the two streams' launch boundaries (drain of one kernel,
        fill of the next) never coincide: one stream's queued workgroups take over the CUs the other one's draining kernel frees."""
        def f():
            main_s = torch.cuda.current_stream()
            src = x0
            for k in range(split_at):
                (name, op, rows, flags) = plan[k]
                op.spmm(src.data_ptr(), B, B, bufs[k].data_ptr(), B, flags, main_s.cuda_stream)
                src = bufs[k]
            for st in streams:
                st.wait_stream(main_s)
            ev = None
            for k in range(split_at, len(plan) + delay_layers):
                for (h, st) in enumerate(streams):
                    kk = k - (delay_layers if h == 1 else 0)
                    if kk < split_at or kk >= len(plan):
                        continue
                    (name, op, rows, flags) = plan[kk]
                    s_in = src if kk == split_at else bufs[kk - 1]
                    op.spmm(s_in.data_ptr() + 4 * H * h, B, H, bufs[kk].data_ptr() + 4 * H * h, B, flags, st.cuda_stream)
                if k - split_at < delay_layers:              # hold stream B back until A has finished this kernel
                    ev = torch.cuda.Event()
                    ev.record(streams[0])
                    streams[1].wait_event(ev)
            for st in streams:
                main_s.wait_stream(st)
            return bufs[-1]
        return f

    variants = [('two streams, B one kernel behind', forward_offset(1)), ('two streams, B two kernels behind', forward_offset(2)),
                ('single stream', forward_single), ('two streams (half after half)', forward_two_streams), ('two streams (interleaved issue)', forward_interleaved)]
    for (nm, f) in variants:
        out = f()
        torch.cuda.synchronize()
        print('%-34s equals forward_linear bit for bit: %s' % (nm, bool(torch.equal(out.t(), ref))), flush=True)
    times = {nm: [] for (nm, _) in variants}
    for r in range(args.rounds):
        for (nm, f) in variants:
            for _ in range(2):
                f()
            torch.cuda.synchronize()
            (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            e0.record()
            for _ in range(4):
                f()
            e1.record()
            torch.cuda.synchronize()
            times[nm].append(e0.elapsed_time(e1) / 4)
    for (nm, _) in variants:
        t = np.array(times[nm])
        print('%-34s median %8.3f ms  min %8.3f ms  -> %7.1f images/s' % (nm, np.median(t), t.min(), B / np.median(t) * 1e3), flush=True)


if __name__ == '__main__':
    main()
