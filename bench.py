#!/usr/bin/env python3
"""bench.py -- keyed forward throughput on MI355X (BASELINE.json metric: encrypted images/sec + roofline, keyed VGG-16 224x224).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path (KeyedModel.forward_linear: all keyed layers, ReLU fused) over one batch of
synthetic encrypted images already resident in HBM.  Default workload = BASELINE.json configs[3]:
    TiledPermutationKeynet(VGG16(num_classes=2622), (3,224,224), tile 64 -> effective 56/28/14/7), 256 images per GPU.
Random-init weights (torch.manual_seed(0)), keys from np.random.seed(0), images ~ N(0,1) (there is no network for
checkpoints or datasets).  With N GPUs every rank runs its own 256-image shard (weak scaling, no collective inside the
forward) and the step ends with ONE RCCL all-gather of the logits.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (convtaps_mfma_kernel, f32 MFMA bound: SURVEY 8d);
`cpu_baseline` is the CPU oracle (oracle/: C restatement of scipy's csr_matvecs, 1 thread as the reference runs it) timed
on a bounded sample of the same workload and extrapolated by non-zeros (labelled).  `parity` is this run's own gate: the
logits of the timed batch's first images against the source network in plain torch f32 (the reference's criterion, atol 1e-3);
a run that fails it raises instead of printing a number.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from keynet_amd import system as ksys          # noqa: E402
from keynet_amd import sparse as ksp           # noqa: E402
from keynet_amd import io as kio               # noqa: E402
from keynet_amd import dist as kdist           # noqa: E402
from keynet_amd.layer import KeyedLayer        # noqa: E402
from keynet_amd.models import VGG16, LeNet_AvgPool, AllConvNet   # noqa: E402
from keynet_amd.torch import affine_to_linear  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md:42 (dense f32-input MFMA = f32 vector peak)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md:36 (spec; 6.29 TB/s measured copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_workload(name, rank, world, exact=None):
    """(sensor, knet, inshape, per_gpu_batch, description, source network).  Deterministic under the seeds, identical on every rank."""
    t0 = time.time()
    if name == 'vgg16':
        torch.manual_seed(0)
        net = VGG16(num_classes=2622).eval()
        np.random.seed(0)
        (sensor, knet) = ksys.TiledPermutationKeynet((3, 224, 224), net, 64, exact=exact)
        (inshape, batch, desc) = ((3, 224, 224), 256, 'TiledPermutationKeynet VGG16(2622) 3x224x224 tile=64 (effective 56/28/14/7)')
    elif name == 'lenet':
        torch.manual_seed(0)
        net = LeNet_AvgPool().eval()
        np.random.seed(0)
        (sensor, knet) = ksys.PermutationKeynet((1, 28, 28), net)
        (inshape, batch, desc) = ((1, 28, 28), 1024, 'PermutationKeynet LeNet_AvgPool 1x28x28')
    elif name == 'allconv':
        torch.manual_seed(0)
        net = AllConvNet(batchnorm=False).eval()
        np.random.seed(0)
        (sensor, knet) = ksys.PermutationKeynet((3, 32, 32), net)
        (inshape, batch, desc) = ((3, 32, 32), 4096, 'PermutationKeynet AllConvNet 3x32x32 (BASELINE configs[2])')
    else:
        raise ValueError('unknown workload "%s"' % name)
    log('[bench rank %d] keyed %s on the host in %.1f s' % (rank, name, time.time() - t0))
    return (sensor, knet, inshape, batch, desc, net)


def _takes_small_k_kernel(W, batch):
    """Mirror of the dispatch in kn_conv.hip (convtaps_spmm): one output pixel's whole contraction fits 28 rows (VGG conv1_1),
    which runs in the write-bound convtaps_smallk_kernel and is therefore not part of the MFMA roofline aggregate."""
    t = getattr(W, '_taps', None)
    if t is None or batch % 256:
        return False
    return int(np.bincount(t['ent_out']).max()) * W._inshape[0] + (1 if t['lastcol'] is not None else 0) <= 28


def layer_table(knet, batch):
    """Per keyed layer: algorithmic MACs (= nnz of the expanded operator the reference applies) and bytes."""
    rows = []
    children = list(knet._keynet.named_children())
    for (i, (name, c)) in enumerate(children):
        if not isinstance(c, KeyedLayer):
            continue
        if type(c.W) is ksp.SparseMatrix and not getattr(c, '_exact', True) and c.W._dense_device_op() is not None:
            kind = 'dense'                     # keyed nn.Linear on the split-K MFMA path (tolerance mode); its CSR twin is never built
            (r, cdim) = c.W.shape
            nnz_exp = int(c.W.nnz())
            wbytes = 4 * r * cdim
        else:
            op = c.W._device_op()
            (r, cdim) = op.shape()
            nnz_exp = op.nnz_expanded()
            if isinstance(c.W, ksp.Conv2dTiledMatrix):
                kind = 'smallk' if (_takes_small_k_kernel(c.W, batch) and not getattr(c, '_exact', False)) else 'convtaps'
                wbytes = 4 * c.W.nnz()        # taps + entries + last column actually read
            else:
                kind = 'csr'
                wbytes = 8 * nnz_exp           # (col,val) per non-zero
        rows.append(dict(name=name, kind=kind, rows=r, cols=cdim, nnz=nnz_exp, flops=2.0 * nnz_exp * batch,
                         bytes=float(wbytes) + 4.0 * batch * (r + cdim), layer=c,
                         fuse=(i + 1 < len(children)) and isinstance(children[i + 1][1], torch.nn.ReLU)))
    return rows


def time_layers(knet, x_cipher, table, iters):
    """Per-layer kernel time with HIP events on the launch stream (torch's current stream is the one kn_spmm launches on).
    Every iteration is timed on its own and the MEDIAN is kept: a multi-GB output allocation can occasionally fall out of
    the caching allocator and cost tens of ms, which must not leak into a kernel's average."""
    y = x_cipher
    for row in table:
        c = row['layer']
        xin = y
        y = None
        out = c.forward(xin, fuse_relu=row['fuse'])       # warm
        torch.cuda.synchronize()
        times = []
        for _ in range(max(iters, 1)):
            del out
            (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            e0.record()
            out = c.forward(xin, fuse_relu=row['fuse'])
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        row['ms'] = float(np.median(times))
        row['ms_min'] = float(np.min(times))
        y = out
    return table


def committed_traffic(workload):
    """HBM bytes per forward of the dominant kernel from the committed PMC passes (profiles/rNN_<workload>_*_traffic.json:
    separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same bench, FETCH doubled per the guide's gfx950
    note).  bench.py cannot collect PMC counters on itself; None when no such file is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_%s_*traffic.json' % workload)))
    if not files:
        return (None, None)
    t = json.load(open(files[-1]))
    return (t.get('convtaps_hbm_bytes_per_forward'), os.path.relpath(files[-1], ROOT))


def cpu_baseline(knet, batch_total_nnz, budget_cols=256, conv_pixels=1024):
    """CPU oracle (oracle/kn_oracle.c: scipy csr_matvecs restated, 1 thread) on a bounded sample of the workload:
    `conv_pixels` output pixels x all output channels of the largest conv layer (realistic gather pattern), the first
    pooling layer and the last two FC layers, `budget_cols` images (about 10-20 s of single-thread CPU work on the VGG-16
    workload); extrapolated to the whole net by non-zeros."""
    import oracle
    import scipy.sparse
    sample = []
    convs = [(n, c) for (n, c) in knet._keynet.named_children() if isinstance(c, KeyedLayer) and isinstance(c.W, ksp.Conv2dTiledMatrix) and c.W._taps is not None]
    rng = np.random.RandomState(0)
    if convs:
        (name, c) = max(convs, key=lambda nc: nc[1].W.shape[0])
        t = c.W._taps
        (Cout, Hout, Wout) = c.W._outshape
        (Cin, Hin, Win) = c.W._inshape
        pix = np.sort(rng.choice(Hout * Wout, size=min(conv_pixels, Hout * Wout), replace=False))
        sel = np.isin(t['ent_out'], pix)
        remap = -np.ones(Hout * Wout, dtype=np.int64)
        remap[pix] = np.arange(len(pix))
        (ic, jc) = np.meshgrid(np.arange(Cout), np.arange(Cin), indexing='ij')
        rows = (remap[t['ent_out'][sel]][:, None, None] + (ic * len(pix))[None]).ravel()
        cols = (t['ent_in'][sel].astype(np.int64)[:, None, None] + (jc * Hin * Win)[None]).ravel()
        vals = t['taps'][t['ent_tap'][sel]].ravel()
        M = scipy.sparse.csr_matrix((vals, (rows, cols)), shape=(Cout * len(pix), c.W.shape[1]))
        sample.append((name + '[%d px]' % len(pix), M))
    others = [(n, c) for (n, c) in knet._keynet.named_children() if isinstance(c, KeyedLayer) and not isinstance(c.W, ksp.Conv2dTiledMatrix)]
    pools = [(n, c) for (n, c) in others if isinstance(c.W, ksp.TiledMatrix)]
    fcs = [(n, c) for (n, c) in others if not isinstance(c.W, ksp.TiledMatrix)]
    for (n, c) in pools[:1]:
        sample.append((n, c.W.tocsr()))
    for (n, c) in fcs[-2:]:
        sample.append((n, c.W._matrix.tocsr()))
    if not convs:       # small nets: every layer
        sample = [(n, (c.W.tocsr() if isinstance(c.W, ksp.TiledMatrix) else c.W._matrix.tocsr())) for (n, c) in knet._keynet.named_children() if isinstance(c, KeyedLayer)]
    (macs, secs) = (0.0, 0.0)
    prepared = []
    for (n, M) in sample:
        X = rng.randn(M.shape[1], budget_cols).astype(np.float32)
        (ip, ix, dt) = (M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float32))
        oracle.csr_matvecs(M.shape, ip, ix, dt, X[:, :1])        # page in
        t0 = time.perf_counter()
        oracle.csr_matvecs(M.shape, ip, ix, dt, X)
        dt_s = time.perf_counter() - t0
        macs += float(M.nnz) * budget_cols
        secs += dt_s
        prepared.append((M.shape, ip, ix, dt, X))
        log('[bench cpu] %-22s nnz=%10d  %.3f s  %.3f ns/(nz*col)' % (n, M.nnz, dt_s, 1e9 * dt_s / (M.nnz * budget_cols)))
    ns_per_mac = 1e9 * secs / macs
    img_s = 1.0 / (ns_per_mac * 1e-9 * batch_total_nnz)
    names = ', '.join(n for (n, _) in sample)
    res = dict(value=img_s, unit='images/s', cores=1, kind='port',
               sample='oracle csr_matvecs (1 thread) on {%s} x %d images = %.3g MAC in %.1f s; %.3f ns/(nz*image) extrapolated to %.4g nnz/image'
                      % (names, budget_cols, macs, secs, ns_per_mac, batch_total_nnz))
    # second figure (SURVEY 8d ii): the same sample with the batch columns sharded over all host cores (one process per
    # core, operators shared copy-on-write) -- what a "whole host" deployment of the reference's algorithm could do
    try:
        import multiprocessing as mp
        P = max(1, min(os.cpu_count() or 1, budget_cols // 8, 32))     # >= 8 batch columns per process, else the operator stream dominates
        if P > 1:
            global _CPU_SHARDS
            _CPU_SHARDS = prepared
            ctx = mp.get_context('fork')          # children only run the C oracle on CPU arrays; they never touch the GPU
            with ctx.Pool(P) as pool:
                pool.map(_cpu_shard, [(k, P, 1) for k in range(P)])            # warm
                t0 = time.perf_counter()
                pool.map(_cpu_shard, [(k, P, 0) for k in range(P)])
                par = time.perf_counter() - t0
            res['all_cores'] = dict(value=1.0 / (par / macs * batch_total_nnz), unit='images/s', cores=P,
                                    sample='same sample, %d processes x %d columns each, %.2f s wall' % (P, (budget_cols + P - 1) // P, par))
    except Exception as e:       # never let the reported-only baseline break the bench line
        res['all_cores'] = dict(value=None, error=str(e))
    return res


_CPU_SHARDS = None


def _cpu_shard(arg):
    import oracle
    (k, P, warm) = arg
    for (shape, ip, ix, dt, X) in _CPU_SHARDS:
        cols = np.array_split(np.arange(X.shape[1]), P)[k]
        if len(cols) == 0:
            continue
        Xs = np.ascontiguousarray(X[:, cols[:1] if warm else cols])
        oracle.csr_matvecs(shape, ip, ix, dt, Xs)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='vgg16', choices=['vgg16', 'lenet', 'allconv'])
    ap.add_argument('--batch', type=int, default=None, help='images per GPU (default: the BASELINE config)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--layer-iters', type=int, default=5)
    ap.add_argument('--exact', action='store_true', help='bit-exact mode for the tiled key-nets (order-preserving kernels everywhere; not the headline)')
    ap.add_argument('--graph', action='store_true', help='replay the forward from a captured HIP graph (launch-bound small nets)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    # Test-only override (single-GPU boxes): KN_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo, so the N>1
    # control flow (sharding, barriers, gather, max-over-ranks timing) can be smoke-tested without 8 GPUs.  Never set by
    # the driver; the real path is one rank per GPU over RCCL.
    share = os.environ.get('KN_BENCH_SHARE_GPU') == '1'
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if share:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm
    assert args.gpus == world, '--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)' % (args.gpus, world)

    (sensor, knet, inshape, batch, desc, net) = build_workload(args.workload, rank, world, exact=True if args.exact else None)
    if args.exact:
        desc += ' [exact mode: order-preserving kernels, bit-exact with the reference algorithm]'
    batch = args.batch if args.batch is not None else batch

    # synthetic encrypted batch, resident in HBM before the timed region
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn((batch,) + tuple(inshape), generator=g, device=dev)
    x_cipher = sensor.fromtensor(x).encrypt().astensor()        # [B, D0+1] view of a feature-major block
    # parity gate of this run: the reference's own integration criterion (test/test_keynet.py:94,112) -- keyed logits
    # equal the source network's on the same plain images -- evaluated on the first images of the timed batch, plain torch f32 on the host
    n_gate = min(4, batch)
    with torch.no_grad():
        y_plain = net(x[:n_gate].cpu()).reshape(n_gate, -1) if rank == 0 else None
    del x
    t0 = time.time()
    y = knet.forward_linear(x_cipher)                            # first call uploads the operators
    torch.cuda.synchronize()
    log('[bench rank %d] operators resident + first forward in %.1f s; logits %s' % (rank, time.time() - t0, tuple(y.shape)))

    replay = knet.capture(x_cipher) if args.graph else None

    def step():
        yl = (replay(x_cipher) if replay is not None else knet.forward_linear(x_cipher))[:, :-1]
        if world == 1:
            return yl
        if share:   # gloo has no device all_gather_into_tensor: bounce through the host (test-only path)
            return kdist.gather_logits(yl.cpu(), total=batch * world).to(dev)
        return kdist.gather_logits(yl, total=batch * world)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device('cpu') if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert out.shape[0] == batch * world and bool(torch.isfinite(out).all())
    parity = None
    if rank == 0:
        err = float((out[:n_gate].contiguous().cpu() - y_plain).abs().max())   # contiguous first: a strided D2H copy is ~1500 tiny copies
        parity = {'check': 'keyed logits of the timed batch vs the source network (torch f32, host) on %d images' % n_gate, 'max_abs_err': err,
                  'max_abs_logit': float(y_plain.abs().max()), 'err_over_max_logit': err / max(float(y_plain.abs().max()), 1e-30), 'atol': 1e-3,
                  'ok': bool(err <= 1e-3)}
        if not parity['ok']:
            raise AssertionError('parity gate failed: %s' % json.dumps(parity))

    if rank == 0:
        table = layer_table(knet, batch)
        table = time_layers(knet, x_cipher, table, args.layer_iters)
        nnz_img = float(sum(r['nnz'] for r in table))
        for r in table:
            log('[bench layer] %-8s %-8s rows=%8d nnz=%12d  %8.3f ms  %7.2f TFLOP/s  %8.1f GB/s(alg)' %
                (r['name'], r['kind'], r['rows'], r['nnz'], r['ms'], r['flops'] / r['ms'] / 1e9, r['bytes'] / r['ms'] / 1e6))
        dom = [r for r in table if r['kind'] == 'convtaps'] or table
        dom_kind = 'mfma' if dom[0]['kind'] == 'convtaps' else 'hbm'
        dom_ms = sum(r['ms'] for r in dom)
        if dom_kind == 'mfma':
            ach = sum(r['flops'] for r in dom) / dom_ms / 1e9
            (traffic, tsrc) = committed_traffic(args.workload) if batch == 256 else (None, None)
            roof = dict(bound='mfma', kernel='convtaps_mfma_kernel (%d launches/forward)' % len(dom), achieved=ach, peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s',
                        frac=ach / PEAK_F32_MFMA_TFLOPS, traffic=traffic, traffic_unit='bytes/forward (PMC, offline pass)', traffic_source=tsrc,
                        algorithmic_bytes=sum(r['bytes'] for r in dom), algorithmic_flops=sum(r['flops'] for r in dom), ms_per_forward=dom_ms)
        else:
            ach = sum(r['bytes'] for r in dom) / dom_ms / 1e6
            roof = dict(bound='hbm', kernel='csr_group_kernel/csr_rows_kernel', achieved=ach, peak=PEAK_HBM_GBS, unit='GB/s', frac=ach / PEAK_HBM_GBS,
                        traffic=None, ms_per_forward=dom_ms)
        total_bytes = sum(r['bytes'] for r in table)
        ms_per_step = 1e3 * elapsed / args.steps
        res = {
            'metric': 'encrypted images/sec (whole node), keyed %s' % {'vgg16': 'VGG-16 224x224', 'lenet': 'LeNet_AvgPool 28x28', 'allconv': 'AllConvNet 32x32'}[args.workload],
            'value': batch * world * args.steps / elapsed, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': desc, 'images_per_gpu': batch, 'global_batch': batch * world, 'nnz_per_image': nnz_img,
                       'parallelism': 'batch shards x%d, all_gather(logits)' % world if world > 1 else 'single GPU'},
            'achieved_hbm_gbs_algorithmic': total_bytes / (ms_per_step * 1e6), 'achieved_tflops_algorithmic': 2.0 * nnz_img * batch / (ms_per_step * 1e9),
            'roofline': roof, 'parity': parity,
        }
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(knet, nnz_img)
        else:
            res['cpu_baseline'] = None
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
