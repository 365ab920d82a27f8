#!/usr/bin/env python3
"""bench.py -- keyed forward throughput on MI355X (BASELINE.json metric: encrypted images/sec + roofline, keyed VGG-16 224x224).

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts N ranks itself, one per GPU, over RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path (KeyedModel.forward_linear: all keyed layers, ReLU fused) over one batch of
synthetic encrypted images already resident in HBM.  Default workload = BASELINE.json configs[3]:
    TiledPermutationKeynet(VGG16(num_classes=2622), (3,224,224), tile 64 -> effective 56/28/14/7), 256 images per GPU.
Random-init weights (torch.manual_seed(0)), keys from np.random.seed(0), images ~ N(0,1) (there is no network for
checkpoints or datasets).  With N GPUs every rank runs its own 256-image shard (weak scaling, no collective inside the
forward) and the step ends with ONE RCCL all-gather of the logits.

Rank 0 prints ONE JSON line.
  roofline      the dominant kernel of the mode that ran: convtaps_mfma_kernel against the f32-MFMA peak (tolerance mode of the
                tiled key-nets), the order-preserving kernels against the no-FMA VALU peak (--exact, AllConvNet), the CSR kernels
                against HBM (LeNet).  `traffic` (HBM bytes per forward of that kernel from a separate rocprofv3 --pmc pass) is
                quoted only when the committed pass was taken on THESE kernel sources (sha256 of keynet_amd/csrc recorded with it).
  exact         (default vgg16 run, N=1) the SAME key-net switched to the bit-exact contract (KeyedModel.exact_mode(True): every
                layer in the reference's accumulation order, no MFMA): images/s, ms/step, its own roofline (VALU without FMA,
                39.3 T MAC/s) and a parity record -- bit-equality with the CPU oracle on sampled rows of real conv layers.
  cpu_baseline  the reference's own arithmetic -- scipy.sparse.csr_matrix.dot (keynet/sparse.py:488-492) -- timed on this node's
                host cores BEFORE the GPU is touched (the worker pool is forked from a GPU-free process): conv1_1, conv5_1, every
                pool, fc6-8 measured directly on 1 thread and on all physical cores; the other layers extrapolated at the measured
                ns/(nz*column) and labelled so; plus the reference-faithful tocsr()+dot of one tiled layer (keynet/sparse.py:603-612).
  parity        this run's own gate: logits of the timed batch's first images against the source network in plain torch f32
                (the reference's criterion, atol 1e-3); a run that fails it raises instead of printing a number.
"""
import argparse
import json
from types import SimpleNamespace
import os
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from keynet_amd import dist as kdist           # noqa: E402

# what the timed path below calls; everything else lives in benchlegs/
from benchlegs.common import log, kernel_sources_sha                                                                    # noqa: E402
from benchlegs.shared import build_workload_shared                                                                      # noqa: E402
from benchlegs.cpu import cpu_baseline                                                                                  # noqa: E402
from benchlegs.layers import layer_table, time_layers, roofline_of, chain_roofline                                      # noqa: E402
from benchlegs.parity import oracle_parity_csr                                                                          # noqa: E402
from benchlegs.legs import run_secondary, collective_record                                                             # noqa: E402
from benchlegs.sidelegs import run_side_legs                                                                            # noqa: E402
from benchlegs.line import compact_record, write_detail, DETAIL_FILE                                                    # noqa: E402
# re-exported for the tools and tests that use `bench.<name>`
from benchlegs.common import PEAK_F32_MFMA_TFLOPS, PEAK_VALU_NOFMA_TMACS, PEAK_HBM_GBS, keyed_layers, host_nnz          # noqa: E402,F401
from benchlegs.workloads import build_workload                                                                          # noqa: E402,F401
from benchlegs.cpu import host_cores                                                                                    # noqa: E402,F401
from benchlegs.layers import committed_traffic                                                                          # noqa: E402,F401
from benchlegs.line import LINE_LIMIT                                                                                   # noqa: E402,F401


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (torch.distributed.run, one per GPU) from THIS
    process, which has not touched the GPU (no torch.cuda call above this point), and exit with their code."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    log('[bench] starting %d ranks: %s' % (args.gpus, ' '.join(cmd)))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--exact-layers-parity', action='store_true',
                    help='float-key workloads (vgg16-*): check the layers the contract keeps in the reference\'s order against the CPU oracle on one sampled output pixel each (OPT-IN: the host '
                         'expansion of a filled-in pixel is 40-90 M stored entries per layer -- minutes of host time at full size; the same check runs on reduced nets in tests/test_vgg16_families_gpu.py)')
    ap.add_argument('--workload', default='vgg16', choices=['vgg16', 'vgg16-gain', 'vgg16-givens', 'vgg16-givens28', 'vgg16-stochastic', 'vgg16-slice', 'lenet', 'allconv'])
    ap.add_argument('--batch', type=int, default=None, help='images per GPU (default: the BASELINE config)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--layer-iters', type=int, default=5)
    ap.add_argument('--exact', action='store_true', help='bit-exact mode for the tiled key-nets as the MAIN measurement (order-preserving kernels everywhere)')
    ap.add_argument('--no-exact-leg', action='store_true', help='skip the additional bit-exact-mode measurement of the default vgg16 run')
    ap.add_argument('--graph', action='store_true', help='replay the forward from a captured HIP graph (launch-bound small nets)')
    ap.add_argument('--dist', action='store_true', help='initialise the RCCL process group and all-gather the logits every step even with ONE rank '
                                                        '(exercises the multi-GPU code path on a single-GPU box)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the LeNet / AllConvNet legs of the default run')
    ap.add_argument('--graph-leg', action='store_true', help='additionally time the forward replayed from a captured HIP graph (reported as `graph`)')
    ap.add_argument('--cpu-budget', type=float, default=24.0, help='seconds of scipy work for the CPU baseline sample')
    ap.add_argument('--experimental', action='store_true', help='additionally run the EXPERIMENTAL bf16x3 leg (never the headline; detail file only)')
    ap.add_argument('--detail', default=None, metavar='FILE', help='where the full record goes (default: bench_detail.json next to bench.py); stdout carries the compact line')
    ap.add_argument('--trace-layers', default=None, metavar='FILE', help='profiling aid (run under rocprofv3 --kernel-trace): after the first forward, launch every layer '
                                                                          '8 times back to back with a marker kernel between layers, write the layer list (name, kind, flops, bytes) to FILE and exit; '
                                                                          'tools/trace_layers.py joins it with the kernel trace into a per-layer table')
    ap.add_argument('--pmc-forward', default=None, metavar='FILE', help='profiling aid (run under rocprofv3 --kernel-trace --pmc <counter>): behind the calibrating forward and two warm ones, launch '
                                                                        'EXACTLY ONE forward_linear -- the timed step\'s launches, nothing else -- between two marker kernels, write FILE and exit; '
                                                                        'tools/pmc_forward.py sums the counter over the launches between the markers (`roofline.traffic`: ONE definition)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))                       # before ANY GPU call in this process
    # stdout carries exactly ONE line, the JSON record: native libraries (RCCL prints a version banner to fd 1) go to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    assert args.gpus == world, '--gpus %d but WORLD_SIZE=%d' % (args.gpus, world)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))

    secondary = None
    if args.workload == 'vgg16' and world == 1 and not args.no_secondary and not args.exact and not args.dist and args.batch is None:
        secondary = run_secondary(args)                   # children own the GPU one after the other; this process has not touched it yet

    # Test-only override (single-GPU boxes): KN_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo, so the N>1 control flow (shared keying, sharding,
    # barriers, gather, max-over-ranks timing) can be exercised without 8 GPUs.  Never set by the driver; the real path is one rank per GPU over RCCL.
    share = os.environ.get('KN_BENCH_SHARE_GPU') == '1'
    if share:
        local_rank = 0
    use_dist = world > 1 or args.dist
    dev = torch.device('cuda', local_rank)

    def init_dist():
        from datetime import timedelta
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if 'MASTER_PORT' not in os.environ:                  # --dist without a launcher: a rendezvous of one
            import socket
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
        if share:
            dist.init_process_group('gloo', rank=rank, world_size=world, timeout=timedelta(minutes=30))
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev, timeout=timedelta(minutes=30))   # "nccl" is RCCL on ROCm

    # (KN_BENCH_TEST_FORCE_SHARED=1 with --dist: test-only -- the hand-over protocol of benchlegs/shared.py with ONE rank, so that its collectives run on the RCCL backend of a
    # single-GPU box: tests/test_rccl_gpu.py; never set by the driver)
    if world > 1 or (args.dist and os.environ.get('KN_BENCH_TEST_FORCE_SHARED') == '1'):
        # N ranks: the process group first -- the ranks of a node key ONCE and hand the net on through it (benchlegs/shared.py); nothing forks at N > 1
        # (the scipy baseline is an N = 1 leg), so the GPU may be initialised before the host phase here
        assert torch.cuda.is_available(), 'bench.py needs an MI355X'
        torch.cuda.set_device(local_rank)
        init_dist()

    # ---- host phase: keying and the CPU baseline; at N = 1 nothing below touches the GPU until "device phase" -------------
    # Arithmetic contract of the headline: BASELINE configs[3] names "MFMA dense sub-tiles", so the tiled VGG-16 key-nets are built with the
    # 'auto' contract EXPLICITLY (matrix cores wherever the 1e-5 tolerance holds, screened on every forward); a permutation-only tiled key-net's
    # own default is the bit-exact contract, which the `exact` leg of the same line measures on the same key-net.
    t_key = time.time()
    (sensor, knet, inshape, batch, desc, net) = build_workload_shared(args.workload, rank, world, exact=True if args.exact else ('auto' if args.workload.startswith('vgg16') else None))
    t_key = time.time() - t_key
    mode = 'exact' if (args.exact or not args.workload.startswith('vgg16')) else 'tolerance'
    mode_desc = {'exact': 'exact: order-preserving kernels, bit-exact with the reference (the default of permutation-only key-nets)',
                 'tolerance': "tolerance, explicit opt-in exact='auto': f32 MFMA inside the reference's element-wise gate |d| <= 1e-5 + 1e-5 |ref|, re-screened every forward; "
                              "bit-exact contract = `exact` leg"}[mode]
    if args.exact:
        desc += ' [exact mode: order-preserving kernels, bit-exact with the reference algorithm]'
    batch = args.batch if args.batch is not None else batch
    cpu = None
    if world == 1 and not args.no_cpu_baseline and args.workload in ('vgg16', 'lenet', 'allconv'):      # (the scipy baseline is measured on the configs BASELINE.json quotes)
        t0 = time.time()
        cpu = cpu_baseline(knet, args.workload, budget_s=args.cpu_budget)
        log('[bench cpu] baseline section took %.1f s' % (time.time() - t0))

    # ---- device phase --------------------------------------------------------------------------------------------------------
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    torch.cuda.set_device(local_rank)
    if use_dist and not dist.is_initialized():
        init_dist()

    # synthetic encrypted batch, resident in HBM before the timed region
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn((batch,) + tuple(inshape), generator=g, device=dev)
    x_cipher = sensor.fromtensor(x).encrypt().astensor()        # [B, D0+1] view of a feature-major block
    # parity gate of this run: the reference's own integration criterion (test/test_keynet.py:94,112) -- keyed logits
    # equal the source network's on the same plain images -- evaluated on the first images of the timed batch, plain torch f32 on the host
    n_gate = min(4, batch)
    with torch.no_grad():
        y_plain = net(x[:n_gate].cpu()).reshape(n_gate, -1) if rank == 0 else None
    x_plain = x if (rank == 0 and world == 1) else None
    del x
    t0 = time.time()
    y = knet.forward_linear(x_cipher)                            # first call uploads the operators
    torch.cuda.synchronize()
    log('[bench rank %d] operators resident + first forward in %.1f s; logits %s' % (rank, time.time() - t0, tuple(y.shape)))

    if args.trace_layers:
        table = layer_table(knet, batch)
        marker = torch.zeros(1031, device=dev)
        yin = x_cipher
        for row in table:
            c = row['layer']
            marker.add_(1.0)                                   # one torch elementwise kernel = the separator tools/trace_layers.py splits on
            for _ in range(8):
                out = c.forward(yin, fuse_relu=row['fuse'])
            yin = out
        marker.add_(1.0)
        torch.cuda.synchronize()
        json.dump({'workload': desc, 'batch': batch, 'launches_per_layer': 8, 'csrc_sha256': kernel_sources_sha(),
                   'layers': [{k: r[k] for k in ('name', 'kind', 'rows', 'cols', 'nnz', 'flops', 'bytes', 'plan')} for r in table]}, open(args.trace_layers, 'w'), indent=1)
        log('[bench] layer list written to %s' % args.trace_layers)
        return

    if args.pmc_forward:
        for _ in range(2):
            knet.forward_linear(x_cipher)
        torch.cuda.synchronize()
        marker = torch.zeros(1031, device=dev)
        marker.add_(1.0)                                       # (one torch elementwise kernel each side: what tools/pmc_forward.py cuts on)
        knet.forward_linear(x_cipher)                          # the launches of ONE timed step
        marker.add_(1.0)
        torch.cuda.synchronize()
        table = layer_table(knet, batch)
        json.dump({'workload': desc, 'batch': batch, 'mode': mode, 'forwards_between_markers': 1, 'csrc_sha256': kernel_sources_sha(),
                   'algorithmic_bytes_per_forward': {k: sum(r['bytes'] for r in table if r['kind'] == k) for k in sorted({r['kind'] for r in table})},
                   'layers': [{k: r[k] for k in ('name', 'kind', 'rows', 'cols', 'nnz', 'flops', 'bytes', 'plan')} for r in table]}, open(args.pmc_forward, 'w'), indent=1)
        log('[bench] one forward between markers; layer list written to %s' % args.pmc_forward)
        return

    replay = knet.capture(x_cipher) if args.graph else None

    def step():
        if not use_dist:
            return (replay(x_cipher) if replay is not None else knet.forward_linear(x_cipher))[:, :-1]
        # N ranks: this rank's shard with the replicas' calibrated contracts kept in agreement (one tiny all-reduce per step for key-nets
        # under the 'auto' contract, none otherwise: keynet_amd.dist.replicated_forward), then ONE RCCL all-gather of the logits over xGMI
        # (gloo rigs bounce through the host)
        yl = replay(x_cipher)[:, :-1] if replay is not None else kdist.replicated_forward(knet, x_cipher)
        return kdist.gather_logits(yl, total=batch * world)

    rank_seconds = {}            # the last timed region's own time on every rank (before the closing barrier): the N > 1 line quotes min / max images/s per rank

    def timed(n_warm, n_steps):
        for _ in range(n_warm):
            step()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            out = step()
        torch.cuda.synchronize()
        own = time.perf_counter() - t0
        if use_dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if use_dist:
            # MAX over ranks of the barrier-bracketed time (every rank learns every rank's pair: the maximum is taken locally)
            t = torch.tensor([elapsed, own], dtype=torch.float64, device=torch.device('cpu') if share else dev)
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            elapsed = max(float(v[0].item()) for v in allt)
            rank_seconds['own'] = [float(v[1].item()) for v in allt]
            rank_seconds['steps'] = n_steps
        return (elapsed, out)

    (elapsed, out) = timed(args.warmup, args.steps)
    assert out.shape[0] == batch * world and bool(torch.isfinite(out).all())
    parity = None
    if rank == 0:
        err = float((out[:n_gate].contiguous().cpu() - y_plain).abs().max())   # contiguous first: a strided D2H copy is ~1500 tiny copies
        parity = {'check': 'keyed logits of the timed batch vs the source network (torch f32, host) on %d images' % n_gate, 'max_abs_err': err,
                  'max_abs_logit': float(y_plain.abs().max()), 'err_over_max_logit': err / max(float(y_plain.abs().max()), 1e-30), 'atol': 1e-3,
                  'ok': bool(err <= 1e-3)}
        if not parity['ok']:
            raise AssertionError('parity gate failed: %s' % json.dumps(parity))
    collective = None
    if use_dist:
        collective = collective_record(knet, sensor, x_cipher, out, batch, world, rank, local_rank, dev, inshape, share)
        per_rank = [batch * rank_seconds['steps'] / t for t in rank_seconds['own']]
        collective['rank_images_per_s'] = {'min': min(per_rank), 'max': max(per_rank), 'all': per_rank,
                                           'what': 'images of the rank\'s own shard / its own time over the timed steps (before the closing barrier; every step ends in the all-gather)'}
        collective['startup'] = {'keying_or_loading_s_rank0': t_key, 'what': 'rank 0 keys once and hands the net to the other ranks through an anonymous file (benchlegs/shared.py)'}
    oracle_par = None
    if rank == 0 and not args.workload.startswith('vgg16'):
        oracle_par = oracle_parity_csr(knet, x_cipher, out[:batch])
        if oracle_par.get('ok') is False:
            raise AssertionError('oracle parity failed: %s' % json.dumps(oracle_par))
    del out

    if rank == 0:
        table = time_layers(x_cipher, layer_table(knet, batch), args.layer_iters)
        nnz_img = float(sum(r['nnz'] for r in table))
        for r in table:
            log('[bench layer] %-8s %-9s rows=%8d nnz=%12d  %8.3f ms  %7.2f TFLOP/s  %8.1f GB/s(alg)%s' %
                (r['name'], r['kind'], r['rows'], r['nnz'], r['ms'], r['flops'] / r['ms'] / 1e9, r['bytes'] / r['ms'] / 1e6,
                 '' if not r.get('flops_executed') else '  (on the stored entries of the fused operator; the split application executes %.2f TFLOP/s)' % (r['flops_executed'] / r['ms'] / 1e9)))
        roof = roofline_of(table, args.workload, batch, mode)
        total_bytes = sum(r['bytes'] for r in table)
        chain = knet._chain_op(dev) if hasattr(knet, '_chain_op') else None
        if chain is not None:                # the forward of this key-net is ONE launch of the whole-net kernel: that launch is the dominant kernel
            roof = chain_roofline(knet, chain, x_cipher, table, batch)
        ms_per_step = 1e3 * elapsed / args.steps
        res = {
            'metric': 'encrypted images/sec (whole node), keyed %s' % {'vgg16': 'VGG-16 224x224', 'vgg16-gain': 'VGG-16 224x224 (float keys: permutation + photometric gain)', 'vgg16-givens': 'VGG-16 224x224 (float keys: Givens rotations + affine photometric, the reference\'s test_vgg16_orthogonal)', 'vgg16-givens28': 'VGG-16 224x224 (float keys: Givens rotations + affine photometric, tile 28: the reference\'s test_vgg16_orthogonal_8)', 'vgg16-stochastic': 'VGG-16 224x224 (float keys: hierarchical permutation + doubly-stochastic blocks + affine photometric: the reference\'s test_vgg16_stochastic)', 'vgg16-slice': 'VGG-16 slice 32x32 width 8 (rehearsal workload)', 'lenet': 'LeNet_AvgPool 28x28', 'allconv': 'AllConvNet 32x32'}[args.workload],
            'value': batch * world * args.steps / elapsed, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': desc, 'mode': mode_desc,
                       # which arithmetic contract the headline `value` was measured under, next to the library default's name (advisor, round 4): a
                       # permutation-only tiled key-net is bit-exact BY DEFAULT; the matrix-core number needs exact='auto' / False at construction
                       'headline_contract': ("opt-in exact='auto' (matrix cores within 1e-5, re-screened)" if mode == 'tolerance' else 'library default (bit-exact)'),
                       'default_contract_value': 'see `exact.images_per_s`' if mode == 'tolerance' else 'this value',
                       'images_per_gpu': batch, 'global_batch': batch * world, 'nnz_per_image': nnz_img,
                       'parallelism': 'batch shards x%d, all_gather(logits)' % world if world > 1 else 'single GPU'},
            'achieved_hbm_gbs_algorithmic': total_bytes / (ms_per_step * 1e6), 'achieved_tflops_algorithmic': 2.0 * nnz_img * batch / (ms_per_step * 1e9),
            'roofline': roof, 'parity': parity,
            'cpu_baseline': cpu if (cpu is not None or world == 1) else 'measured at N=1 only (the scipy baseline runs on rank 0 of a single-GPU run; see BENCH / profiles)',
            'collective': collective,
            'layers_ms': {r['name']: round(r['ms'], 4) for r in table},
            'plans': {r['name']: r['plan'] for r in table},        # kn_spmm_plan: the kernels (tile shape, loader) each layer really takes at this batch
        }
        if oracle_par is not None:
            res['oracle_parity'] = oracle_par
        if secondary is not None:
            res['secondary'] = secondary
        # ---- the headline record is complete here.  Every leg below is reported-only: it runs inside leg(), which turns a failure into
        # an entry of res['errors'], and the line is printed from the `finally` -- a late failure (an out-of-memory in a side leg, say)
        # can no longer lose the headline.
        res['errors'] = {}
        try:
            run_side_legs(SimpleNamespace(args=args, res=res, knet=knet, sensor=sensor, x_cipher=x_cipher, x_plain=x_plain, y_plain=y_plain, n_gate=n_gate, batch=batch, world=world,
                                          dev=dev, table=table, timed=timed, replay=replay))
        finally:
            if not res['errors']:
                del res['errors']
            detail = args.detail or os.path.join(ROOT, DETAIL_FILE)
            write_detail(res, args.detail)
            log('[bench detail] ' + json.dumps(res, default=str))
            os.write(json_fd, (compact_record(res, os.path.relpath(detail, ROOT) if detail.startswith(ROOT) else detail) + '\n').encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
