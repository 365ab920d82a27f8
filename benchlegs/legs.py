"""Reported-only side legs of the default bench line: BASELINE configs[1] / [2] as child benches, plaintext-to-logits, and the N > 1 collective record."""
import hashlib
import json
import os
import subprocess
import sys
import time

import torch
import torch.distributed as dist

from keynet_amd import dist as kdist
from .common import ROOT, log, _pick


def run_secondary(args):
    """BASELINE configs[1] and [2] in the driver's line: LeNet_AvgPool B=1024 and AllConvNet B=4096 each run as a CHILD process of this
    bench (its own host phase, scipy baseline, device phase, oracle parity) BEFORE this process touches the GPU; the child's JSON line is
    condensed into `secondary`.  (A child process, not an exec: the parent goes on to the VGG legs.)"""
    out = {}
    # (LeNet: a forward is 37 us -- three warm-up steps are 0.1 ms, not enough for the GPU to leave its idle clock: 2 000 warm-up steps = 75 ms)
    for (wl, steps, warm, extra) in (('lenet', max(args.steps, 200), max(args.warmup, 2000), ['--graph-leg']), ('allconv', max(args.steps, 10), max(args.warmup, 3), [])):
        t0 = time.time()
        child_detail = os.path.join(ROOT, 'bench_detail_%s.json' % wl)
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', wl, '--steps', str(steps), '--warmup', str(warm), '--layer-iters', '3',
               '--no-secondary', '--cpu-budget', '8', '--detail', child_detail] + extra
        env = dict(os.environ)
        for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
            env.pop(k, None)
        try:
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
            lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
            for l in p.stderr.splitlines():
                if l.startswith('[bench'):
                    log('  [%s] %s' % (wl, l))
            if p.returncode != 0 or len(lines) != 1:
                out[wl] = {'error': 'child exited with %d' % p.returncode, 'stderr_tail': p.stderr[-800:]}
                continue
            r = json.loads(lines[0])                      # the child's compact line; its full record is in its own detail file
            cpu = r.get('cpu_baseline') or {}
            out[wl] = {'workload': r['config']['workload'], 'images_per_gpu': r['config']['images_per_gpu'], 'images_per_s': r['value'], 'ms_per_step': r['ms_per_step'],
                       'steps': r['steps'], 'warmup': r['warmup'], 'roofline': r['roofline'],
                       'parity': {'bit_equal': (r.get('parity') or {}).get('oracle_bit_equal'), 'check': 'logits of the timed batch vs the CPU oracle run through every layer on the first 8 images'},
                       'parity_vs_source_network': _pick(r.get('parity') or {}, ('ok', 'max_abs_err', 'atol')),
                       'cpu_baseline': cpu, 'detail': r.get('detail'), 'child_wall_s': time.time() - t0}
            try:
                out[wl]['full'] = json.load(open(child_detail))
            except Exception:
                pass
        except Exception as e:      # a reported-only section must never break the headline
            out[wl] = {'error': str(e)}
    return out


def end_to_end(sensor, knet, x_plain, steps, warmup):
    """Plaintext -> logits (keynet/system.py:250-255 + 130-133): sensor.fromtensor(x).encrypt() -- homogenise on the device
    (kn_affine_to_linear) and apply the image key (the same SpMM primitive) -- inside the timed loop, then the keyed forward."""
    def step():
        return knet.forward_linear(sensor.fromtensor(x_plain).encrypt().astensor())
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        y = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    e0.record()
    for _ in range(5):
        xc = sensor.fromtensor(x_plain).encrypt().astensor()
    e1.record()
    torch.cuda.synchronize()
    del y, xc
    n = x_plain.shape[0]
    return {'images_per_s': n * steps / el, 'ms_per_step': 1e3 * el / steps, 'steps': steps, 'encrypt_ms': e0.elapsed_time(e1) / 5,
            'what': 'sensor.fromtensor(x_plain).encrypt() + forward_linear per step, plaintext batch resident in HBM'}


def collective_record(knet, sensor, x_cipher, gathered, batch, world, rank, local_rank, dev, inshape, share):
    """What the N>1 line says about itself (every rank takes part; rank 0 keeps the record): the ranks and devices that were really
    there, the cost of the logits all-gather alone (HIP events on the launch stream), and two bit-level checks of the gathered block --
    every rank's own shard against its local forward, and the LAST rank's shard recomputed on rank 0 from that rank's input seed
    (weights are replicated and batch columns independent, so a single process must reproduce any shard bit for bit)."""
    info = {'rank': rank, 'local_rank': local_rank, 'device_index': dev.index, 'device_name': torch.cuda.get_device_name(dev), 'pid': os.getpid()}
    infos = [None] * world
    dist.all_gather_object(infos, info)
    yl = knet.forward_linear(x_cipher)[:, :-1].contiguous()
    for _ in range(3):
        kdist.gather_logits(yl, total=batch * world)
    torch.cuda.synchronize()
    dist.barrier()
    n_calls = 20
    (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n_calls):
        g = kdist.gather_logits(yl, total=batch * world)
    e1.record()
    torch.cuda.synchronize()
    wall_ms = 1e3 * (time.perf_counter() - t0) / n_calls
    ev_ms = e0.elapsed_time(e1) / n_calls
    own = bool(torch.equal(g[rank * batch:(rank + 1) * batch], yl)) and bool(torch.equal(gathered[rank * batch:(rank + 1) * batch], yl))
    flag = torch.tensor([1 if own else 0], dtype=torch.int32, device=torch.device('cpu') if share else dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    peer = world - 1
    peer_equal = None
    if rank == 0:
        gp = torch.Generator(device=dev).manual_seed(1234 + peer)
        xp = torch.randn((batch,) + tuple(inshape), generator=gp, device=dev)
        yp = knet.forward_linear(sensor.fromtensor(xp).encrypt().astensor())[:, :-1]
        peer_equal = bool(torch.equal(g[peer * batch:(peer + 1) * batch], yp))
        del xp, yp
    return {'backend': dist.get_backend(), 'ranks_seen': dist.get_world_size(), 'ranks': infos,
            'op': 'all_gather_into_tensor of [%d, %d] f32 logits per rank' % (batch, yl.shape[1]), 'bytes_per_rank': int(yl.numel() * 4),
            'ms_per_call': ev_ms, 'ms_per_call_wall': wall_ms, 'calls_timed': n_calls,
            'every_rank_shard_bit_equal_to_its_local_forward': bool(flag.item() == 1), 'rank0_shard_bit_equal': own if rank == 0 else None,
            'rank0_shard_sha256': hashlib.sha256(yl.cpu().numpy().tobytes()).hexdigest() if rank == 0 else None,
            'peer_shard_recomputed_on_rank0': {'peer_rank': peer, 'bit_equal': peer_equal}}
