"""World-size-N harness shared by the CPU (gloo, stand-in operator) and the GPU (gloo, REAL KeyedModel on a shared cuda:0)
tests of the N>1 path: every rank builds the key-net, shards the same encrypted batch with keynet_amd.dist, runs its shard
and all-gathers the logits; each rank reports whether the gathered result equals the single-process forward bit for bit."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from keynet_amd import dist as kdist

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class StandInKeynet(object):
    """A KeyedModel stand-in for a GPU-less box: a fixed linear map with a homogeneous column (the sharding and the
    collective are what the CPU test covers; the kernels run in the GPU test through the same harness)."""
    def __init__(self):
        g = torch.Generator().manual_seed(0)
        self.M = torch.randn(13, 5, generator=g)

    def forward_linear(self, x):
        return torch.cat((x[:, :-1] @ self.M, x[:, -1:]), dim=1)


def make_case(kind, n):
    """(knet, full encrypted batch [n, D+1]) -- identical on every rank."""
    if kind == 'standin':
        g = torch.Generator().manual_seed(1)
        return (StandInKeynet(), torch.cat((torch.randn(n, 13, generator=g), torch.ones(n, 1)), dim=1))
    from keynet_amd import io as kio
    z = np.load(os.path.join(HERE, 'golden', {'lenet': 'lenet_perm.npz', 'tiled': 'mini_tiled_permutation.npz'}[kind]), allow_pickle=False)
    knet = kio.keynet_from_arrays(z)
    xc = torch.as_tensor(z['x_cipher'])
    reps = (n + xc.shape[0] - 1) // xc.shape[0]
    x = torch.cat([xc] * reps, dim=0)[:n].clone()
    x[:, :-1] += 0.01 * torch.arange(n, dtype=torch.float32)[:, None]       # distinct rows, so a mis-ordered gather cannot pass
    return (knet, x.to('cuda:0'))


def worker(rank, world_size, port, kind, n, q, backend='gloo'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend == 'nccl':                                      # RCCL: one rank per GPU, the communicator bound to the device up front
        torch.cuda.set_device(rank)
        dist.init_process_group('nccl', rank=rank, world_size=world_size, device_id=torch.device('cuda', rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world_size)
    try:
        calls = []
        real = dist.all_gather_into_tensor

        def counted(out, inp, *a, **k):
            calls.append((str(inp.device), tuple(inp.shape)))
            return real(out, inp, *a, **k)
        dist.all_gather_into_tensor = counted
        (knet, x) = make_case(kind, n)
        y = kdist.sharded_forward(knet, x)                     # this rank's shard through the real path + all-gather
        ref = knet.forward_linear(x)[:, :-1]                   # the single-process result of the SAME batch
        (lo, hi) = kdist.shard_bounds(n, rank, world_size)
        q.put((rank, bool(torch.equal(y, ref)), (lo, hi), tuple(y.shape), str(y.device), dist.get_backend(), calls))
    finally:
        dist.all_gather_into_tensor = real
        dist.destroy_process_group()


def contract_worker(rank, world_size, port, mode, q):
    """Replicated key-nets whose calibration decisions differ between ranks (each rank calibrates on ITS batches) must end up running the
    same kernels: KeyedModel.sync_contract (one all-reduce: the most conservative decision per layer wins).
    mode 'host' (CPU box): the decisions are planted, no forward runs.  mode 'device' (two ranks sharing cuda:0, real kernels): a float-key
    mini-net under the 'auto' contract; the LAST rank's images are 300x larger than the others', so only ITS data can trip a layer's tolerance screen."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    try:
        from keynet_amd import io as kio
        from keynet_amd.layer import KeyedLayer
        z = np.load(os.path.join(HERE, 'golden', 'mini_tiled_orthogonal.npz'), allow_pickle=False)
        if mode in ('host', 'host_split'):
            knet = kio.keynet_from_arrays(z)
            plant = [{'conv1': ('split', {'decided': 'split', 'max_abs_x': 2.0}), 'conv2': ('split', {'decided': 'split', 'max_abs_x': 3.0}), 'pool1': (True, None), 'fc1': (True, None), 'pool2': (True, None)},
                     {'conv1': (False, {'decided': 'mfma', 'max_abs_x': 5.0}), 'conv2': ('split', {'decided': 'split', 'max_abs_x': 9.0}), 'pool1': (True, None), 'fc1': (True, None), 'pool2': (True, None)}][rank] if mode == 'host_split' else [{'conv1': (False, {'decided': 'mfma', 'max_abs_x': 2.0}), 'conv2': ('bf16x3', {'decided': 'bf16x3', 'max_abs_x': 3.0}), 'pool1': (True, None), 'fc1': (True, None), 'pool2': (True, None)},
                     {'conv1': (True, {'decided': 'exact', 'bound': 1.0}), 'conv2': (False, {'decided': 'mfma', 'max_abs_x': 9.0}), 'pool1': (True, None), 'fc1': (True, None), 'pool2': (True, None)}][rank]
            for (n, c) in knet._keynet.named_children():
                if isinstance(c, KeyedLayer):
                    (c._exact, rec) = plant[n]
                    if rec is not None:
                        c._contract_record = rec
            changed = knet.sync_contract()
            state = {n: c._exact for (n, c) in knet._keynet.named_children() if isinstance(c, KeyedLayer)}
            again = knet.sync_contract()                       # agreed: a second round changes nothing anywhere
            q.put((rank, changed, state, again, knet.conv1.screened(), knet.conv2.screened()))
            return
        import warnings
        from keynet_amd import system as ksys
        from nets import MiniNet, load_weights
        net = load_weights(MiniNet(), z)
        np.random.seed(0)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            (sensor, knet) = ksys.TiledOrthogonalKeynet((2, 16, 16), net, 4)
        dev = torch.device('cuda:0')
        g = torch.Generator(device=dev).manual_seed(3)
        n = 8 * world_size
        x = torch.randn((n, 2, 16, 16), generator=g, device=dev)
        x[kdist.shard_bounds(n, world_size - 1, world_size)[0]:] *= 300.0      # the LAST rank's shard
        xc = sensor.fromtensor(x).encrypt().astensor()
        y = kdist.sharded_forward(knet, xc)                     # local forward (calibrates on this rank's shard) + sync_contract + all-gather
        state = {n_: c._exact for (n_, c) in knet._keynet.named_children() if isinstance(c, KeyedLayer)}
        (lo, hi) = kdist.shard_bounds(n, rank, world_size)
        mine = knet.forward_linear(xc[lo:hi])[:, :-1]           # this rank's shard again, under the agreed contracts
        own = bool(torch.equal(y[lo:hi], mine))
        # a peer's shard recomputed here (what bench.py's collective record does on rank 0): replicas are bit-identical
        (plo, phi) = kdist.shard_bounds(n, (rank + 1) % world_size, world_size)
        peer = bool(torch.equal(y[plo:phi], knet.forward_linear(xc[plo:phi])[:, :-1]))
        q.put((rank, state, own, peer, knet.contract_report()['switched']))
    finally:
        dist.destroy_process_group()


def run_contract(mode, world_size=2, timeout=300):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=contract_worker, args=(r, world_size, port, mode, q)) for r in range(world_size)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=timeout) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def run(kind, n, world_size=2, timeout=300, backend='gloo'):
    ctx = mp.get_context('spawn')        # fresh processes: a forked child must never inherit an initialised GPU runtime
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world_size, port, kind, n, q, backend)) for r in range(world_size)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=timeout) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res
