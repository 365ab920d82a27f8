"""N>1 path on CPU: world_size-2 gloo processes shard a batch and all-gather logits in rank-major order."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from keynet_amd import dist as kdist


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeKeynet(object):
    """Stands in for a KeyedModel on a GPU-less box: a fixed linear map with a homogeneous column (the sharding and the
    collective are what is under test here; the kernels are covered by the gpu tests)."""
    def __init__(self):
        g = torch.Generator().manual_seed(0)
        self.M = torch.randn(13, 5, generator=g)

    def forward_linear(self, x):
        return torch.cat((x[:, :-1] @ self.M, x[:, -1:]), dim=1)


def _worker(rank, world_size, port, n, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    try:
        g = torch.Generator().manual_seed(1)
        x = torch.cat((torch.randn(n, 13, generator=g), torch.ones(n, 1)), dim=1)
        knet = _FakeKeynet()
        y = kdist.sharded_forward(knet, x)
        ref = knet.forward_linear(x)[:, :-1]
        (lo, hi) = kdist.shard_bounds(n, rank, world_size)
        q.put((rank, bool(torch.equal(y, ref)), (lo, hi), tuple(y.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n', [8, 7])
def test_sharded_forward_equals_single_process(n):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert res[0][2][0] == 0 and res[0][2][1] == res[1][2][0] and res[1][2][1] == n
    assert all(r[3] == (n, 5) for r in res)


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 256, 2048, 2049):
        for ws in (1, 2, 4, 8):
            b = [kdist.shard_bounds(n, r, ws) for r in range(ws)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(ws - 1))
            sizes = [hi - lo for (lo, hi) in b]
            assert max(sizes) - min(sizes) <= 1
