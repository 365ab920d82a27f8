"""Multi-GPU keyed inference: independent batch shards, one all-gather of logits.

Every batch column of the keyed forward is independent (scipy's csr_matvecs never mixes columns), so the batch is split
evenly over ranks with NO collective inside the forward; keyed weights are replicated.  The only exchange is one
all_gather of the [B/world, classes] f32 logits (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests),
rank-major so the gathered result equals the single-GPU output row for row.
"""
import torch
import torch.distributed as dist


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def shard_bounds(n, rank, world_size):
    """Contiguous, balanced [lo, hi) of `n` batch rows for `rank` (first n % world ranks get one extra)."""
    (q, r) = divmod(n, world_size)
    lo = rank * q + min(rank, r)
    return (lo, lo + q + (1 if rank < r else 0))


def shard_batch(x, rank=None, world_size=None):
    (rk, ws) = world()
    rank = rk if rank is None else rank
    world_size = ws if world_size is None else world_size
    (lo, hi) = shard_bounds(x.shape[0], rank, world_size)
    return x[lo:hi]


def gather_logits(local_logits, total=None):
    """all_gather of per-rank logits [n_local, classes] -> [total, classes] on every rank, rank-major.

    Uneven shards (total % world != 0) are padded to the largest shard for the collective and trimmed afterwards."""
    (rank, ws) = world()
    if not (dist.is_available() and dist.is_initialized()):
        return local_logits
    # an initialised group of ONE rank still runs the collective: `bench.py --gpus 1 --dist` and tests/test_rccl_gpu.py drive RCCL's
    # communicator set-up and the device all_gather_into_tensor on a single-GPU box that way (a few microseconds)
    if local_logits.is_cuda and dist.get_backend() == 'gloo':
        # gloo has no device collectives: bounce through the host (single-GPU test rigs; the production backend is RCCL)
        return gather_logits(local_logits.cpu(), total=total).to(local_logits.device)
    n_local = local_logits.shape[0]
    if total is None:
        t = torch.tensor([n_local], dtype=torch.int64, device=local_logits.device)
        dist.all_reduce(t)
        total = int(t.item())
    sizes = [shard_bounds(total, r, ws)[1] - shard_bounds(total, r, ws)[0] for r in range(ws)]
    assert sizes[rank] == n_local, 'local shard size does not match the balanced split'
    m = max(sizes)
    buf = local_logits.contiguous()
    if n_local < m:
        buf = torch.cat((buf, buf.new_zeros((m - n_local,) + tuple(buf.shape[1:]))), dim=0)
    out = buf.new_empty((ws * m,) + tuple(buf.shape[1:]))
    dist.all_gather_into_tensor(out, buf)
    if all(s == m for s in sizes):
        return out
    return torch.cat([out[r * m:r * m + sizes[r]] for r in range(ws)], dim=0)


def sharded_forward(knet, x_cipher_full):
    """Forward of this rank's shard of an [N, D0+1] encrypted batch + all-gather: returns [N, classes] logits on every rank."""
    (rank, ws) = world()
    xs = shard_batch(x_cipher_full, rank, ws)
    return gather_logits(replicated_forward(knet, xs), total=x_cipher_full.shape[0])


def replicated_forward(knet, x_cipher_local):
    """This rank's forward of ITS batch with the replicas' arithmetic kept identical: [n_local, classes] logits.
    Collective when a process group is initialised and the key-net has layers decided by calibration (one all-reduce of one small integer
    per keyed layer, KeyedModel.sync_contract): a layer that ANY rank's batch moved to the more conservative contract runs there on every
    rank from this step on, and a rank whose decision changed recomputes its batch -- so every shard of the gathered result is what a
    single process would have computed for those images with the final contracts (bench.py re-computes a peer's shard on rank 0 and
    compares bit for bit).  Key-nets under declared contracts (exact=True / False everywhere) need and do no collective."""
    y = knet.forward_linear(x_cipher_local)
    if dist.is_available() and dist.is_initialized() and hasattr(knet, 'sync_contract') and any(r['calibration'] is not None or r['exact'] == 'auto' or r['declared'] == 'auto' for r in knet.contract_report()['layers']):
        # exactly ONE collective per call on every rank (the condition above only looks at state that is replicated: which layers are under
        # the calibrated contract); the recompute below is local
        if knet.sync_contract():
            y = knet.forward_linear(x_cipher_local)
    return y[:, :-1]
