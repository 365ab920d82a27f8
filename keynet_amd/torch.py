"""Tensor-side helpers of the keyed forward (mirror of the parts of keynet/torch.py that sit on the path).

On a CUDA(ROCm) tensor the homogeneous augmentation runs in HIP kernels (kn_affine_to_linear / kn_linear_to_affine);
CPU tensors take the trivial torch route (this is plumbing before/after the path, not the path).
"""
from collections import OrderedDict
import numpy as np
import torch
from torch import nn

from . import _capi


def _stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def affine_to_linear(x):
    """NxCxHxW (or CxHxW) -> Nx(C*H*W+1) with a trailing ones column (keynet/torch.py:65-68).

    For a device tensor the result is a transposed VIEW of a feature-major [D+1, N] buffer, i.e. already in the layout
    the keyed layers consume (their x.t() is then free)."""
    if x.dim() == 3:
        x = x.unsqueeze(0)
    (N, D) = (x.shape[0], int(np.prod(x.shape[1:])))
    if x.is_cuda:
        xc = x.reshape(N, D).contiguous().float()
        out = torch.empty((D + 1, N), dtype=torch.float32, device=x.device)
        _capi.affine_to_linear(xc.data_ptr(), N, D, out.data_ptr(), N, _stream_ptr())
        return out.t()
    return torch.cat((x.reshape(N, D), torch.ones(N, 1, dtype=x.dtype)), dim=1)


def linear_to_affine(x, outshape=None):
    """Nx(K+1) -> NxK, checking that the homogeneous column is 1 within 1e-3 (ValueError otherwise), then reshaping to
    `outshape` (keynet/torch.py:71-77)."""
    assert x.dim() == 2
    (N, K) = (x.shape[0], x.shape[1] - 1)
    if x.is_cuda and x.t().is_contiguous() and x.dtype == torch.float32:
        xt = x.t()
        out = torch.empty((N, K), dtype=torch.float32, device=x.device)
        dev = torch.zeros(1, dtype=torch.float32, device=x.device)
        _capi.linear_to_affine(xt.data_ptr(), N, N, K, out.data_ptr(), dev.data_ptr(), _stream_ptr())
        d = float(dev.item())
        if not (d <= 1e-3):
            raise ValueError('invalid affine vector: homogeneous coordinate deviates from 1 by %g' % d)
        return out.reshape(outshape) if outshape is not None else out
    last = x[:, -1].detach().cpu().numpy()
    if not np.allclose(last, 1, atol=1e-3):
        raise ValueError('invalid affine vector: homogeneous coordinate deviates from 1 by %g' % float(np.max(np.abs(last - 1))))
    xa = torch.narrow(x, 1, 0, K)
    return xa.reshape(outshape) if outshape is not None else xa


def affine_to_linear_matrix(W_affine, bias=None):
    """(Wx+b)^T as one left-multiplied matrix [[W^T, 0], [b, 1]] of shape (in+1, out+1) (keynet/torch.py:80-89)."""
    Wt = W_affine.t()
    (R, C) = Wt.shape
    M = torch.zeros(R + 1, C + 1, dtype=Wt.dtype)
    M[:R, :C] = Wt
    if bias is not None:
        M[R, :C] = bias.reshape(C)
    M[R, C] = 1
    return M


def fuse_conv2d_and_bn(conv2d_weight, conv2d_bias, bn_running_mean, bn_running_var, bn_eps, bn_weight, bn_bias):
    """Fold an eval-mode BatchNorm2d into the preceding conv (keynet/torch.py:99-113).  The association of the f32
    operations is the reference's -- weights scaled by (bn_weight / std), bias as ((b - mean) / std) * bn_weight + bn_bias --
    because the folded parameters feed the stored keyed operators, which must match bit for bit under the same seed."""
    std = torch.sqrt(bn_running_var + np.float32(bn_eps))
    b = conv2d_bias if conv2d_bias is not None else bn_running_mean.new_zeros(bn_running_mean.shape)
    w = conv2d_weight * (bn_weight / std).reshape(-1, 1, 1, 1)
    return (w, ((b - bn_running_mean) / std) * bn_weight + bn_bias)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def netshape(net, inshape):
    """Trace one dummy forward and return an ordered {name: {inshape, outshape, prevlayer, nextlayer}} chain with the
    pseudo entries 'input' and 'output' (keynet/torch.py:21-62).  Leaf modules are visited in execution order;
    nn.Sequential containers are descended into.  Shapes are canonicalised to (C,H,W) ((C,1,1) for vectors)."""
    chain = OrderedDict()
    hooks = []

    def canon(t):
        return (t.shape[1], t.shape[2], t.shape[3]) if t.dim() == 4 else (t.shape[1], 1, 1)

    def attach(container):
        for (name, layer) in container._modules.items():
            if isinstance(layer, nn.Sequential):
                attach(layer)
            else:
                hooks.append(layer.register_forward_hook(lambda m, i, o, _n=name: record(_n, i, o)))

    def record(name, inp, out):
        (ishape, oshape) = (canon(inp[0]), canon(out))
        if 'input' not in chain:
            chain['input'] = {'prevlayer': None, 'nextlayer': name, 'inshape': ishape, 'outshape': oshape}
        chain.pop('output', None)
        prev = next(reversed(chain))
        chain[name] = {'inshape': ishape, 'outshape': oshape, 'prevlayer': prev, 'nextlayer': None}
        chain[prev]['nextlayer'] = name
        chain['output'] = {'nextlayer': None, 'prevlayer': name, 'inshape': ishape, 'outshape': oshape}

    net.eval()
    attach(net)
    try:
        with torch.no_grad():
            net.forward(torch.rand(1, inshape[0], inshape[1], inshape[2]))
    finally:
        for h in hooks:
            h.remove()
    return chain


class TiledMatrix(object):
    """The explicit tile loop of keynet/torch.py:165-184 (`keynet.torch.TiledMatrix._torchdot`; dead code in the reference, kept for API coverage):

        for (i, j, k) in blocks:  for (ii, jj, v) in tiles[k]:  y[i + ii, :] += v * x[j + jj, :]

    Here the loop is the ORDER of an order-preserving CSR: every output row accumulates its terms in exactly the sequence the loop visits them
    (blocks in the given order, a tile's entries in their stored order), f32 multiply then f32 add, on the device (kn_csr_create + kn_spmm).  The
    reference compiles its loop with numba fastmath + parallel, so its own rounding is unspecified; the serial loop is what tests compare with."""

    @staticmethod
    def _loop_csr(tileshape, shape, tiles, blocks):
        """(indptr, indices, data) whose rows list the loop's terms in visiting order.  `tiles[k]`: array [nnz_k, 3] of (ii, jj, v)."""
        (H, W) = (int(shape[0]), int(shape[1]))
        (rows, cols, vals) = ([], [], [])
        for (i, j, k) in blocks:
            b = np.asarray(tiles[int(k)], dtype=np.float64).reshape(-1, 3)
            rows.append(int(i) + b[:, 0].astype(np.int64))
            cols.append(int(j) + b[:, 1].astype(np.int64))
            vals.append(b[:, 2].astype(np.float32))
        cat = (lambda L, dt: np.concatenate(L).astype(dt) if len(L) else np.zeros(0, dt))
        (r, c, v) = (cat(rows, np.int64), cat(cols, np.int64), cat(vals, np.float32))
        assert r.size == 0 or (r.min() >= 0 and r.max() < H and c.min() >= 0 and c.max() < W), 'tile entry outside the operator'
        order = np.argsort(r, kind='stable')                              # stable: a row keeps the loop's sequence
        indptr = np.zeros(H + 1, np.int64)
        np.cumsum(np.bincount(r, minlength=H), out=indptr[1:])
        return (indptr.astype(np.int32), c[order].astype(np.int32), v[order])

    @staticmethod
    def _torchdot(x, tileshape, shape, tiles, blocks):
        """x: [W, N] float32 torch tensor on a ROCm device (or numpy, moved to the current device) -> y [H, N] on the same device."""
        was_numpy = isinstance(x, np.ndarray)
        if was_numpy:
            x = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).to(torch.device('cuda', torch.cuda.current_device()))
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] == int(shape[1]), 'non-conformal operand'
        (ip, ix, dt) = TiledMatrix._loop_csr(tileshape, shape, tiles, blocks)
        with torch.cuda.device(x.device):
            op = _capi.Operator.csr((int(shape[0]), int(shape[1])), ip, ix, dt)
            xc = x.contiguous()
            y = torch.empty((int(shape[0]), xc.shape[1]), dtype=torch.float32, device=x.device)
            op.spmm(xc.data_ptr(), xc.shape[1], xc.shape[1], y.data_ptr(), xc.shape[1], _capi.KN_FLAG_EXACT, _stream_ptr())
            torch.cuda.current_stream().synchronize()                     # the operator is destroyed on return
        return y.cpu().numpy() if was_numpy else y
