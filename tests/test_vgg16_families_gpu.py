"""The reference's two remaining VGG-16 key configurations (test/test_keynet.py:116-129 test_vgg16_stochastic -- hierarchical block permutation
at levels 0, 1, 2 + block-local doubly-stochastic keys, alpha = 2, tile = blocksize = H / 16, asserted at atol 1e-5 by the reference; and
:155-173 test_vgg16_orthogonal_8 -- block Givens rotations, tile = blocksize = H / 8) on the VGG-16 topology at reduced width and size
(21 keyed layers, 3 x 32 x 32 images: the full-size key-nets take 18 minutes of host keying each and run as `bench.py --workload
vgg16-stochastic | vgg16-givens28`, profiles/r04_*).  Per configuration: keyed == plain at the reference's tolerance, the order-preserving
kernels bit-equal to the CPU oracle on sampled rows of two conv layers, the float-key contract per layer, slots per output pixel and
which loader ran (kn_spmm_plan)."""
import warnings

import numpy as np
import pytest
import torch

import oracle
from keynet_amd import system as ksys
from keynet_amd import sparse as ksp
from keynet_amd.layer import KeyedLayer, FLOAT_KEY_TOL, gate
from keynet_amd.models import VGG16

pytestmark = pytest.mark.gpu

CONFIGS = {
    # name: (Keynet kwargs at 32 x 32, the reference's own atol for keyed == plain)
    'stochastic': (dict(tileshape=(2, 2), global_geometric='hierarchical_permutation', hierarchical_blockshape=(2, 2), hierarchical_permute_at_level=(0, 1, 2),
                        local_geometric='doubly_stochastic', alpha=2.0, blocksize=2, local_photometric='uniform_random_affine', beta=1.0, gamma=1.0, memoryorder='channel'), 1e-5),
    'givens_tile_h8': (dict(tileshape=(4, 4), global_geometric='identity', hierarchical_blockshape=(2, 2), hierarchical_permute_at_level=(0, 1, 2),
                            local_geometric='givens_orthogonal', alpha=2.0, blocksize=4, local_photometric='uniform_random_affine', beta=1.0, gamma=1.0, memoryorder='channel'), 1e-3),
}


@pytest.fixture(scope='module', params=sorted(CONFIGS))
def keyed(request):
    assert torch.cuda.is_available()
    (kw, atol) = CONFIGS[request.param]
    torch.manual_seed(0)
    net = VGG16(num_classes=10, width=8, fc_width=64, insize=32).eval()
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        (sensor, knet) = ksys.Keynet((3, 32, 32), net, **kw)
    return (request.param, atol, net, sensor, knet)


def test_structure_and_default_contract(keyed):
    (name, atol, net, sensor, knet) = keyed
    layers = {n: c for (n, c) in knet._keynet.named_children() if isinstance(c, KeyedLayer)}
    assert len(layers) == 21
    convs = {n: c.W for (n, c) in layers.items() if isinstance(c.W, ksp.Conv2dTiledMatrix)}
    assert len(convs) == 13
    assert all(c._exact == 'auto' for c in layers.values())              # float keys: the contract is decided per layer at the first forward
    fill = {}
    for (n, W) in convs.items():
        t = W._taps if W._taps is not None else None
        if t is not None:
            assert t['ent_coef'] is not None, n                           # every entry carries a coefficient
            fill[n] = int(np.bincount(t['ent_out']).max())
    print(name, 'max slots per output pixel:', fill)
    if name == 'stochastic' and fill:
        assert max(fill.values()) > 9                                     # the inverse of a doubly-stochastic block is dense: the operators fill in


def test_keyed_equals_plain_at_the_references_tolerance(keyed):
    (name, atol, net, sensor, knet) = keyed
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    x = torch.randn(256, 3, 32, 32, generator=g)
    xc = sensor.fromtensor(x.to(dev)).encrypt().astensor()
    y = knet.forward_linear(xc)                                           # calibrates the contract
    rep = knet.contract_report()
    assert not rep['undecided']
    with torch.no_grad():
        yp = net(x).reshape(256, -1).numpy()
    err = float(np.abs(y[:, :-1].cpu().numpy() - yp).max())
    print('%s: keyed vs plain %.3g (reference atol %g); switched to exact: %s; rescreen %s' % (name, err, atol, rep['switched'], rep['rescreen']))
    assert err <= atol, err
    # single image through the reference's own entry point (N = 1 shape)
    y1 = knet.forward(sensor.fromtensor(x[:1].to(dev)).encrypt().astensor())
    assert tuple(y1.shape) == (10, 1, 1) and np.allclose(y1.flatten().cpu().numpy(), yp[0], atol=atol)
    # every conv layer, as shipped, within the float-key tolerance of the order-preserving path on the same input
    children = list(knet._keynet.named_children())
    yin = xc
    for (i, (lname, c)) in enumerate(children):
        if not isinstance(c, KeyedLayer):
            continue
        fuse = (i + 1 < len(children)) and isinstance(children[i + 1][1], torch.nn.ReLU)
        out = c.forward(yin, fuse_relu=fuse)
        if isinstance(c.W, ksp.Conv2dTiledMatrix):
            ye = c.W.torchdot(yin.t(), relu=fuse, exact=True).t()
            (d, m) = (float((ye - out).abs().max()), float(ye.abs().max()))
            assert gate(out, ye)[0] <= 1.0, (lname, d, m, c._exact)          # element-wise np.allclose(atol=1e-5), the reference's own form
            with torch.cuda.device(dev):
                plan = c.W._device_op(dev).plan(256, (1 if fuse else 0) | (2 if c._exact is True else 0))
            print(lname, 'exact' if c._exact is True else 'mfma', 'diff %.3g of %.3g' % (d, m), '|', plan)
        yin = out
    # decrypt round trip of the image key (float keys: to rounding)
    back = sensor.fromtensor(x[:2].to(dev)).encrypt().decrypt().astensor().cpu().numpy()
    assert np.allclose(back, x[:2].numpy(), rtol=1e-4, atol=1e-4)


def test_exact_mode_bit_equal_to_oracle(keyed):
    """Order-preserving kernels on the real operators (coefficient entries, dense fill-in, several taps on one (output, input) pixel pair)
    == the CPU oracle on the expansion of sampled output rows, bit for bit, chained layer to layer on 8 images."""
    (name, atol, net, sensor, knet) = keyed
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(6)
    x = torch.randn(8, 3, 32, 32, generator=g)
    xin = sensor.fromtensor(x.to(dev)).encrypt().astensor().t().contiguous()
    rng = np.random.RandomState(2)
    checked = []
    for (lname, c) in knet._keynet.named_children():
        if not isinstance(c, KeyedLayer):
            continue
        relu = lname.startswith(('conv', 'fc6', 'fc7'))
        W = c.W
        ye = W.torchdot(xin, relu=relu, exact=True)
        if isinstance(W, ksp.Conv2dTiledMatrix) and lname in ('conv1_2', 'conv3_1'):
            (Cout, Hout, Wout) = W._outshape
            if W._taps is not None:
                ns = np.bincount(W._taps['ent_out'], minlength=Hout * Wout)
                pix = np.unique(np.concatenate((rng.choice(Hout * Wout, size=2, replace=False), [int(np.argmax(ns))])))
                M = W.rows_csr(pix)
                rows = (np.arange(Cout)[:, None] * Hout * Wout + pix[None, :]).ravel()
            else:
                full = W.tocsr()
                rows = np.sort(rng.choice(full.shape[0] - 1, size=40, replace=False))
                M = full[rows]
            ref = oracle.csr_matvecs(M.shape, M.indptr, M.indices, M.data.astype(np.float32), xin.cpu().numpy())
            if relu:
                ref = np.maximum(ref, 0)
            assert np.array_equal(ye.cpu().numpy()[rows], ref), lname
            checked.append(lname)
        xin = ye
        if lname == 'conv3_1':
            break
    assert checked == ['conv1_2', 'conv3_1']


# ---- the same doubly-stochastic configuration keyed DIRECTLY (factored operators: what the full-size key-net is made of) --------------------------------
@pytest.fixture(scope='module')
def stochastic_direct():
    """3 x 64 x 64, reduced width, direct keying: every conv operator is taps x entries with float coefficients, up to ~300 slots per output pixel and
    several taps on one (output, input) pixel pair -- the structure of `bench.py --workload vgg16-stochastic` (500 - 5 400 slots) in a few seconds of keying."""
    assert torch.cuda.is_available()
    (kw, atol) = CONFIGS['stochastic']
    kw = dict(kw, tileshape=(4, 4), blocksize=4)
    torch.manual_seed(0)
    net = VGG16(num_classes=10, width=8, fc_width=64, insize=64).eval()
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        (sensor, knet) = ksys.Keynet((3, 64, 64), net, direct=True, **kw)
    return (atol, net, sensor, knet)


def test_filled_in_operators_take_the_filled_in_kernels(stochastic_direct):
    """kn_spmm_plan on the directly keyed doubly-stochastic VGG-16: in the reference's order every conv layer with several taps per pixel pair or
    more than 64 slots per pixel runs convtaps_exact_fill_kernel (bit-equal to the generic kernel it replaces: a fresh handle under KN_NO_FILL_EXACT=1); on the
    matrix cores a layer with more than 64 slots per pixel walks its slots group by group on the wave-uniform-pointer loaders."""
    import copy
    import os
    from keynet_amd import _capi
    (atol, net, sensor, knet) = stochastic_direct
    dev = torch.device('cuda:0')
    convs = [(n, c.W) for (n, c) in knet._keynet.named_children() if isinstance(c, KeyedLayer) and isinstance(c.W, ksp.Conv2dTiledMatrix)]
    assert len(convs) == 13
    (n_fill, n_groups) = (0, 0)
    for (n, W) in convs:
        t = W._taps
        assert t is not None and t['ent_coef'] is not None, n
        HiWi = W._inshape[1] * W._inshape[2]
        dups = len(np.unique(t['ent_out'].astype(np.int64) * HiWi + t['ent_in'])) < len(t['ent_out'])
        slots = int(np.bincount(t['ent_out']).max())
        with torch.cuda.device(dev):
            pe = W._device_op(dev).plan(256, _capi.KN_FLAG_EXACT)
            pm = W._device_op(dev).plan(256, 0)
        if dups or slots > 64:
            assert 'convtaps_exact_fill_kernel<taps in registers' in pe, (n, pe)      # (at 256 columns: the forms with two column tiles per wavefront where the layer has the work for them)
            n_fill += 1
        if slots > 64 and W._inshape[0] % 16 == 0:                        # (the wave-uniform-pointer loaders take whole 16-channel chunks: conv1_1 / conv1_2 / conv2_1 of this reduced net keep the generic loader)
            assert 'slot groups' in pm, (n, pm)
            n_groups += 1
    assert n_fill >= 8 and n_groups >= 3, (n_fill, n_groups)
    # one real filled-in layer, whole: the filled-in kernel against the generic one
    (n, W) = next((n, W) for (n, W) in convs if n == 'conv2_1')
    x = torch.randn(W.shape[1], 64, generator=torch.Generator().manual_seed(1)).to(dev)
    x[-1] = 1.0
    y = W.torchdot(x, relu=True, exact=True)
    os.environ['KN_NO_FILL_EXACT'] = '1'
    try:
        Wg = copy.deepcopy(W)
        Wg._op = None
        yg = Wg.torchdot(x, relu=True, exact=True)
        with torch.cuda.device(dev):
            assert 'convtaps_exact_kernel' in Wg._device_op(dev).plan(64, _capi.KN_FLAG_EXACT)
    finally:
        del os.environ['KN_NO_FILL_EXACT']
    assert torch.equal(y, yg)
    # ... and against the CPU oracle on the canonical CSR of sampled output pixels (host expansion: a pair's terms summed in entry order), on three real layers
    rng = np.random.RandomState(4)
    for (n, W) in convs:
        if n not in ('conv1_1', 'conv2_1', 'conv4_1'):
            continue
        (Cout, Hout, Wout) = W._outshape
        ns = np.bincount(W._taps['ent_out'], minlength=Hout * Wout)
        pix = np.unique(np.concatenate((rng.choice(Hout * Wout, size=2, replace=False), [int(np.argmax(ns))])))
        M = W.rows_csr(pix)
        xs = torch.randn(W.shape[1], 8, generator=torch.Generator().manual_seed(2))
        xs[-1] = 1.0
        ref = oracle.csr_matvecs(M.shape, M.indptr, M.indices, M.data.astype(np.float32), xs.numpy())
        got = W.torchdot(xs.to(dev), exact=True).cpu().numpy()
        rows = (np.arange(Cout)[:, None] * Hout * Wout + pix[None, :]).ravel()
        HiWi = W._inshape[1] * W._inshape[2]
        (_, cnt) = np.unique(W._taps['ent_out'].astype(np.int64) * HiWi + W._taps['ent_in'], return_counts=True)
        assert np.array_equal(got[rows], ref), n
        print(n, 'bit-equal to the oracle on pixels', pix.tolist(), '| terms per stored entry: max', int(cnt.max()), 'mean %.2f' % float(cnt.mean()))


def test_directly_keyed_stochastic_net_meets_the_references_tolerance_fused_and_split(stochastic_direct):
    """Keyed == plain at the reference's own atol (test/test_keynet.py:116-129: 1e-5) under the float-key contract, twice: as calibration decides by itself (the
    cost rule offers the split application only where its estimate is under half the fused launch -- not at this size) and with the split application offered to
    every filled-in layer (what the full-size key-net's eight tolerance layers run): at least one layer then decides 'split', every conv layer as shipped is inside
    the element-wise gate against the order-preserving kernel, and the logits agree with the fused run to 1e-5."""
    (atol, net, sensor, knet) = stochastic_direct
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(7)
    x = torch.randn(256, 3, 64, 64, generator=g)
    xc = sensor.fromtensor(x.to(dev)).encrypt().astensor()
    with torch.no_grad():
        yp = net(x).reshape(256, -1).numpy()
    knet.exact_mode(None)
    y_fused = knet.forward_linear(xc)[:, :-1].cpu().numpy()
    rep = knet.contract_report()
    assert not rep['undecided'] and float(np.abs(y_fused - yp).max()) <= atol
    decided_fused = {r['name']: r['exact'] for r in rep['layers']}
    offered = ksp.Conv2dTiledMatrix.split_capable
    ksp.Conv2dTiledMatrix.split_capable = lambda self, n_vecs=None: self._taps is not None and self.fill_factor() >= self.SPLIT_MIN_FILL
    try:
        knet.exact_mode(None)
        y_split = knet.forward_linear(xc)[:, :-1].cpu().numpy()
        rep2 = knet.contract_report()
        split = [r['name'] for r in rep2['layers'] if r['exact'] == 'split']
        print('fused decisions:', decided_fused, '| decided split when offered:', split)
        assert len(split) >= 1 and float(np.abs(y_split - yp).max()) <= atol and float(np.abs(y_split - y_fused).max()) <= 1e-5
        children = list(knet._keynet.named_children())
        yin = xc
        for (i, (lname, c)) in enumerate(children):
            if not isinstance(c, KeyedLayer):
                continue
            fuse = (i + 1 < len(children)) and isinstance(children[i + 1][1], torch.nn.ReLU)
            out = c.forward(yin, fuse_relu=fuse)
            if lname in split:
                ye = c.W.torchdot(yin.t(), relu=fuse, exact=True).t()
                assert gate(out, ye)[0] <= 1.0, lname
            yin = out
        assert torch.equal(knet.forward_linear(xc)[:, :-1].cpu(), torch.as_tensor(y_split))      # decided: the second forward repeats the first
    finally:
        ksp.Conv2dTiledMatrix.split_capable = offered
        knet.exact_mode(None)


def test_one_full_size_filled_in_layer_against_the_oracle():
    """The driver's tier sees the doubly-stochastic family on reduced nets; the full-size key-net (minutes of keying) is builder-run (profiles/r06_vgg16_stochastic_*).  This is ONE
    layer of it at FULL size, keyed the way `Keynet(...)` keys conv5_1 of VGG-16 under test/test_keynet.py:116-129 -- input key: hierarchical permutation + doubly-stochastic
    14 x 14 blocks + affine photometric (its inverse is dense inside the block: the fill-in), output key: the following ReLU's permutation + gain -- 512 -> 512 channels on 14 x 14
    pixels: ~780 slots per output pixel, 4.4 terms per stored entry, 9.1e9 stored values (no CSR of it can exist).  The order-preserving kernel AS THE BENCH RUNS IT (256 columns:
    two column tiles per wavefront; 64 columns: one) against the CPU oracle on the canonical rows of one output pixel (all 512 channels, 49.8 M stored entries), bit for bit."""
    import time
    from torch import nn
    from keynet_amd import _capi
    dev = torch.device('cuda:0')
    common = dict(memoryorder='channel', blocksize=14, tileshape=(14, 14), alpha=2.0, beta=1.0, gamma=1.0, hierarchical_blockshape=(2, 2), hierarchical_permute_at_level=(0, 1, 2))
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        (_, a_in_inv) = ksys.keygen((512, 14, 14), global_photometric='identity', local_photometric='uniform_random_affine', global_geometric='hierarchical_permutation',
                                    local_geometric='doubly_stochastic', **common)
        (a_out, _) = ksys.keygen((512, 14, 14), global_photometric='identity', local_photometric='uniform_random_gain', global_geometric='identity', local_geometric='permutation', **common)
    torch.manual_seed(0)
    layer = KeyedLayer(nn.Conv2d(512, 512, 3, padding=1), (512, 14, 14), (512, 14, 14), a_out, a_in_inv, tileshape=(14, 14), direct=True)
    W = layer.W
    t = W._taps
    assert isinstance(W, ksp.Conv2dTiledMatrix) and t is not None and t['ent_coef'] is not None
    pairs = len(np.unique(t['ent_out'].astype(np.int64) * 196 + t['ent_in']))
    assert len(t['ent_out']) / 196.0 > 500 and len(t['ent_out']) / pairs > 3 and pairs * 512 * 512 > 8e9
    with torch.cuda.device(dev):
        (p256, p64) = (W._device_op(dev).plan(256, _capi.KN_FLAG_EXACT), W._device_op(dev).plan(64, _capi.KN_FLAG_EXACT))
    assert 'convtaps_exact_fill_kernel' in p256 and 'two column tiles per wavefront' in p256 and 'convtaps_exact_fill_kernel' in p64 and 'two column tiles' not in p64, (p256, p64)
    x = torch.randn(W.shape[1], 256, generator=torch.Generator().manual_seed(3))
    x[-1] = 1.0
    xd = x.to(dev)
    y256 = W.torchdot(xd, relu=True, exact=True)
    y64 = W.torchdot(xd[:, :64].contiguous(), relu=True, exact=True)
    assert torch.equal(y256[:, :64], y64)                                   # the two forms agree bit for bit on the columns both computed
    pix = np.array([int(np.argmax(np.bincount(t['ent_out'], minlength=196)))])                          # the pixel with the most slots
    t0 = time.time()
    M = W.rows_csr(pix)
    cols = [0, 1, 63, 64, 129, 190, 254, 255]                               # columns of both tiles of both 128-column items
    ref = np.maximum(oracle.csr_matvecs(M.shape, M.indptr, M.indices, M.data.astype(np.float32), np.ascontiguousarray(x.numpy()[:, cols])), 0)
    rows = np.arange(512) * 196 + int(pix[0])
    got = y256[torch.as_tensor(rows, device=dev)][:, torch.as_tensor(cols, device=dev)].cpu().numpy()
    assert np.array_equal(got, ref), np.abs(got - ref).max()
    print('full-size conv5_1-like layer: %d slots per pixel at pixel %d, %d stored entries in its 512 rows, bit-equal; host expansion + oracle %.1f s' % (int(np.bincount(t['ent_out']).max()), int(pix[0]), M.nnz, time.time() - t0))
