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
