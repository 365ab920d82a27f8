"""The bench workloads (host only: no GPU call in this module): BASELINE.json's configs and the reference's float-key VGG-16 tests, keyed under fixed seeds."""
import time

import numpy as np
import torch

from keynet_amd import system as ksys
from keynet_amd.models import VGG16, LeNet_AvgPool, AllConvNet
from .common import log


def build_workload(name, rank, exact=None):
    """(sensor, knet, inshape, per_gpu_batch, description, source network).  Deterministic under the seeds, identical on every rank."""
    t0 = time.time()
    if name == 'vgg16':
        torch.manual_seed(0)
        net = VGG16(num_classes=2622).eval()
        np.random.seed(0)
        (sensor, knet) = ksys.TiledPermutationKeynet((3, 224, 224), net, 64, exact=exact)
        (inshape, batch, desc) = ((3, 224, 224), 256, 'TiledPermutationKeynet VGG16(2622) 3x224x224 tile=64 (effective 56/28/14/7)')
    elif name == 'vgg16-gain':
        # the float-key variant of the same config that is constructible at full size: block permutation + block-local photometric gain
        # (every keyed entry carries the coefficient a_out[o] / a_in[i]; 1e-5 contract); orthogonal tile keys fill every tile in
        torch.manual_seed(0)
        net = VGG16(num_classes=2622).eval()
        np.random.seed(0)
        (sensor, knet) = ksys.Keynet((3, 224, 224), net, local_geometric='permutation', local_photometric='uniform_random_gain', beta=0.5,
                                     tileshape=(64, 64), blocksize=64, exact=exact)
        (inshape, batch, desc) = ((3, 224, 224), 256, 'Keynet(permutation + uniform_random_gain, tile=64) VGG16(2622) 3x224x224: float keys')
    elif name == 'vgg16-givens':
        # the reference's OWN float-key VGG-16 configuration (test/test_keynet.py:133-151, test_vgg16_orthogonal): block-local Givens rotations
        # (alpha = 2) + block-local affine photometric keys (beta = gamma = 1), tile = blocksize = 224 // 16 = 14, channel memory order.
        # Keyed directly in factored form (the reference route cannot build it: 15 G non-zeros); fill-in: ~9.0-9.3 slots per output pixel
        # on average, up to 19, every entry carries a coefficient.
        torch.manual_seed(0)
        net = VGG16(num_classes=2622).eval()
        np.random.seed(0)
        (sensor, knet) = ksys.Keynet((3, 224, 224), net, tileshape=(224 // 16, 224 // 16), global_geometric='identity', hierarchical_blockshape=(2, 2),
                                     hierarchical_permute_at_level=(0, 1, 2), local_geometric='givens_orthogonal', alpha=2.0, blocksize=224 // 16,
                                     local_photometric='uniform_random_affine', beta=1.0, gamma=1.0, memoryorder='channel', exact=exact)
        (inshape, batch, desc) = ((3, 224, 224), 256, 'Keynet(givens_orthogonal alpha=2 + uniform_random_affine beta=gamma=1, tile=blocksize=14) VGG16(2622) 3x224x224: '
                                                       'the float-key configuration of test/test_keynet.py:133-151')
    elif name == 'vgg16-givens28':
        # test/test_keynet.py:155-173 (test_vgg16_orthogonal_8): the same float-key family with tile = blocksize = 224 // 8 = 28
        torch.manual_seed(0)
        net = VGG16(num_classes=2622).eval()
        np.random.seed(0)
        (sensor, knet) = ksys.Keynet((3, 224, 224), net, tileshape=(224 // 8, 224 // 8), global_geometric='identity', hierarchical_blockshape=(2, 2),
                                     hierarchical_permute_at_level=(0, 1, 2), local_geometric='givens_orthogonal', alpha=2.0, blocksize=224 // 8,
                                     local_photometric='uniform_random_affine', beta=1.0, gamma=1.0, memoryorder='channel', exact=exact)
        (inshape, batch, desc) = ((3, 224, 224), 256, 'Keynet(givens_orthogonal alpha=2 + uniform_random_affine beta=gamma=1, tile=blocksize=28) VGG16(2622) 3x224x224: '
                                                       'the float-key configuration of test/test_keynet.py:155-173')
    elif name == 'vgg16-stochastic':
        # test/test_keynet.py:116-129 (test_vgg16_stochastic; the reference asserts 1e-5 there): hierarchical block permutation at levels 0, 1, 2 +
        # block-local doubly-stochastic keys (alpha = 2) + affine photometric keys, tile = blocksize = 14.  The INVERSE of a doubly-stochastic block
        # is dense, so every 14 x 14 block of a keyed operator fills in: ~490-560 (first layer of a stage: 1 700-5 400) slots per output pixel instead
        # of 9 -- 60x the multiply-adds of the permutation key-net (0.9 T per image), which is why this workload runs 16 images per step.
        torch.manual_seed(0)
        net = VGG16(num_classes=2622).eval()
        np.random.seed(0)
        (sensor, knet) = ksys.Keynet((3, 224, 224), net, tileshape=(224 // 16, 224 // 16), global_geometric='hierarchical_permutation', hierarchical_blockshape=(2, 2),
                                     hierarchical_permute_at_level=(0, 1, 2), local_geometric='doubly_stochastic', alpha=2.0, blocksize=224 // 16,
                                     local_photometric='uniform_random_affine', beta=1.0, gamma=1.0, memoryorder='channel', exact=exact)
        (inshape, batch, desc) = ((3, 224, 224), 16, 'Keynet(hierarchical_permutation levels 0-2 + doubly_stochastic alpha=2 + uniform_random_affine, tile=blocksize=14) VGG16(2622) '
                                                      '3x224x224: test/test_keynet.py:116-129')
    elif name == 'vgg16-slice':
        # cfg5's topology at a size that keys in seconds (the 21 keyed layers of VGG-16 at width 8 on 32 x 32 inputs, tile 8): what the 8-rank rehearsal test of the
        # tiers runs (tests/test_dist_gpu.py); never a reported number
        torch.manual_seed(0)
        net = VGG16(num_classes=10, width=8, fc_width=64, insize=32).eval()
        np.random.seed(0)
        (sensor, knet) = ksys.TiledPermutationKeynet((3, 32, 32), net, 8, exact=exact)
        (inshape, batch, desc) = ((3, 32, 32), 16, 'TiledPermutationKeynet VGG16 slice (width 8, 3x32x32, tile 8): rehearsal workload, not a BASELINE config')
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
