"""Random small source networks in the reference's naming convention (convN / reluN / poolN, fcN behind a flatten): shared by the GPU and the host fuzzers."""
from torch import nn


def random_net(rng, sides=(6, 8, 12, 16, 20)):
    """A random small source network in the reference's naming convention (convN / reluN / poolN, fcN behind a flatten), its input shape."""
    from keynet_amd.models import _Chain
    (C, side) = (int(rng.choice([1, 2, 3])), int(rng.choice(list(sides))))
    spec = []
    (c, hw, k) = (C, side, 0)
    for _ in range(rng.randint(0, 4)):
        k += 1
        (ks, stride, co) = (int(rng.choice([1, 3, 3, 5])), int(rng.choice([1, 1, 2])), int(rng.randint(2, 9)))
        if hw < 3:
            break
        if hw % 2:
            stride = 1                                              # (strided layers on even sides only: what the reference's Toeplitz builder supports, keynet/sparse.py:900-960)
        spec.append(('conv%d' % k, nn.Conv2d(c, co, ks, stride=stride, padding=ks // 2)))
        (c, hw) = (co, (hw + 2 * (ks // 2) - ks) // stride + 1)
        if rng.rand() < 0.8:
            spec.append(('relu%d' % k, nn.ReLU()))
        if rng.rand() < 0.5 and hw >= 4 and hw % 2 == 0:            # (odd pooling windows only, as the reference's Toeplitz builder: keynet/sparse.py:920)
            spec.append(('pool%d' % k, nn.AvgPool2d(3, stride=2, padding=1)))
            hw = (hw + 2 - 3) // 2 + 1
    feat = c * hw * hw
    n_fc = int(rng.randint(1, 4))
    for j in range(n_fc):
        out = int(rng.randint(2, 150)) if j + 1 < n_fc else int(rng.randint(2, 20))
        spec.append(('fc%d' % (j + 1), nn.Linear(feat, out)))
        feat = out
        if j + 1 < n_fc:
            k += 1
            spec.append(('relu%d' % (k + 10), nn.ReLU()))

    class Net(_Chain):
        flatten_before = 'fc1'

        def __init__(self):
            super(Net, self).__init__()
            for (name, m) in spec:
                setattr(self, name, m)
    return (Net().eval(), (C, side, side), [name for (name, _) in spec])
