"""Source-network ARCHITECTURES used as test / bench inputs (random-initialised; no training code).

Declared from the architecture tables of the reference's model files -- LeNet_AvgPool (keynet/mnist.py:49-63),
AllConvNet (keynet/cifar10.py:14-65), VGG16 (keynet/vgg.py:42-122) -- with the layer-naming convention the keying
relies on ('reluN' after a linear layer, 'xyz_bn' after 'xyz', 'dropoutN' skipped).
"""
from collections import OrderedDict
import torch
from torch import nn


class _Chain(nn.Module):
    """Named children applied in order; `flatten_before` names the first Linear (input is flattened there)."""
    flatten_before = None

    def forward(self, x):
        for (name, m) in self.named_children():
            if name == self.flatten_before:
                x = x.reshape(x.shape[0], -1)
            x = m(x)
        return x


class LeNet_AvgPool(_Chain):
    flatten_before = 'fc1'

    def __init__(self):
        super(LeNet_AvgPool, self).__init__()
        self.conv1 = nn.Conv2d(1, 6, kernel_size=3, stride=1, padding=1)
        self.relu1 = nn.ReLU()
        self.pool1 = nn.AvgPool2d(kernel_size=3, stride=2, padding=1)
        self.conv2 = nn.Conv2d(6, 16, kernel_size=3, stride=1, padding=1)
        self.relu2 = nn.ReLU()
        self.pool2 = nn.AvgPool2d(kernel_size=3, stride=2, padding=1)
        self.fc1 = nn.Linear(7 * 7 * 16, 120)
        self.relu3 = nn.ReLU()
        self.fc2 = nn.Linear(120, 84)
        self.relu4 = nn.ReLU()
        self.fc3 = nn.Linear(84, 10)


class AllConvNet(_Chain):
    flatten_before = 'fc1'

    def __init__(self, batchnorm=False, n_input_channels=3, n_classes=10, width=96):
        super(AllConvNet, self).__init__()
        (w1, w2) = (width, 2 * width)
        layers = OrderedDict()
        layers['dropout0'] = nn.Dropout(p=0.2)
        layers['conv1'] = nn.Conv2d(n_input_channels, w1, 3, padding=1)
        layers['relu1'] = nn.ReLU()
        layers['conv2'] = nn.Conv2d(w1, w1, 3, padding=1)
        layers['relu2'] = nn.ReLU()
        layers['conv3'] = nn.Conv2d(w1, w1, 3, padding=1, stride=2)
        if batchnorm:
            layers['conv3_bn'] = nn.BatchNorm2d(w1)
        layers['dropout3'] = nn.Dropout(p=0.5)
        layers['relu3'] = nn.ReLU()
        layers['conv4'] = nn.Conv2d(w1, w2, 3, padding=1)
        layers['relu4'] = nn.ReLU()
        layers['conv5'] = nn.Conv2d(w2, w2, 3, padding=1)
        layers['relu5'] = nn.ReLU()
        layers['conv6'] = nn.Conv2d(w2, w2, 3, padding=1, stride=2)
        if batchnorm:
            layers['conv6_bn'] = nn.BatchNorm2d(w2)
        layers['dropout6'] = nn.Dropout(p=0.5)
        layers['relu6'] = nn.ReLU()
        layers['conv7'] = nn.Conv2d(w2, w2, 3, padding=1)
        layers['relu7'] = nn.ReLU()
        layers['conv8'] = nn.Conv2d(w2, w2, 1)
        layers['relu8'] = nn.ReLU()
        layers['conv9'] = nn.Conv2d(w2, n_classes, 1)
        layers['relu9'] = nn.ReLU()
        layers['fc1'] = nn.Linear(n_classes * 8 * 8, 100)
        layers['relu10'] = nn.ReLU()
        layers['fc2'] = nn.Linear(100, 10)
        for (k, m) in layers.items():
            self.add_module(k, m)


class VGG16(_Chain):
    """VGG-16 with average pooling.  NB the KEYED pool is always AvgPool2d(3, 2, padding=1) semantics regardless of
    this module's padding/ceil_mode (keynet/layer.py:48-56); `keyed_pool_semantics=True` declares the
    pools that way so that the plain net equals the keyed net."""
    flatten_before = 'fc6'

    def __init__(self, num_classes=2622, keyed_pool_semantics=True, width=64, fc_width=4096, insize=224):
        super(VGG16, self).__init__()
        def pool():
            return nn.AvgPool2d(3, 2, 1) if keyed_pool_semantics else nn.AvgPool2d((3, 3), (2, 2), (0, 0), ceil_mode=True)
        w = width
        plan = [('1_1', 3, w), ('1_2', w, w), 'P1_2', ('2_1', w, 2 * w), ('2_2', 2 * w, 2 * w), 'P2_2',
                ('3_1', 2 * w, 4 * w), ('3_2', 4 * w, 4 * w), ('3_3', 4 * w, 4 * w), 'P3_3',
                ('4_1', 4 * w, 8 * w), ('4_2', 8 * w, 8 * w), ('4_3', 8 * w, 8 * w), 'P4_3',
                ('5_1', 8 * w, 8 * w), ('5_2', 8 * w, 8 * w), ('5_3', 8 * w, 8 * w), 'P5_3']
        for item in plan:
            if isinstance(item, str):
                self.add_module('pool' + item[1:], pool())
            else:
                (tag, cin, cout) = item
                self.add_module('conv' + tag, nn.Conv2d(cin, cout, (3, 3), (1, 1), (1, 1)))
                self.add_module('relu' + tag, nn.ReLU())
        s = insize // 32
        self.fc6 = nn.Linear(8 * w * s * s, fc_width)
        self.relu6 = nn.ReLU()
        self.dropout7 = nn.Dropout(0.5)
        self.fc7 = nn.Linear(fc_width, fc_width)
        self.relu7 = nn.ReLU()
        self.dropout8 = nn.Dropout(0.5)
        self.fc8 = nn.Linear(fc_width, num_classes)
