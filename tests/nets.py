"""Test nets with the same architecture/naming as the ones tests/golden/make_golden.py keyed with the reference."""
import numpy as np
import torch
from torch import nn
from keynet_amd.models import _Chain, LeNet_AvgPool  # noqa: F401


class MiniNet(_Chain):
    flatten_before = 'fc1'

    def __init__(self):
        super(MiniNet, self).__init__()
        self.conv1 = nn.Conv2d(2, 4, 3, stride=1, padding=1)
        self.relu1 = nn.ReLU()
        self.pool1 = nn.AvgPool2d(3, stride=2, padding=1)
        self.conv2 = nn.Conv2d(4, 4, 3, stride=1, padding=1)
        self.relu2 = nn.ReLU()
        self.pool2 = nn.AvgPool2d(3, stride=2, padding=1)
        self.fc1 = nn.Linear(4 * 4 * 4, 10)


class TinyAllConv(_Chain):
    flatten_before = 'fc1'

    def __init__(self):
        super(TinyAllConv, self).__init__()
        self.dropout0 = nn.Dropout(p=0.2)
        self.conv1 = nn.Conv2d(3, 6, 3, padding=1)
        self.relu1 = nn.ReLU()
        self.conv2 = nn.Conv2d(6, 6, 3, padding=1)
        self.relu2 = nn.ReLU()
        self.conv3 = nn.Conv2d(6, 6, 3, padding=1, stride=2)
        self.dropout3 = nn.Dropout(p=0.5)
        self.relu3 = nn.ReLU()
        self.conv4 = nn.Conv2d(6, 8, 3, padding=1)
        self.relu4 = nn.ReLU()
        self.conv6 = nn.Conv2d(8, 8, 3, padding=1, stride=2)
        self.dropout6 = nn.Dropout(p=0.5)
        self.relu6 = nn.ReLU()
        self.conv8 = nn.Conv2d(8, 8, 1)
        self.relu8 = nn.ReLU()
        self.conv9 = nn.Conv2d(8, 4, 1)
        self.relu9 = nn.ReLU()
        self.fc1 = nn.Linear(4 * 4 * 4, 12)
        self.relu10 = nn.ReLU()
        self.fc2 = nn.Linear(12, 10)


class TinyBN(_Chain):
    """conv -> '<conv>_bn' -> dropout -> relu (the batch-norm placement of AllConvNet(batchnorm=True)), see make_golden.py f7."""
    flatten_before = 'fc1'

    def __init__(self):
        super(TinyBN, self).__init__()
        self.conv1 = nn.Conv2d(3, 6, 3, padding=1)
        self.relu1 = nn.ReLU()
        self.conv3 = nn.Conv2d(6, 8, 3, padding=1, stride=2)
        self.conv3_bn = nn.BatchNorm2d(8)
        self.dropout3 = nn.Dropout(p=0.5)
        self.relu3 = nn.ReLU()
        self.conv4 = nn.Conv2d(8, 5, 3, padding=1)
        self.conv4_bn = nn.BatchNorm2d(5)
        self.relu4 = nn.ReLU()
        self.fc1 = nn.Linear(5 * 4 * 4, 7)


def load_weights(net, z):
    """Copy the 'net.*' arrays of a golden file into `net` (same parameter names)."""
    sd = {k[4:]: torch.as_tensor(np.array(z[k])) for k in z.files if k.startswith('net.')}
    net.load_state_dict(sd)
    return net.eval()
