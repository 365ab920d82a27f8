#!/usr/bin/env python3
"""Launch the LeNet whole-net kernel a few times at N images (for rocprofv3 --pmc passes: tools/chain_pmc.sh)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keynet_amd import system as ksys       # noqa: E402
from keynet_amd.models import LeNet_AvgPool  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
net = LeNet_AvgPool().eval()
np.random.seed(0)
(sensor, knet) = ksys.PermutationKeynet((1, 28, 28), net)
x = sensor.fromtensor(torch.randn(n, 1, 28, 28, device='cuda:0')).encrypt().astensor()
for _ in range(6):
    y = knet.forward_linear(x)
torch.cuda.synchronize()
print(float(y.abs().sum()))
