#!/usr/bin/env python3
"""One VGG-shaped conv layer through the f32 MFMA kernel and the bf16x3 kernel, a few launches each (for rocprofv3 --pmc passes)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bf16x3_bench import make, dev    # noqa: E402  (runs its table first)
W = make(512, 512, 28)
x = torch.randn(512 * 28 * 28 + 1, 256, device=dev).clamp_(min=0)
for mode in (False, 'bf16x3'):
    for _ in range(4):
        y = W.torchdot(x, relu=True, exact=mode)
torch.cuda.synchronize()
print(float(y.sum()))
