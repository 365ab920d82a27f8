"""Shared by bench.py and its legs: repository root, stderr logging, the peaks the roofline fractions are priced against, small helpers."""
import hashlib
import os
import sys

import numpy as np

from keynet_amd import sparse as ksp
from keynet_amd.layer import KeyedLayer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md:42 (dense f32-input MFMA = f32 vector peak)
PEAK_VALU_NOFMA_TMACS = 39.3     # the same vector peak with separate multiply and add (bit-exact contract): 157.3 / 4 T MAC/s
PEAK_L2_READ_GBS_MEASURED = 28300.0   # all 256 CUs reading one L2-resident array (2.7 MB, 16-byte loads): tools/micro/l2_read_rate.hip, profiles/r06_micro_l2_read_rate.txt (no guide figure)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md:36 (spec; 6.29 TB/s measured copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def keyed_layers(knet):
    return [(n, c) for (n, c) in knet._keynet.named_children() if isinstance(c, KeyedLayer)]


def host_nnz(c):
    """nnz of the operator the reference would apply (= algorithmic MACs per image), from the host description alone."""
    W = c.W
    if isinstance(W, ksp.Conv2dTiledMatrix) and W._taps is not None:
        t = W._taps
        return int(len(t['ent_out'])) * W._outshape[0] * W._inshape[0] + (int(np.count_nonzero(t['lastcol'])) if t['lastcol'] is not None else 0)
    if isinstance(W, ksp.TiledMatrix):
        return int(W.tocsr().nnz)
    return int(W.nnz())


def kernel_sources_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'keynet_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()


def _num(v, sig=6):
    """Floats to `sig` significant digits (the line is a summary; bench_detail.json keeps full precision)."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        return float('%.*g' % (sig, v)) if np.isfinite(v) else None
    if isinstance(v, dict):
        return {k: _num(x, sig) for (k, x) in v.items()}
    if isinstance(v, (list, tuple)):
        return [_num(x, sig) for x in v]
    return v


def _pick(d, keys):
    return {k: d.get(k) for k in keys if isinstance(d, dict) and k in d}


def _clip(s, n):
    return s if (not isinstance(s, str) or len(s) <= n) else s[:n - 3] + '...'
