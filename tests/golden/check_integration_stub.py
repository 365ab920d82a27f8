#!/usr/bin/env python3
"""Build-container check that the `keynet/hip.py` stub printed in INTEGRATION.md is real code: it is extracted from the document,
executed against the REFERENCE's own container classes (imported through _refimport), and every `wrap()`-ed operator must marshal
its arrays through the C ABI's argument validation -- the call may only stop at the device check ('no HIP device', this container
has none) or succeed.  Prints 'STUB OK'."""
import os
import re
import sys
import types
import numpy as np
import scipy.sparse

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

keynet = _refimport.import_reference()
ks = keynet.sparse


def main():
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    code = re.findall(r"```python\n(# keynet/hip.py.*?)```", doc, re.S)[0]
    code = code.replace("ctypes.CDLL('libkeynet_hip.so')", "ctypes.CDLL(%r)" % os.path.join(ROOT, 'keynet_amd', 'libkeynet_hip.so'))
    mod = types.ModuleType('keynet.hip')
    exec(compile(code, 'keynet/hip.py', 'exec'), mod.__dict__)
    rng = np.random.RandomState(0)
    Wc = ks.sparse_toeplitz_conv2d((2, 8, 8), rng.rand(3, 2, 3, 3).astype(np.float32), bias=rng.rand(3).astype(np.float32))
    objs = [ks.SparseMatrix(scipy.sparse.random(5, 7, 0.5, format='csr', dtype=np.float32, random_state=1)),
            ks.SparseMatrix(scipy.sparse.random(5, 7, 0.5, format='csr', dtype=np.float64, random_state=3)),       # a float64 operator: kn_csr_create_f64
            ks.TiledMatrix(scipy.sparse.random(8, 8, 0.5, format='coo', dtype=np.float32, random_state=2), (4, 4)),
            ks.DiagonalTiledMatrix(rng.rand(3, 3).astype(np.float32), (10, 10)),
            ks.Conv2dTiledMatrix(Wc, (2, 8, 8), (3, 8, 8), (4, 4), bias=True)]
    for W in objs:
        ref_type = type(W)
        o = mod.wrap(W)
        assert isinstance(o, ref_type) and hasattr(o, '_create') and 'torchdot' in type(o).__mro__[1].__dict__
        try:
            o._create()
        except RuntimeError as e:
            assert 'no HIP device' in str(e), str(e)          # argument validation passed; only the device is missing here
        assert '_kn_handles' not in o.__getstate__()
    print('STUB OK')


if __name__ == '__main__':
    os.chdir('/tmp')
    main()
