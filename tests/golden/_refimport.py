"""Import shim used ONLY by tests/golden/make_golden.py (runs in the build container, never on the GPU box).

The reference (/root/reference, Python) imports three packages that are absent from this image at module
top level (numba, vipy, torchvision: keynet/sparse.py:14-22, keynet/system.py:7-9, keynet/mnist.py:3-8).
This module writes throw-away stand-ins for those *third-party* packages into a temp dir (they contain no
reference code: `numba.jit` = identity decorator, etc.), puts that dir and /root/reference on sys.path and
imports the reference.  Nothing here travels to the GPU box; only the .npz vectors it helps generate do.
"""
import os
import sys
import tempfile
import textwrap

REFERENCE = '/root/reference'

_STUBS = {
    'numba/__init__.py': '''
        def jit(*args, **kwargs):
            if len(args) == 1 and callable(args[0]) and not kwargs:
                return args[0]
            def deco(f):
                return f
            return deco
        njit = jit
        from . import typed
    ''',
    'numba/typed.py': '''
        class List(list):
            pass
        class Dict(dict):
            pass
    ''',
    'vipy/__init__.py': '''
        from . import util
        from . import image
    ''',
    'vipy/util.py': '''
        import time, tempfile, pickle
        from collections import defaultdict
        def try_import(*args, **kwargs):
            return None
        def tolist(x):
            return list(x) if isinstance(x, (list, tuple)) else [x]
        class Stopwatch(object):
            def __init__(self):
                self._t = time.time(); self.elapsed = 0.0
            def since(self):
                return time.time() - self._t
            def __enter__(self):
                self._t = time.time(); return self
            def __exit__(self, *a):
                self.elapsed = time.time() - self._t
        def tempdir():
            return tempfile.gettempdir()
        def groupbyasdict(xs, f):
            d = defaultdict(list)
            for x in xs:
                d[f(x)].append(x)
            return dict(d)
        def flatlist(xs):
            return [y for x in xs for y in x]
        def save(obj, f):
            with open(f, 'wb') as fh:
                pickle.dump(obj, fh)
            return f
        def load(f):
            with open(f, 'rb') as fh:
                return pickle.load(fh)
    ''',
    'vipy/image.py': '''
        class Image(object):
            def __init__(self, *a, **k):
                raise NotImplementedError('vipy is not installed (golden generator stub)')
    ''',
    'torchvision/__init__.py': '''
        from . import datasets
        from . import transforms
    ''',
    'torchvision/datasets.py': '''
    ''',
    'torchvision/transforms.py': '''
        class _T(object):
            def __init__(self, *a, **k):
                pass
            def __call__(self, x):
                return x
        class Compose(_T): pass
        class ToTensor(_T): pass
        class Normalize(_T): pass
        class RandomCrop(_T): pass
        class RandomHorizontalFlip(_T): pass
        class Resize(_T): pass
        class CenterCrop(_T): pass
        class Lambda(_T): pass
        class Grayscale(_T): pass
    ''',
}


def import_reference():
    """Returns the imported `keynet` reference package (with .sparse/.system/.layer/.torch/.mnist/.cifar10/.vgg loaded)."""
    assert os.path.isdir(REFERENCE), 'reference not mounted: golden vectors can only be regenerated in the build container'
    sys.dont_write_bytecode = True
    d = tempfile.mkdtemp(prefix='kn_stubs_')
    for (rel, src) in _STUBS.items():
        p = os.path.join(d, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, 'w') as f:
            f.write(textwrap.dedent(src))
    sys.path.insert(0, REFERENCE)
    sys.path.insert(0, d)
    import keynet
    import keynet.globals
    import keynet.sparse, keynet.torch, keynet.layer, keynet.system, keynet.util  # noqa
    import keynet.mnist, keynet.cifar10, keynet.vgg  # noqa
    keynet.globals.verbose(False)
    return keynet
