#!/usr/bin/env python3
"""Convert a REFERENCE pickle of a keyed network into the neutral .npz container of keynet_amd.io.

    cd /tmp && python /root/repo/tests/golden/import_pickle.py <in.pkl> <out.npz>

The reference ships key-nets as whole-object pickles (test/test_keynet.py:106-107 `vipy.util.save((sensor, knet), f)`,
keynet/system.py:147-151 `public()`, demo/challenge.ipynb cell 1), which need the reference's classes to load.  This script
runs where the reference is importable (the build container: tests/golden/_refimport.py) and writes only DATA: layer order,
operator arrays in stored order, ReLU markers, and the sensor's key pair when the pickle holds one.  The result loads with
keynet_amd.io.load_keynet on a box that has never seen the reference.
"""
import os
import pickle
import sys
import numpy as np
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference through _refimport; shares dump_operator / csr_triplet)

keynet = mg.keynet


def convert(pkl, npz):
    with open(pkl, 'rb') as f:
        obj = pickle.load(f)
    (sensor, knet) = (None, obj)
    if isinstance(obj, (tuple, list)):
        knet = [o for o in obj if hasattr(o, '_keynet')][0]
        sensors = [o for o in obj if isinstance(o, keynet.system.KeyedSensor)]
        sensor = sensors[0] if sensors else None
    out = {}
    names = []
    for (name, child) in knet._keynet.named_children():
        names.append(name)
        if isinstance(child, keynet.layer.KeyedLayer):
            mg.dump_operator('L.%s.' % name, child.W, out)
            out['L.%s.layertype' % name] = np.array(child._layertype)
        else:
            assert isinstance(child, nn.ReLU), 'unexpected module %s in a key-net' % str(type(child))
            out['L.%s.kind' % name] = np.array('relu')
    out['layer_names'] = np.array(names)
    if getattr(knet, '_outshape', None) is not None:
        out['outshape'] = np.array(knet._outshape, dtype=np.int64)
    if sensor is not None and getattr(sensor, '_encryptkey', None) is not None:
        for (tag, M) in (('enc', sensor._encryptkey), ('dec', sensor._decryptkey)):
            (ip, ix, dt) = mg.csr_triplet(M)
            (out['sensor.%s.indptr' % tag], out['sensor.%s.indices' % tag], out['sensor.%s.data' % tag]) = (ip, ix, dt)
        out['sensor.shape'] = np.array(sensor._encryptkey.shape, dtype=np.int64)
        out['sensor.inshape'] = np.array(sensor._inshape[1:], dtype=np.int64)
    np.savez_compressed(npz, **out)
    print('[import_pickle]: %s -> %s (%d layers, sensor keys: %s, %d bytes)' % (pkl, npz, len(names), 'yes' if 'sensor.shape' in out else 'no', os.path.getsize(npz)))
    return npz


if __name__ == '__main__':
    assert len(sys.argv) == 3, __doc__
    convert(os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2]))
