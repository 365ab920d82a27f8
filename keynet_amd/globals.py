"""Process-wide switches (the reference keeps the same two in keynet/globals.py:4,31-34)."""
GLOBAL = {'VERBOSE': False}


def verbose(b=None):
    """Get/set the per-layer print switch read by KeyedLayer.forward (reference default True; ours False: a print per
    layer per call would serialise the stream-ordered GPU forward)."""
    if b is not None:
        GLOBAL['VERBOSE'] = bool(b)
    return GLOBAL['VERBOSE']


def backend():
    return 'hip'
