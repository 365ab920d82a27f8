"""Child of tests/test_host_sanitize.py: runs under LD_PRELOAD=<clang ASan runtime> with KEYNET_HIP_LIB pointing at the KN_HOST_PACK_ONLY
build of libkeynet_hip.so (host heap stands in for device memory; no compute entry point runs).  Drives every operator-create path of the C ABI
with the golden fixtures -- the same arrays the product uploads -- plus export / nnz / shape / destroy, the whole-net chain packer, and the
malformed / absurd-size calls of tests/test_capi.py.  Any heap overflow, use-after-free, signed overflow or misaligned access in the host
packing code aborts this process with a sanitizer report; the parent asserts a clean exit.  No torch in this process."""
import ctypes
import os
import sys

import numpy as np
import scipy.sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['KEYNET_HIP_NO_TORCH'] = '1'

import importlib.util                                                     # noqa: E402
spec = importlib.util.spec_from_file_location('kn_capi_hostonly', os.path.join(ROOT, 'keynet_amd', '_capi.py'))
capi = importlib.util.module_from_spec(spec)
spec.loader.exec_module(capi)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def stored_csr(p, z):
    return ((int(z[p + 'shape'][0]), int(z[p + 'shape'][1])), z[p + 'indptr'], z[p + 'indices'], z[p + 'data'].astype(np.float32))


def main():
    L = capi.lib()
    assert L.kn_abi_version() == capi.KN_ABI_VERSION
    n_ops = 0
    # ---- every golden key-net: csr / tiled / conv2dtiled operators through their create functions, export back, compare ---------------
    for name in sorted(f for f in os.listdir(GOLD) if f.endswith('.npz')):
        z = np.load(os.path.join(GOLD, name), allow_pickle=False)
        if 'layer_names' not in z.files:
            continue
        handles = []
        for lname in [str(n) for n in z['layer_names']]:
            p = 'L.%s.' % lname
            kind = str(z[p + 'kind'])
            if kind == 'relu':
                continue
            if kind == 'csr':
                (shape, ip, ix, dt) = stored_csr(p, z)
                op = capi.Operator.csr(shape, ip, ix, dt)
                (eip, eix, edt) = op.export_csr()
                assert np.array_equal(eip, ip) and np.array_equal(eix, ix) and np.array_equal(edt, dt), (name, lname)       # stored order kept
                handles.append(op)
            elif kind in ('tiled', 'diagtiled'):
                shape = tuple(int(v) for v in z[p + 'shape'])
                op = capi.Operator.tiled(shape, z[p + 'blocks'], z[p + 'tile_ptr'], z[p + 'tile_row'], z[p + 'tile_col'], z[p + 'tile_val'])
                (eip, eix, edt) = op.export_csr()
                M = scipy.sparse.csr_matrix((edt, eix, eip), shape=shape)
                if (p + 'data') in z.files:           # the reference's tocsr() of the same operator
                    R = scipy.sparse.csr_matrix((z[p + 'data'].astype(np.float32), z[p + 'indices'], z[p + 'indptr']), shape=shape)
                    assert (M != R).nnz == 0, (name, lname)
                handles.append(op)
            elif kind == 'conv2dtiled':
                shape = tuple(int(v) for v in z[p + 'shape'])
                op = capi.Operator.conv2dtiled(shape, z[p + 'inshape'], z[p + 'outshape'], z[p + 'blocks'], z[p + 'tile_keys'], z[p + 'tile_isbias'].astype(np.uint8),
                                               z[p + 'tile_chan'], z[p + 'tile_bias'])
                (eip, eix, edt) = op.export_csr()
                M = scipy.sparse.csr_matrix((edt, eix, eip), shape=shape)
                if (p + 'data') in z.files:
                    R = scipy.sparse.csr_matrix((z[p + 'data'].astype(np.float32), z[p + 'indices'], z[p + 'indptr']), shape=shape)
                    assert (M != R).nnz == 0, (name, lname)
                assert op.nnz_expanded() == M.nnz and op.shape() == shape
            else:
                raise AssertionError(kind)
            n_ops += 1
        # the whole-net packer on the CSR key-nets (sliced ELL in quads, pattern pools, slice records: pure host index work)
        if handles and len(handles) == sum(1 for n in z['layer_names'] if str(z['L.%s.kind' % str(n)]) != 'relu') and len(handles) <= 12:
            try:
                ch = capi.Operator.chain(handles, [1] * (len(handles) - 1) + [0])
                assert ch.shape()[0] == handles[-1].shape()[0]
                del ch
                n_ops += 1
            except capi.KeynetHipError as e:
                assert 'LDS' in str(e) or 'chain' in str(e), e
        del handles
    # ---- tiled cases of test/test_sparse.py:122-199 (ragged edge tiles, a dense matrix, a diagonal-tiled one, conv operators with a zero filter / bias)
    z = np.load(os.path.join(GOLD, 'tiled_cases.npz'), allow_pickle=False)
    for c in sorted(set(k.split('.')[1] for k in z.files if k.startswith('C.'))):
        p = 'C.%s.' % c
        kind = str(z[p + 'kind'])
        shape = tuple(int(v) for v in z[p + 'shape'])
        if kind == 'conv2dtiled':
            op = capi.Operator.conv2dtiled(shape, z[p + 'inshape'], z[p + 'outshape'], z[p + 'blocks'], z[p + 'tile_keys'], z[p + 'tile_isbias'].astype(np.uint8),
                                           z[p + 'tile_chan'], z[p + 'tile_bias'])
        else:
            op = capi.Operator.tiled(shape, z[p + 'blocks'], z[p + 'tile_ptr'], z[p + 'tile_row'], z[p + 'tile_col'], z[p + 'tile_val'])
        (eip, eix, edt) = op.export_csr()
        M = scipy.sparse.csr_matrix((edt, eix, eip), shape=shape)
        R = scipy.sparse.csr_matrix((z[p + 'data'].astype(np.float32), z[p + 'indices'], z[p + 'indptr']), shape=shape)
        assert (M != R).nnz == 0, c                    # == the reference's tocsr()
        n_ops += 1
    # ---- factored conv operators (direct keying) incl. coefficients, duplicate (out, in) pairs, no bias column, the small-K descriptor builder
    rng = np.random.RandomState(0)
    for (cin, cout, hw, coef, last, dup) in ((3, 64, 9, False, True, False), (16, 128, 6, True, True, False), (5, 7, 4, True, False, True), (32, 64, 5, False, True, True)):
        HW = hw * hw
        (eo, ei, et) = ([], [], [])
        for o in range(HW):
            for t in range(9):
                i = (o + 7 * t) % HW
                eo.append(o); ei.append(i if not dup else (o % 3)); et.append(t)
        taps = rng.randn(9, cout, cin).astype(np.float32)
        ec = (rng.rand(len(eo)).astype(np.float32) + 0.5) if coef else None
        lc = np.concatenate((rng.randn(cout * HW), [1.0])).astype(np.float32) if last else None
        op = capi.Operator.convtaps((cin, hw, hw), (cout, hw, hw), taps, np.array(eo, np.int32), np.array(ei, np.int32), np.array(et, np.int32), ec, lc)
        (rows, cols) = op.shape()
        assert rows == cout * HW + (1 if last else 0) and op.nnz() > 0
        if not dup:
            (eip, eix, edt) = op.export_csr()
            assert len(eix) == op.nnz_expanded()
        buf = ctypes.create_string_buffer(1024)
        assert L.kn_spmm_plan(op.handle, 256, 256, 256, 0, buf, 1024) == 0 and b'kernel' in buf.value       # the dispatch logic is host code too
        assert L.kn_spmm_plan(op.handle, 256, 256, 256, 2, buf, 1024) == 0
        assert L.kn_spmm_plan(op.handle, 256, 256, 256, 4, buf, 1024) == 0
        taps[0, 0, 0] = 0.0
        op2 = capi.Operator.convtaps((cin, hw, hw), (cout, hw, hw), taps, np.array(eo, np.int32), np.array(ei, np.int32), np.array(et, np.int32), ec, lc).drop_zero_entries()
        assert L.kn_spmm_plan(op2.handle, 256, 256, 256, 2, buf, 1024) == 0 and (dup or b'convtaps_zero_guard_kernel<1 zero tap entries>' in buf.value)
        del op2
        n_ops += 1
    # ---- dense (keyed nn.Linear on the split-K path): slicing into pseudo-pixels ----------------------------------------------------------------
    D = rng.randn(37, 513).astype(np.float32)
    D[-1, :] = 0
    D[-1, -1] = 1
    op = capi.Operator.dense(D)
    assert op.shape() == (37, 513)
    del op
    # ---- pattern groups / big groups / patched members / long rows of csr_build_groups ---------------------------------------------------------------
    cols = rng.permutation(4000)[:1500].astype(np.int32)
    (ip, ix, dt) = ([0], [], [])
    for r in range(70):                         # 64 rows sharing one unsorted sequence (a keyed Linear), 3 of them with one entry missing, 3 unrelated
        seq = cols if r < 64 else rng.permutation(4000)[:20].astype(np.int32)
        if r in (5, 17, 40):
            seq = np.delete(seq, [r])
        ix.extend(seq.tolist()); dt.extend(rng.randn(len(seq)).tolist()); ip.append(len(ix))
    op = capi.Operator.csr((70, 4000), np.array(ip, np.int32), np.array(ix, np.int32), np.array(dt, np.float32))
    (eip, eix, edt) = op.export_csr()
    assert np.array_equal(eix, np.array(ix, np.int32))
    del op
    # conv-like groups: 40 pixels x 12 rows sharing a 27-column sequence, some rows patched
    (ip, ix, dt) = ([0], [], [])
    for px in range(40):
        seq = rng.permutation(900)[:27].astype(np.int32)
        for r in range(12):
            s = np.delete(seq, [3]) if (px % 7 == 0 and r == 2) else seq
            ix.extend(s.tolist()); dt.extend(rng.randn(len(s)).tolist()); ip.append(len(ix))
    op = capi.Operator.csr((480, 900), np.array(ip, np.int32), np.array(ix, np.int32), np.array(dt, np.float32))
    del op
    n_ops += 3
    # ---- malformed and absurd-size calls: codes, never a crash ---------------------------------------------------------------------------------------
    h = ctypes.c_void_p()
    vp = (lambda a: a.ctypes.data_as(ctypes.c_void_p))
    (sip, six, sdt) = (np.array([0, 1, 2], np.int32), np.array([0, 1], np.int32), np.array([1.0, 2.0], np.float32))
    for nnz in (1 << 60, (1 << 31) + 5, -3, 1):
        assert L.kn_csr_create(2, 2, nnz, vp(sip), vp(six), vp(sdt), ctypes.byref(h)) != 0 and h.value is None
    assert L.kn_csr_create(2, 2, 2, vp(sip), vp(np.array([0, 5], np.int32)), vp(sdt), ctypes.byref(h)) == 1          # column out of range
    assert L.kn_csr_create(2, 2, 2, vp(np.array([0, 2, 1], np.int32)), vp(six), vp(sdt), ctypes.byref(h)) == 1       # indptr not monotone / does not span
    shp = np.array([1, 2, 2], np.int64)
    (blocks, keys, isb, chan, bias) = (np.zeros((1, 3), np.int64), np.zeros((1, 3), np.int64), np.zeros(1, np.uint8), np.ones((1, 1, 1), np.float32), np.zeros(1, np.float32))
    assert L.kn_conv2dtiled_create(4, 4, vp(shp), vp(shp), 1, vp(blocks), 1 << 60, vp(keys), vp(isb), vp(chan), vp(bias), ctypes.byref(h)) == 4
    assert L.kn_conv2dtiled_create(4, 4, vp(shp), vp(shp), -1, vp(blocks), 1, vp(keys), vp(isb), vp(chan), vp(bias), ctypes.byref(h)) != 0
    tp = np.array([0, 1], np.int64)
    (tr, tc, tv) = (np.array([9], np.int32), np.array([0], np.int32), np.array([1.0], np.float32))
    assert L.kn_tiled_create(4, 4, 1, vp(np.array([[0, 0, 0]], np.int64)), 1, vp(tp), vp(tr), vp(tc), vp(tv), ctypes.byref(h)) != 0          # tile entry outside the matrix
    assert L.kn_tiled_create(4, 4, 1, vp(np.array([[0, 0, 3]], np.int64)), 1, vp(tp), vp(tr), vp(tc), vp(tv), ctypes.byref(h)) != 0          # block names a tile that does not exist
    assert L.kn_tiled_create(4, 4, -1, vp(np.array([[0, 0, 0]], np.int64)), 1, vp(tp), vp(tr), vp(tc), vp(tv), ctypes.byref(h)) != 0
    i3 = np.array([1, 2, 2], np.int64)
    e = np.array([0], np.int32)
    assert L.kn_convtaps_create(vp(i3), vp(i3), 1, vp(np.ones(1, np.float32)), 1, vp(e), vp(np.array([9], np.int32)), vp(e), None, None, ctypes.byref(h)) != 0     # input pixel out of range
    assert L.kn_convtaps_create(vp(i3), vp(i3), 1, vp(np.ones(1, np.float32)), 1, vp(e), vp(e), vp(np.array([4], np.int32)), None, None, ctypes.byref(h)) != 0     # tap id out of range
    assert L.kn_convtaps_create(vp(np.array([1, -2, 2], np.int64)), vp(i3), 1, vp(np.ones(1, np.float32)), 1, vp(e), vp(e), vp(e), None, None, ctypes.byref(h)) != 0
    # (sizes that fail only inside operator new are tests/test_capi.py's: ASan's allocator aborts on them instead of throwing std::bad_alloc)
    assert L.kn_dense_create(1 << 40, 1 << 40, vp(np.ones(1, np.float32)), ctypes.byref(h)) != 0
    assert L.kn_chain_create(1 << 50, None, None, ctypes.byref(h)) != 0
    assert L.kn_destroy(None) in (0, 1)
    # compute entry points refuse in this build (and on any box without a device)
    ok = capi.Operator.csr((2, 2), sip, six, sdt)
    assert L.kn_spmm(ok.handle, 4096, 4, 4, 8192, 4, 0, None) == 5
    assert L.kn_reserve_workspace(ok.handle, 8, None) == 0                       # nothing to reserve for a CSR handle
    del ok
    print('HOST_SANITIZE_OK operators=%d' % n_ops)


if __name__ == '__main__':
    main()
