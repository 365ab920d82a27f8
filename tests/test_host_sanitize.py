"""Sanitizer pass over the HOST code (SURVEY section 5 "sanitizers"; GPU AddressSanitizer does not exist on this pool, so this is the CPU
build only): (1) the operator-packing side of libkeynet_hip.so -- thousands of lines of std::vector index work on caller-sized inputs in
kn_api.hip / kn_csr.hip / kn_conv.hip / kn_chain.hip -- compiled with hipcc -fsanitize=address,undefined behind -DKN_HOST_PACK_ONLY (host
heap stands in for device memory, compute entry points refuse) and driven with the golden fixtures and the absurd-size ABI calls;
(2) oracle/kn_oracle.c, the checker everything else is compared with, under gcc's ASan + UBSan on the golden vectors."""
import glob
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAD = ('ERROR: AddressSanitizer', 'runtime error:', 'ERROR: LeakSanitizer', 'AddressSanitizer:DEADLYSIGNAL')


def _clang_rt(name):
    hits = sorted(glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.%s-x86_64.so' % name))
    return hits[-1] if hits else None


@pytest.mark.skipif(shutil.which('hipcc') is None or _clang_rt('asan') is None, reason='needs hipcc and its host sanitizer runtimes')
def test_operator_packing_under_asan_ubsan(tmp_path):
    from keynet_amd import build as kbuild
    lib = str(tmp_path / 'libkeynet_hip_hostsan.so')
    kbuild.build(out=lib, defines=('KN_HOST_PACK_ONLY',),
                 extra=('-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer', '-g', '-O1', '-shared-libsan'))
    env = dict(os.environ, KEYNET_HIP_LIB=lib, KEYNET_HIP_NO_TORCH='1', LD_PRELOAD=_clang_rt('asan'),
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:halt_on_error=1:allocator_may_return_null=1', UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')
    env['LD_LIBRARY_PATH'] = os.path.dirname(_clang_rt('asan')) + ':' + env.get('LD_LIBRARY_PATH', '')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'host_sanitize_child.py')], env=env, capture_output=True, text=True, timeout=1200)
    out = p.stdout + p.stderr
    assert not any(b in out for b in BAD), out[-6000:]
    assert p.returncode == 0 and 'HOST_SANITIZE_OK' in p.stdout, out[-6000:]


@pytest.mark.skipif(shutil.which('gcc') is None, reason='needs gcc')
def test_oracle_c_under_asan_ubsan(tmp_path):
    """oracle/kn_oracle.c (scipy's csr_matvecs restated) under ASan + UBSan: the LeNet golden key-net through every layer (bit-equal to
    the reference's vectors), an empty matrix, empty rows and a single huge row."""
    so = str(tmp_path / 'libkn_oracle_san.so')
    subprocess.check_call(['gcc', '-O1', '-g', '-fPIC', '-shared', '-ffp-contract=off', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined',
                           '-fno-omit-frame-pointer', '-o', so, os.path.join(ROOT, 'oracle', 'kn_oracle.c')])
    asan = subprocess.check_output(['gcc', '-print-file-name=libasan.so'], text=True).strip()
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
os.environ['KN_ORACLE_LIB'] = %r
import oracle
z = np.load(os.path.join(%r, 'tests', 'golden', 'lenet_perm.npz'), allow_pickle=False)
y = oracle.keynet_forward(oracle.load_golden_layers(z), z['x_cipher'])
assert np.array_equal(y, z['Y.fc3'])
e = oracle.csr_matvecs((0, 3), np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), np.ones((3, 2), np.float32))
assert e.shape == (0, 2)
r = oracle.csr_matvecs((3, 4), np.array([0, 0, 4, 4], np.int32), np.array([3, 0, 0, 2], np.int32), np.array([1, 2, 3, 4], np.float32), np.ones((4, 5), np.float32))
assert np.array_equal(r[1], np.full(5, 10, np.float32)) and not r[0].any() and not r[2].any()
big = np.random.RandomState(0).randint(0, 50, size=100000).astype(np.int32)
oracle.csr_matvecs((1, 50), np.array([0, 100000], np.int32), big, np.ones(100000, np.float32), np.ones((50, 3), np.float32))
print('ORACLE_SANITIZE_OK')
''' % (ROOT, so, ROOT)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:halt_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    p = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    out = p.stdout + p.stderr
    assert not any(b in out for b in BAD), out[-4000:]
    assert p.returncode == 0 and 'ORACLE_SANITIZE_OK' in p.stdout, out[-4000:]
