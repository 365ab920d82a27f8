"""Builds libkeynet_hip.so in-tree (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['kn_api.hip', 'kn_csr.hip', 'kn_csr_f64.hip', 'kn_csr_mfma.hip', 'kn_conv.hip', 'kn_elementwise.hip', 'kn_chain.hip']
LIB = os.path.join(HERE, 'libkeynet_hip.so')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, 'kn_internal.h'), os.path.join(HERE, '..', 'include', 'keynet_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, out=None, defines=(), extra=()):
    """hipcc --offload-arch=gfx950 ... -> keynet_amd/libkeynet_hip.so.  -ffp-contract=off: the order-preserving kernels
    must round the product and the sum separately (bit-exact with scipy's csr_matvecs).  `out` / `defines`: diagnostic variants
    (tools/ablate_conv.sh builds one with -DKN_ABLATION next to the product library; tests/test_host_sanitize.py builds the host side with
    -DKN_HOST_PACK_ONLY and `extra` = the sanitizer flags; nothing loads either by default)."""
    if out is None and not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', 'hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
           '-Wall', '-Wextra', '-Wno-unused-parameter'] + list(extra) + ['-D' + d for d in defines] + ['-o', out or LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return out or LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
