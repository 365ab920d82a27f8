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
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall', '-Wextra', '-Wno-unused-parameter'] + list(extra) + ['-D' + d for d in defines]
    # one object per source, compiled concurrently (the sources share no device symbols: no -fgpu-rdc), then one link.  Objects of the product build are kept
    # under keynet_amd/build/ (git-ignored) and re-used when neither the source nor a header is newer; variant builds use a throw-away directory.
    import hashlib
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    variant = out is not None or defines or extra
    objdir = tempfile.mkdtemp(prefix='kn_obj_') if variant else os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, 'kn_internal.h'), os.path.join(HERE, '..', 'include', 'keynet_hip.h')]
    tag = hashlib.sha256(' '.join(flags).encode()).hexdigest()[:8]

    def compile_one(src):
        obj = os.path.join(objdir, '%s.%s.o' % (os.path.splitext(src)[0], tag))
        deps = [os.path.join(CSRC, src)] + headers
        if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
            return obj
        cmd = [hipcc] + flags + ['-c', '-o', obj, os.path.join(CSRC, src)]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
        return obj

    try:
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
            objs = list(ex.map(compile_one, SOURCES))
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + list(extra) + ['-o', out or LIB] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    finally:
        if variant:
            import shutil
            shutil.rmtree(objdir, ignore_errors=True)
    return out or LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
