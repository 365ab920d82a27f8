"""Static checks on the gfx950 ISA of the hand-scheduled kernels (hipcc cross-compiles without a GPU).

The software-pipelined kernels issue their loads as inline asm and wait for them with explicit, counted `s_waitcnt`s, so the
compiler knows neither that a destination register is still owned by a load in flight nor where the data becomes valid.  Two ways
this has gone wrong during development, both silent at run time on most inputs: (i) a register copy of a destination between the
load and its wait (live-range splitting / phi copies around a branch) reads the register before the data has landed; (ii) an `"s"`
operand that the compiler kept in vector registers is emitted as a VGPR pair (caught by the assembler, i.e. by the build).  This
test pins (i): in those kernels no v_mov inside a loop body (from its first asm-issued load to the first store of its epilogue) may read a
register that is the destination of an asm-issued global load, and nothing may spill to scratch."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'keynet_amd', 'csrc')

PIPELINED = {
    'kn_conv.hip': [r'_ZN2kn26convtaps_exact_pipe_kernel', r'_ZN2kn20convtaps_mfma_kernelILi\d+ELi\d+ELi16ELi\dELi\dELi2E'],
    'kn_csr.hip': [r'_ZN2kn21csr_group_pipe_kernel'],
    'kn_csr_mfma.hip': [r'_ZN2kn21csr_group_mfma_kernel', r'_ZN2kn23csr_group_mfma16_kernel'],
}

# what an asm-issued load looks like per file (the compiler's own saddr-form dword loads in the other files are tracked by its waitcnt pass)
ASM_LOAD = {'default': r'global_load_dwordx4 (v\[\d+:\d+\]), v\d+, s\[', 'kn_csr_mfma.hip': r'global_load_dword (v\d+), v\d+, s\['}


def _isa(src, tmp_path):
    out = os.path.join(str(tmp_path), src + '.s')
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-S', '--cuda-device-only',
                           os.path.join(CSRC, src), '-o', out], stderr=subprocess.DEVNULL)
    return open(out).read()


def _regs(txt, prefix):
    found = set()
    for m in re.finditer(prefix + r'\[(\d+):(\d+)\]', txt):
        found.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'(?<![\w\[:])' + prefix + r'(\d+)\b', txt):
        found.add(int(m.group(1)))
    return found


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='needs hipcc')
@pytest.mark.parametrize('src', sorted(PIPELINED))
def test_no_copy_of_a_register_owned_by_a_load_in_flight(src, tmp_path):
    s = _isa(src, tmp_path)
    # nothing in the file spills
    for m in re.finditer(r'\.private_segment_fixed_size:\s+(\d+)', s):
        assert int(m.group(1)) == 0, 'a kernel of %s uses scratch' % src
    checked = 0
    for pat in PIPELINED[src]:
        names = re.findall(r'^(%s[^\n:]*):' % pat, s, re.M)
        assert names, 'no kernel matches %s in %s' % (pat, src)
        for name in names:
            body = s[s.index(name + ':'):]
            body = body[:body.index('.Lfunc_end')]                # (not the first s_endpgm: an early wave-uniform return may be laid out ahead of the loops)
            lines = [l.split(';')[0].strip() for l in body.split('\n') if l.strip()]
            # In program order: the destinations of the asm-issued loads of a loop body are "owned" from the body's first such load until its
            # epilogue begins (= the first global store behind it: the loops themselves store nothing, and every operand has landed and been
            # consumed by then).  A register copy out of an owned register is the bug; a copy in the epilogue of a register that merely WAS a
            # load destination (the allocator reuses registers there: the running max of kn_spmm_screen, for one) is not.  Kernels with a
            # quarter-tile tail hold two bodies in sequence: an asm-issued load behind an epilogue opens the next body.
            dests = set()
            for l in lines:
                m = re.match(ASM_LOAD.get(src, ASM_LOAD['default']), l)
                if m:
                    dests |= _regs(m.group(1), 'v')
            assert dests, name
            in_body = False
            for l in lines:
                if re.match(ASM_LOAD.get(src, ASM_LOAD['default']), l):
                    in_body = True
                elif l.startswith('global_store') or l.startswith('buffer_store'):
                    in_body = False
                if in_body and (l.startswith('v_mov') or l.startswith('v_accvgpr')):
                    srcs = ','.join(l.split(',')[1:])
                    assert not (_regs(srcs, 'v') & dests), (name, l)
            checked += 1
    assert checked >= len(PIPELINED[src])
