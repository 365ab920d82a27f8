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
    'kn_conv.hip': [r'_ZN2kn26convtaps_exact_pipe_kernel', r'_ZN2kn20convtaps_mfma_kernelILi\d+ELi\d+ELi16ELi\dELi\dELi2E',
                    (r'_ZN2kn26convtaps_exact_fill_kernel', r'global_load_dword(?:x2)? (v\d+|v\[\d+:\d+\]), v\d+, s\[')],      # (pattern, what ITS asm-issued loads look like; two column tiles: 8-byte loads)
    'kn_csr.hip': [r'_ZN2kn21csr_group_pipe_kernel'],
    'kn_csr_mfma.hip': [r'_ZN2kn21csr_group_mfma_kernel', r'_ZN2kn23csr_group_mfma16_kernel'],
}

# what an asm-issued load looks like per file (the compiler's own saddr-form dword loads in the other files are tracked by its waitcnt pass)
ASM_LOAD = {'default': r'global_load_dwordx[24] (v\[\d+:\d+\]), v\d+, s\[', 'kn_csr_mfma.hip': r'global_load_dword (v\d+), v\d+, s\['}


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
        load_re = ASM_LOAD.get(src, ASM_LOAD['default'])
        if isinstance(pat, tuple):
            (pat, load_re) = pat
        names = re.findall(r'^(%s[^\n:]*):' % pat, s, re.M)
        assert names, 'no kernel matches %s in %s' % (pat, src)
        for name in names:
            body = s[s.index(name + ':'):]
            body = body[:body.index('.Lfunc_end')]                # (not the first s_endpgm: an early wave-uniform return may be laid out ahead of the loops)
            lines = [l.split(';')[0].strip() for l in body.split('\n') if l.strip()]
            # Which registers are OWNED by an asm-issued load in flight, at every instruction: a forward data-flow over the kernel's code.  An asm-issued load
            # adds its destination (youngest); an explicit `s_waitcnt vmcnt(N)` -- the waits of these kernels are written out, loads return in order -- leaves
            # only the N youngest; the state at a label is the union of the fall-through state and of the states at every branch to it (loops: iterated to a
            # fixed point).  A register copy (v_mov / v_accvgpr) out of an owned register is the bug.  (Until round 4 the check took every register that is
            # EVER a load destination in the kernel as owned throughout a loop body; with more code paths per kernel the allocator re-uses such registers for
            # constants after their loads have long landed, which that rule flagged.)
            pat_load = re.compile(load_re)
            labels = {l[:-1]: i for (i, l) in enumerate(lines) if l.endswith(':')}
            state_at = {}                                          # label -> tuple of register-sets, oldest load first
            n_loads = sum(1 for l in lines if pat_load.match(l))
            assert n_loads, name

            def merge(x, y):                                       # union of two in-flight lists, keeping an order (longest first: conservative for counted waits)
                (x, y) = (list(x), list(y))
                if len(x) < len(y):
                    (x, y) = (y, x)
                out = list(x)
                for (k, r) in enumerate(y):                        # align the YOUNGEST ends
                    j = len(out) - len(y) + k
                    out[j] = out[j] | r
                return tuple(frozenset(r) for r in out)

            changed = True
            passes = 0
            while changed and passes < 12:
                changed = False
                passes += 1
                fly = ()
                epilogue = False                                   # in program order, from the first store behind a loop body to the next asm-issued load
                for (i, l) in enumerate(lines):
                    if epilogue and not pat_load.match(l):
                        fly = ()
                        continue
                    if l.endswith(':'):
                        if l[:-1] in state_at:
                            fly = merge(fly, state_at[l[:-1]])
                        continue
                    m = pat_load.match(l)
                    if m:
                        epilogue = False
                        fly = fly + (frozenset(_regs(m.group(1), 'v')),)
                        continue
                    m = re.match(r's_waitcnt .*vmcnt\((\d+)\)', l)
                    if m:
                        n = int(m.group(1))
                        fly = fly[len(fly) - n:] if n else ()
                        continue
                    if l.startswith('global_store') or l.startswith('buffer_store'):
                        epilogue = True                            # an epilogue begins: the loops store nothing, every operand has landed and been consumed by then
                        fly = ()                                   # (the analysis is path-insensitive: "no next chunk, so no load" and "no next chunk, so no wait" look
                        continue                                   # independent to it, and an epilogue's stores sit behind branches of their own)
                    if l.startswith('v_mov') or l.startswith('v_accvgpr'):
                        owned = set().union(*fly) if fly else set()
                        srcs = ','.join(l.split(',')[1:])
                        assert not (_regs(srcs, 'v') & owned), (name, l)
                    m = re.match(r's_c?branch\w*\s+(\S+)', l)
                    if m and m.group(1) in labels:
                        t = m.group(1)
                        new = merge(state_at.get(t, ()), fly)
                        if new != state_at.get(t, ()):
                            state_at[t] = new
                            changed = True
                        if l.startswith('s_branch'):
                            fly = ()                               # (unreachable by fall-through)
                    if l.startswith('s_endpgm'):
                        fly = ()
            assert passes < 12, name
            checked += 1
    assert checked >= len(PIPELINED[src])


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='needs hipcc')
def test_no_use_of_a_scalar_register_owned_by_a_scalar_load_in_flight(tmp_path):
    """convtaps_exact_fill_kernel fetches its slot records with asm-issued s_load_dwordx16 one batch ahead and waits with an explicit s_waitcnt lgkmcnt(0).
    The compiler takes the destination tuple as defined the moment the load is ISSUED: a copy of it ahead of the wait (seen once: loop-carried tuples
    were moved at the loop header while their load was still in flight -- stale records, wild addresses) is the scalar twin of the bug the test above
    looks for.  Between every such load and the next lgkmcnt(0) nothing may read or write its destination registers."""
    s = _isa('kn_conv.hip', tmp_path)
    names = re.findall(r'^(_ZN2kn26convtaps_exact_fill_kernel[^\n:]*):', s, re.M)
    assert len(names) == 5, names                                # (taps in registers: 64 and 32 channels per wavefront, each with one and two column tiles; one value-row load per slot)
    for name in names:
        body = s[s.index(name + ':'):]
        body = body[:body.index('.Lfunc_end')]
        owned = set()
        n_loads = 0
        for l in (x.split(';')[0].strip() for x in body.split('\n')[1:]):
            if not l or l.endswith(':') or l.startswith('.'):
                continue
            if l.startswith('s_waitcnt') and 'lgkmcnt(0)' in l:
                owned = set()
                continue
            m = re.match(r's_load_dwordx(?:4|8|16) s\[(\d+):(\d+)\]', l)
            if m:
                owned |= set(range(int(m.group(1)), int(m.group(2)) + 1))
                n_loads += 1
                continue
            assert not (_regs(l, 's') & owned), (name, l)
        assert n_loads >= 6, (name, n_loads)


# ---- the bit-exact contract in the ISA -----------------------------------------------------------------------------------------------------
# The order-preserving kernels must round every product and every sum separately (scipy's csr_matvecs: `y += a * x` compiled without
# contraction; oracle/kn_oracle.c).  That hangs on -ffp-contract=off + `#pragma clang fp contract(off)`: one flag regression would pass every
# CPU test and fail on the GPU box only.  So: no fused multiply-add of any kind in those kernels.  The one exception is the compiler's 64-bit
# INTEGER division idiom (index arithmetic: item / n_work), which refines a reciprocal with v_fmac / v_fmamk against the literals +-2^32.
FUSED = re.compile(r'^(v_fma\w*|v_fmac\w*|v_fmamk\w*|v_fmaak\w*|v_pk_fma\w*|v_mac\w*|v_mad_f\w*|v_mad_mix\w*|v_fma_mix\w*|v_dot\w*|v_pk_mad\w*)\b')
INT_DIVISION_LITERALS = ('0x4f800000', '0xcf800000')
ORDER_PRESERVING = {
    'kn_csr.hip': [r'_ZN2kn'],                                  # every kernel of the file
    'kn_csr_f64.hip': [r'_ZN2kn'],
    'kn_csr_mfma.hip': [r'_ZN2kn'],
    'kn_chain.hip': [r'_ZN2kn12chain_kernel'],
    'kn_conv.hip': [r'_ZN2kn21convtaps_exact_kernel', r'_ZN2kn26convtaps_exact_pipe_kernel', r'_ZN2kn26convtaps_exact_fill_kernel', r'_ZN2kn26convtaps_zero_guard_kernel',
                    r'_ZN2kn19conv_lastrow_kernel'],
}


def _kernel_bodies(s, patterns):
    out = []
    for pat in patterns:
        for name in re.findall(r'^(%s[^\n:]*):' % pat, s, re.M):
            body = s[s.index(name + ':'):]
            body = body[:body.index('.Lfunc_end')]
            out.append((name, [l.split(';')[0].strip() for l in body.split('\n')[1:] if l.split(';')[0].strip()]))
    return out


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='needs hipcc')
@pytest.mark.parametrize('src', sorted(ORDER_PRESERVING))
def test_order_preserving_kernels_contain_no_fused_multiply_add(src, tmp_path):
    s = _isa(src, tmp_path)
    kernels = _kernel_bodies(s, ORDER_PRESERVING[src])
    assert kernels, src
    (n_mul, n_add) = (0, 0)
    for (name, lines) in kernels:
        for l in lines:
            if FUSED.match(l):
                assert any(c in l for c in INT_DIVISION_LITERALS), 'fused multiply-add in an order-preserving kernel: %s: %s' % (name, l)
            n_mul += bool(re.match(r'v_(pk_)?mul_f(32|64)\b', l)) or l.startswith('v_mfma_f32_32x32x1') or l.startswith('v_mfma_f32_16x16x1')
            n_add += bool(re.match(r'v_(pk_)?add_f(32|64)\b', l))
    assert n_mul > 0 and n_add > 0, (src, n_mul, n_add)          # (the check looked at real arithmetic: separate multiplies and adds are there)


def test_the_fused_multiply_add_lint_catches_a_contracted_build(tmp_path):
    """The same source compiled WITHOUT the pragma / flag contracts a * x + y: the lint above must see it (i.e. it is not vacuous)."""
    if shutil.which('hipcc') is None:
        pytest.skip('needs hipcc')
    src = os.path.join(str(tmp_path), 'contracted.hip')
    open(src, 'w').write('#include <hip/hip_runtime.h>\nnamespace kn { __global__ void probe(const float* a, const float* x, float* y) {'
                         ' const int i = threadIdx.x; const float p = a[i] * x[i]; y[i] = y[i] + p; } }\n')
    out = os.path.join(str(tmp_path), 'contracted.s')
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-ffp-contract=fast', '-S', '--cuda-device-only', src, '-o', out], stderr=subprocess.DEVNULL)
    lines = [l.split(';')[0].strip() for l in open(out).read().split('\n')]
    assert any(FUSED.match(l) and not any(c in l for c in INT_DIVISION_LITERALS) for l in lines)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-S', '--cuda-device-only', src, '-o', out], stderr=subprocess.DEVNULL)
    lines = [l.split(';')[0].strip() for l in open(out).read().split('\n')]
    assert not any(FUSED.match(l) for l in lines)


# A vector-ALU read of a matrix instruction's result needs passes + 2 wait states behind it (what LLVM's hazard recognizer inserts on gfx950: 18 for
# the 16-pass v_mfma_f32_32x32x1 -- `s_nop 15; s_nop 1` in the compiler-managed tails of these very kernels -- and 10 for the 8-pass
# v_mfma_f32_16x16x1; the last assertion below pins those two figures to the compiler's own output).  The compiler inserts them for its own instructions but NOT for inline asm,
# and the running sums of kn_csr_mfma.hip are inline-asm v_pk_add_f32: their distance is held by construction (two result blocks alternate, one
# s_nop behind every matrix instruction) and measured here, along every control-flow path, for every instruction that touches the result.
MFMA_WAIT = {'v_mfma_f32_32x32x1': 18, 'v_mfma_f32_16x16x1': 10}


def _wait_states(l):
    m = re.match(r's_nop (\d+)', l)
    return int(m.group(1)) + 1 if m else 1


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='needs hipcc')
@pytest.mark.parametrize('src,patterns', [('kn_csr_mfma.hip', [r'_ZN2kn21csr_group_mfma_kernel', r'_ZN2kn23csr_group_mfma16_kernel']),
                                          ('kn_conv.hip', [r'_ZN2kn26convtaps_exact_fill_kernel'])])
def test_matrix_instruction_results_are_read_after_the_required_wait_states(src, patterns, tmp_path):
    s = _isa(src, tmp_path)
    checked = 0
    worst = {}
    for (name, lines) in _kernel_bodies(s, patterns):
        labels = {l[:-1]: i for (i, l) in enumerate(lines) if l.endswith(':')}

        def first_touch(i, regs, need, seen):
            """min wait states from instruction i (exclusive of the producer) to the first instruction touching `regs`, over all paths; None = none within reach"""
            best = None
            w = 0
            while i < len(lines) and w < need + 8:
                l = lines[i]
                if l.endswith(':'):
                    i += 1
                    continue
                if l.startswith('s_endpgm'):
                    break
                ops = l.split(None, 1)[1] if ' ' in l else ''
                if (_regs(ops, 'v') & regs) and not l.startswith('s_'):
                    return w if best is None else min(best, w)
                w += _wait_states(l)
                m = re.match(r's_(c?branch\w*)\s+(\S+)', l)
                if m and m.group(2) in labels and (m.group(2), w) not in seen and len(seen) < 64:
                    seen.add((m.group(2), w))
                    t = first_touch(labels[m.group(2)], regs, need - w, seen)
                    if t is not None:
                        best = w + t if best is None else min(best, w + t)
                    if m.group(1) == 'branch':
                        return best
                i += 1
            return best

        for (i, l) in enumerate(lines):
            for (op, need) in MFMA_WAIT.items():
                if l.startswith(op):
                    dest = _regs(l.split(None, 1)[1].split(',')[0], 'v')
                    if not dest:
                        continue                          # (an AGPR destination is read back by the compiler's own v_accvgpr_read, which it spaces itself)
                    d = first_touch(i + 1, dest, need, set())
                    if d is not None:
                        assert d >= need, '%s: result of `%s` is touched after %d wait states, %d required' % (name, l, d, need)
                        worst[op] = min(worst.get(op, 1 << 30), d)
                    checked += 1
    if src == 'kn_csr_mfma.hip':
        assert checked >= 100 and worst == MFMA_WAIT, (checked, worst)      # (the closest reader anywhere is a compiler-spaced one, at exactly the compiler's figure)
    else:                                                                   # the filled-in conv kernel: 8 column ends per loop body x (1 or 2) matrix instructions x 3 instantiations
        assert checked >= 30 and set(worst) == {'v_mfma_f32_32x32x1'} and worst['v_mfma_f32_32x32x1'] >= MFMA_WAIT['v_mfma_f32_32x32x1'], (checked, worst)


# gfx950 wants a wait state between a packed f32 instruction and a vector instruction that reads its result in the very next slot (the compiler's own `s_nop 0` between two
# dependent v_pk_add_f32).  The compiler inserts it for its own instructions, NOT inside inline asm -- and the sequential thin walk of the whole-net kernel
# (kn_chain.hip: chain_rows_thin_seq) is one asm block of four adds with a multiply between each two.  Held by construction there; measured here for every packed instruction
# of the kernel (straight-line neighbours; a label or branch between two instructions is a slot of its own).
@pytest.mark.skipif(shutil.which('hipcc') is None, reason='needs hipcc')
def test_no_packed_result_is_read_in_the_next_issue_slot_of_the_whole_net_kernel(tmp_path):
    s = _isa('kn_chain.hip', tmp_path)
    kernels = _kernel_bodies(s, [r'_ZN2kn12chain_kernel'])
    assert len(kernels) == 2
    (n_pk, n_block) = (0, 0)
    for (name, lines) in kernels:
        for (a, b) in zip(lines, lines[1:]):
            if not a.startswith('v_pk_') or not b.startswith('v_'):
                continue
            n_pk += 1
            dst = _regs(a.split(None, 1)[1].split(',')[0], 'v')
            src_b = _regs(b.split(None, 1)[1].split(',', 1)[1], 'v')
            assert not (dst & src_b), 'result of a packed instruction read in the next slot: %s: %s -> %s' % (name, a, b)
        # the block itself is there: add, multiply, add, multiply ... on one accumulator
        for i in range(len(lines) - 7):
            w = lines[i:i + 8]
            if all(x.startswith('v_pk_add_f32') for x in w[0::2]) and all(x.startswith('v_pk_mul_f32') for x in w[1::2]):
                acc = {x.split(None, 1)[1].split(',')[0] for x in w[0::2]}
                n_block += len(acc) == 1
    assert n_pk > 100 and n_block >= 12, (n_pk, n_block)          # (six ring slots per kernel instance at least)
