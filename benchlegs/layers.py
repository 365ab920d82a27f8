"""Per-layer tables of a key-net in its current mode (kernel family, algorithmic work per SURVEY 8d, HIP-event timings) and the roofline record built from them."""
import json
import os

import numpy as np
import torch

from keynet_amd import sparse as ksp
from keynet_amd.layer import KeyedLayer
from .common import ROOT, PEAK_F32_MFMA_TFLOPS, PEAK_VALU_NOFMA_TMACS, PEAK_HBM_GBS, PEAK_L2_READ_GBS_MEASURED, kernel_sources_sha


def _takes_small_k_kernel(W, batch):
    """Mirror of the dispatch in kn_conv.hip (convtaps_spmm): one output pixel's whole contraction fits 28 rows (VGG conv1_1),
    which runs in the write-bound convtaps_smallk_kernel and is therefore not part of the MFMA roofline aggregate."""
    t = getattr(W, '_taps', None)
    if t is None or batch % 256:
        return False
    return int(np.bincount(t['ent_out']).max()) * W._inshape[0] + (1 if t['lastcol'] is not None else 0) <= 28


def layer_table(knet, batch):
    """Per keyed layer: the kernel family that runs it in the key-net's CURRENT mode, algorithmic MACs (= nnz of the expanded
    operator the reference applies) and bytes."""
    rows = []
    children = list(knet._keynet.named_children())
    for (i, (name, c)) in enumerate(children):
        if not isinstance(c, KeyedLayer):
            continue
        contract = getattr(c, '_exact', True)
        exact = contract is True or contract == 'auto'
        if type(c.W) is ksp.SparseMatrix and not exact and c.W._dense_device_op() is not None:
            kind = 'dense'                     # keyed nn.Linear on the split-K MFMA path (tolerance mode)
            (r, cdim) = c.W.shape
            nnz_exp = int(c.W.nnz())
            wbytes = 4 * r * cdim
        elif isinstance(c.W, ksp.Conv2dTiledMatrix):
            op = c.W._device_op()
            (r, cdim) = op.shape()
            nnz_exp = op.nnz_expanded()
            kind = 'convexact' if exact else ('convsplit' if contract == 'split' else ('smallk' if _takes_small_k_kernel(c.W, batch) else 'convtaps'))
            wbytes = 4 * c.W.nnz()             # taps + entries + last column actually read
        elif isinstance(c.W, ksp.FactoredSparseMatrix):
            # an untiled keyed conv whose stored CSR is provably the expansion of its factored form: runs the order-preserving conv pipeline from
            # the taps (sparse.py: FactoredSparseMatrix); algorithmic MACs = the stored non-zeros of the reference's CSR
            op = c.W._device_op()
            (r, cdim) = op.shape()
            nnz_exp = int(c.W.nnz())
            kind = 'convexact'
            wbytes = 4 * c.W._factored.nnz()   # taps + entries + last column actually read
        else:
            op = c.W._device_op()
            (r, cdim) = op.shape()
            nnz_exp = op.nnz_expanded()
            kind = 'csr'
            wbytes = 8 * nnz_exp               # (col,val) per non-zero
        flags = (1 if ((i + 1 < len(children)) and isinstance(children[i + 1][1], torch.nn.ReLU)) or c.iskeyedrelu() else 0) | (2 if (exact and kind != 'dense') else 0) | (4 if contract == 'bf16x3' else 0)
        plan = (c.W._dense_device_op() if kind == 'dense' else op).plan(batch, flags)
        if 'bf16x3' in plan:
            kind = 'convbf16x3'
        flops_exec = None
        if kind == 'convsplit':                # the split application (Conv2dTiledMatrix._split_ops): spatial CSR per input channel, then an ntaps-slot conv-taps operator
            (opK, op2) = c.W._split_ops()
            t = c.W._taps
            plan = '%d x [%s]; %s' % (c.W._inshape[0], opK.plan(batch, 2), op2.plan(batch, flags & 1))
            flops_exec = 2.0 * batch * (len(t['ent_out']) * c.W._inshape[0] + len(t['taps']) * c.W._outshape[1] * c.W._outshape[2] * c.W._inshape[0] * c.W._outshape[0])
        rows.append(dict(name=name, kind=kind, rows=r, cols=cdim, nnz=nnz_exp, flops=2.0 * nnz_exp * batch, flops_executed=flops_exec,
                         bytes=float(wbytes) + 4.0 * batch * (r + cdim), layer=c, plan=plan,
                         fuse=(i + 1 < len(children)) and isinstance(children[i + 1][1], torch.nn.ReLU)))
    return rows


def time_layers(x_cipher, table, iters, reps=3):
    """Per-layer kernel time with HIP events on the launch stream (torch's current stream is the one kn_spmm launches on).
    In the forward the kernels run back to back, so each timing is over `reps` back-to-back launches behind one untimed launch:
    a launch that follows an idle period runs 1-20 % slower while the clock ramps back up (measured with per-workgroup time
    stamps, profiles/r02_workgroup_timeline_conv_layers.txt), which is not what happens inside the timed step.  The MEDIAN over
    `iters` such timings is kept: a multi-GB output allocation can occasionally fall out of the caching allocator and cost tens
    of ms, which must not leak into a kernel's average."""
    y = x_cipher
    for row in table:
        c = row['layer']
        xin = y
        y = None
        out = c.forward(xin, fuse_relu=row['fuse'])       # warm
        torch.cuda.synchronize()
        times = []
        for _ in range(max(iters, 1)):
            del out
            out = c.forward(xin, fuse_relu=row['fuse'])
            (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            e0.record()
            for _ in range(reps):
                del out
                out = c.forward(xin, fuse_relu=row['fuse'])
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / reps)
        row['ms'] = float(np.median(times))
        row['ms_min'] = float(np.min(times))
        y = out
    return table


def committed_traffic(workload, mode):
    """HBM bytes per forward of the dominant kernel from the committed PMC passes (profiles/rNN_<workload>_*traffic.json:
    separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same bench, FETCH doubled per the guide's gfx950
    note).  bench.py cannot collect PMC counters on itself, so the figure is quoted only when that pass was taken in the same
    mode on byte-identical kernel sources (`csrc_sha256` recorded by tools/make_profiles.py); otherwise (None, reason)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_%s_*traffic.json' % workload)))
    if not files:
        return (None, 'no committed PMC pass', None)
    t = json.load(open(files[-1]))
    rel = os.path.relpath(files[-1], ROOT)
    if t.get('mode', 'tolerance') != mode:
        return (None, '%s is a %s-mode pass' % (rel, t.get('mode', 'tolerance')), None)
    if t.get('csrc_sha256') != kernel_sources_sha():
        return (None, '%s was taken on other kernel sources' % rel, None)
    # (the pass counts EVERY launch of the dominant kernel in one forward -- for the conv-taps kernel also the split-K launches of fc6-8 and both half-batch windows of a layer: the
    #  ratio is priced against the algorithmic bytes of exactly those launches, which the source file names)
    return (t.get('convtaps_hbm_bytes_per_forward', t.get('dominant_hbm_bytes_per_forward')),
            '%s (%s launches of one marked forward, %.4g algorithmic bytes)' % (rel, t.get('dominant_launches', t.get('launches_per_forward', '?')), t.get('algorithmic_bytes_dominant') or float('nan')),
            t.get('traffic_ratio'))


def roofline_of(table, workload, batch, mode):
    """Roofline record of the dominant kernel family of `table` (the layers as they ran in this mode)."""
    kinds = {}
    for r in table:
        kinds.setdefault(r['kind'], []).append(r)
    by_ms = sorted(kinds.items(), key=lambda kv: -sum(r['ms'] for r in kv[1]))
    (kind, dom) = by_ms[0]
    dom_ms = sum(r['ms'] for r in dom)
    if kind == 'convbf16x3':
        peak = PEAK_F32_MFMA_TFLOPS * 16.0 / 6.0
        ach = sum(r['flops'] for r in dom) / dom_ms / 1e9
        return dict(bound='mfma', kernel='convtaps_bf16x3_kernel (%d launches/forward)' % len(dom), achieved=ach, peak=peak, unit='TFLOP/s (f32-equivalent)', frac=ach / peak, traffic=None,
                    algorithmic_flops=sum(r['flops'] for r in dom), algorithmic_bytes=sum(r['bytes'] for r in dom), ms_per_forward=dom_ms,
                    note='peak = six v_mfma_f32_32x32x16_bf16 per f32 product block at 16x the f32-input MFMA rate: 157.3 * 16 / 6')
    if kind == 'convsplit':
        ach = sum(r['flops_executed'] for r in dom) / dom_ms / 1e9
        return dict(bound='mfma', kernel='split application of filled-in conv layers: spatial CSR kernels per input channel + convtaps_mfma_kernel (%d layers/forward)' % len(dom), achieved=ach,
                    peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s (executed)', frac=ach / PEAK_F32_MFMA_TFLOPS, traffic=None, algorithmic_flops=sum(r['flops'] for r in dom),
                    executed_flops=sum(r['flops_executed'] for r in dom), algorithmic_bytes=sum(r['bytes'] for r in dom), ms_per_forward=dom_ms,
                    note='algorithmic_flops = the stored entries of the fused operator the reference applies; the split application executes executed_flops for the same product')
    if kind in ('convtaps', 'dense'):
        dom = kinds.get('convtaps', []) or dom
        dom_ms = sum(r['ms'] for r in dom)
        ach = sum(r['flops'] for r in dom) / dom_ms / 1e9
        (traffic, tsrc, tratio) = committed_traffic(workload, mode) if batch == 256 else (None, 'PMC pass is for 256 images', None)
        return dict(bound='mfma', kernel='convtaps_mfma_kernel (%d launches/forward)' % len(dom), achieved=ach, peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s',
                    frac=ach / PEAK_F32_MFMA_TFLOPS, traffic=traffic, traffic_unit='bytes/forward (PMC, offline pass over ONE marked forward: tools/pmc_forward.py)', traffic_source=tsrc,
                    traffic_ratio=tratio,        # traffic / the algorithmic bytes of the launches the pass counted (the conv layers + the split-K launches of fc6-8)
                    algorithmic_bytes=sum(r['bytes'] for r in dom), algorithmic_flops=sum(r['flops'] for r in dom), ms_per_forward=dom_ms)
    macs = sum(r['nnz'] for r in dom) * float(batch)
    intensity = 2.0 * macs / sum(r['bytes'] for r in dom)
    mains = []                                            # the first kernel of each layer's plan (kn_spmm_plan lists the main launch first, guards / last-row helpers behind it)
    for r in dom:
        k = [w.split('<')[0] for w in str(r.get('plan', '')).replace(',', ' ').split() if w.split('<')[0].endswith('_kernel')]
        if k and k[0] not in mains:
            mains.append(k[0])
    names = ' / '.join(mains) or \
        {'convexact': 'convtaps_exact_pipe_kernel / convtaps_exact_kernel', 'csr': 'csr_group_kernel / csr_rows_kernel', 'smallk': 'convtaps_smallk_kernel'}[kind]
    if kind == 'smallk' or intensity < 2.0 * PEAK_VALU_NOFMA_TMACS * 1e3 / PEAK_HBM_GBS:      # below the balance point of the no-FMA VALU roof: HBM-bound
        ach = sum(r['bytes'] for r in dom) / dom_ms / 1e6
        return dict(bound='hbm', kernel='%s (%d launches/forward)' % (names, len(dom)), achieved=ach, peak=PEAK_HBM_GBS, unit='GB/s', frac=ach / PEAK_HBM_GBS,
                    traffic=None, algorithmic_bytes=sum(r['bytes'] for r in dom), ms_per_forward=dom_ms)
    ach = macs / dom_ms / 1e9
    (traffic, tsrc, tratio) = committed_traffic(workload, mode)
    return dict(bound='valu-nofma', kernel='%s (%d launches/forward)' % (names, len(dom)), achieved=ach, peak=PEAK_VALU_NOFMA_TMACS, unit='T MAC/s',
                frac=ach / PEAK_VALU_NOFMA_TMACS, traffic=traffic, traffic_ratio=tratio, traffic_unit='bytes/forward (PMC, offline pass over ONE marked forward: tools/pmc_forward.py)', traffic_source=tsrc, algorithmic_macs=macs,
                algorithmic_bytes=sum(r['bytes'] for r in dom), ms_per_forward=dom_ms,
                note='bit-exact contract: a separately rounded f32 product and an f32 add per stored value, in the reference\'s order -- no fused multiply-add, no '
                     'accumulating matrix instruction; roof = one product + one add per lane per 2 cycles = 157.3 TFLOP/s / 4 (kernels that take their products '
                     'from K = 1 matrix instructions with a zero accumulator still pay the adds on the same lanes: DESIGN.md section 8)')


def chain_roofline(knet, chain, x_cipher, table, batch):
    """Roofline record of a key-net whose forward is ONE launch of the whole-net kernel (csrc/kn_chain.hip).  Algorithmic bytes: every operator once (8 B per stored
    non-zero) + activations in and out of every layer; next to it what the kernel really streams -- the operator words once PER WORKGROUP, from L2."""
    import re
    total_bytes = sum(r['bytes'] for r in table)
    # the forward of this key-net is ONE launch of the whole-net kernel (csrc/kn_chain.hip): that launch is the dominant kernel.
    # Algorithmic bytes: every operator once (8 B per stored non-zero) + activations in and out of every layer.
    # (a launch is 37 us: 2 000 untimed launches = 75 ms bring the GPU off its idle clock, as in the timed loop; 5 + 50 launches -- 2 ms -- read 41 us)
    for _ in range(2000):
        knet.forward_linear(x_cipher)
    (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    torch.cuda.synchronize()
    e0.record()
    for _ in range(500):
        knet.forward_linear(x_cipher)
    e1.record()
    torch.cuda.synchronize()
    ch_ms = e0.elapsed_time(e1) / 500
    ach = total_bytes / ch_ms / 1e6
    # What really bounds it: bit-exactness with scipy forbids the FMA, so a stored non-zero costs one packed multiply and one
    # packed add per two batch columns, 4 cycles each on one of the CU's four SIMDs; a workgroup owns 4 columns, 256 CUs run a round.
    nnz_net = float(sum(r['nnz'] for r in table))
    rounds = -(-((batch + 3) // 4) // 256)
    valu_floor_ms = rounds * (nnz_net / 64.0) * 4 * 4 / 4 / 2.4e9 * 1e3
    # ... and what it really streams (round 6): every workgroup (4 batch columns) reads the key-net's operator words from L2 once -- n_workgroups x the bytes the plan
    # names -- against the L2 read rate all 256 CUs reach together on an L2-resident array (tools/micro/l2_read_rate.hip, profiles/r06_micro_l2_read_rate.txt)
    m_l2 = re.search(r'(\d+) B of operator words per workgroup', chain.plan(batch))
    l2_bytes = float(m_l2.group(1)) * ((batch + 3) // 4) if m_l2 else None
    roof = dict(bound='hbm', kernel=chain.plan(batch), achieved=ach, peak=PEAK_HBM_GBS, unit='GB/s', frac=ach / PEAK_HBM_GBS, traffic=None,
                l2_stream_bytes_per_forward=l2_bytes, l2_read_peak_gbs_measured=PEAK_L2_READ_GBS_MEASURED,
                frac_of_l2_read_roof=(l2_bytes / ch_ms / 1e6 / PEAK_L2_READ_GBS_MEASURED) if l2_bytes else None,
                valu_floor_ms=valu_floor_ms, frac_of_valu_floor=valu_floor_ms / ch_ms,
                algorithmic_bytes=total_bytes, algorithmic_macs=float(sum(r['nnz'] for r in table)) * batch, ms_per_forward=ch_ms,
                t_mac_per_s=float(sum(r['nnz'] for r in table)) * batch / ch_ms / 1e9,
                launch_per_layer_ms={r['name']: round(r['ms'], 4) for r in table},
                note='one launch for the whole key-net, activations in LDS; launch_per_layer_ms = the seven separate kernels it replaces (KN_NO_CHAIN=1)')
    return roof
