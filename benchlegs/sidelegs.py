"""Reported-only legs behind the headline record of bench.py: each runs inside leg(), which turns a failure into an entry of res['errors'] -- the line is printed by the caller's
`finally`, so a late failure (an out-of-memory in a side leg, say) cannot lose the headline.  `c` = the bench run's context (bench.py: SimpleNamespace)."""
import json
import time

import numpy as np
import torch

from keynet_amd import sparse as ksp
from .common import log
from .layers import layer_table, time_layers, roofline_of
from .legs import end_to_end
from .parity import exact_parity, float_key_parity


def run_side_legs(c):
    (args, res, knet, sensor, x_cipher, x_plain, y_plain, n_gate, batch, world, dev, table, timed, replay) = (
        c.args, c.res, c.knet, c.sensor, c.x_cipher, c.x_plain, c.y_plain, c.n_gate, c.batch, c.world, c.dev, c.table, c.timed, c.replay)
    built_mode = None          # exact_mode(None) = back to the contract this key-net was built with (the headline's)

    def leg(name, fn):
        t0 = time.time()
        try:
            fn()
        except Exception as e:
            res['errors'][name] = '%s: %s' % (type(e).__name__, e)
            log('[bench] leg "%s" failed: %s: %s' % (name, type(e).__name__, e))
        log('[bench] leg "%s" took %.1f s' % (name, time.time() - t0))

    def leg_contract():
        rep = knet.contract_report()
        if any(r['calibration'] is not None for r in rep['layers']):
            # float-key contract (KeyedLayer._calibrate): which layers the first forward left on the matrix cores, which it moved to the
            # order-preserving kernels so that |y - y_reference| <= 1e-5 max(1, |y|) holds, and the evidence per layer
            res['contract'] = {'tolerance': 1e-5, 'layers_switched_to_exact': rep['switched'], 'rescreened_every_forward': bool(rep.get('rescreen', False)),
                               'layers': {r['name']: ({k: r['calibration'].get(k) for k in ('decided', 'bound', 'measured_mfma_vs_exact', 'tol', 'max_abs_rowsum', 'max_abs_x', 'max_abs_y')}
                                                      if r['calibration'] is not None else {'decided': 'exact' if r['exact'] else 'mfma', 'declared': True}) for r in rep['layers']}}

    def leg_slots():
        convs = [(r['name'], r['layer'].W) for r in table if isinstance(r['layer'].W, ksp.Conv2dTiledMatrix) and r['layer'].W._taps is not None]
        res['config']['slots_per_output_pixel'] = {n: {'mean': round(float(len(W._taps['ent_out'])) / (W._outshape[1] * W._outshape[2]), 3),
                                                       'max': int(np.bincount(W._taps['ent_out']).max())} for (n, W) in convs}
        res['config']['entries_carry_coefficients'] = bool(any(W._taps['ent_coef'] is not None for (n, W) in convs))

    def leg_end_to_end():
        res['end_to_end'] = end_to_end(sensor, knet, x_plain, args.steps, 1)

    def leg_graph():
        rp = knet.capture(x_cipher)
        for _ in range(3):
            rp(x_cipher)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            og = rp(x_cipher)
        torch.cuda.synchronize()
        el_g = time.perf_counter() - t0
        res['graph'] = {'images_per_s': batch * args.steps / el_g, 'ms_per_step': 1e3 * el_g / args.steps,
                        'bit_equal_to_eager': bool(torch.equal(og, knet.forward_linear(x_cipher))), 'what': 'the same forward replayed from ONE captured HIP graph (KeyedModel.capture)'}

    def leg_bf16x3():
        # EXPERIMENTAL (--experimental only; never the headline): the same key-net with the bf16x3 kernel as the first candidate of the
        # float-key contract.  Each conv layer keeps it only if its result, measured against the order-preserving kernel on the
        # calibration batch, has 4x headroom under 1e-5 * max(1, |y|).
        try:
            knet.exact_mode('auto-bf16x3')
            knet.forward_linear(x_cipher)
            torch.cuda.synchronize()
            (el_b, out_b) = timed(1, args.steps)
            err_b = float((out_b[:n_gate].contiguous().cpu() - y_plain).abs().max())
            del out_b
            table_b = time_layers(x_cipher, layer_table(knet, batch), max(1, min(args.layer_iters, 3)))
            rep_b = knet.contract_report()
            res['experimental'] = {'bf16x3': {
                'what': 'KeyedModel.exact_mode(\'auto-bf16x3\'): f32 products emulated on the bf16 matrix pipe (three-way split of both operands, six of the nine '
                        'cross products, f32 accumulate: KN_FLAG_BF16X3) in every conv layer whose calibration measured 4x headroom under the 1e-5 tolerance',
                'dtype': 'f32 emulated (3 x bf16 split, 6 of 9 cross products, f32 accumulate)',
                'images_per_s': batch * args.steps / el_b, 'ms_per_step': 1e3 * el_b / args.steps, 'steps': args.steps,
                'parity': {'vs_source_network_max_abs_err': err_b, 'ok': bool(err_b <= 1e-3),
                           'per_layer_vs_order_preserving_kernel': {r['name']: {k: r['calibration'].get(k) for k in ('decided', 'measured_bf16x3_vs_exact', 'measured_mfma_vs_exact', 'tol')}
                                                                    for r in rep_b['layers'] if r['calibration'] is not None and 'tol' in r['calibration']}},
                'layers_on_bf16x3': [r['name'] for r in rep_b['layers'] if r['exact'] == 'bf16x3'],
                'roofline': roofline_of(table_b, args.workload, batch, 'tolerance'),
                'layers_ms': {r['name']: round(r['ms'], 4) for r in table_b}}}
        finally:
            knet.exact_mode(built_mode)

    def leg_exact():
        # the same key-net under the bit-exact contract (north_star: "bit-exact for the permutation-only key"; the DEFAULT of a
        # permutation-only tiled key-net -- the headline above opted into the matrix cores explicitly, config.mode says so)
        try:
            knet.exact_mode(True)
            t0 = time.time()
            knet.forward_linear(x_cipher)                    # uploads the CSR twins of fc6-8
            torch.cuda.synchronize()
            log('[bench exact] exact-mode operators resident + first forward in %.1f s' % (time.time() - t0))
            (el_x, out_x) = timed(1, args.steps)
            err_x = float((out_x[:n_gate].contiguous().cpu() - y_plain).abs().max())
            del out_x
            table_x = time_layers(x_cipher, layer_table(knet, batch), max(1, min(args.layer_iters, 3)))
            for r in table_x:
                log('[bench exact] %-8s %-9s %8.3f ms  %7.2f T MAC/s' % (r['name'], r['kind'], r['ms'], r['nnz'] * batch / r['ms'] / 1e9))
            par_x = exact_parity(knet, x_cipher)
            par_x['vs_source_network_max_abs_err'] = err_x
            par_x['ok'] = bool(par_x['ok'] and err_x <= 1e-3)
            par_x['note'] = 'this leg samples conv1_2 and conv4_2; all 21 layers are checked the same way by tests/test_vgg16_full_gpu.py'
            res['exact'] = {'images_per_s': batch * args.steps / el_x, 'ms_per_step': 1e3 * el_x / args.steps, 'steps': args.steps,
                            'mode': 'KeyedModel.exact_mode(True): every layer in the reference\'s accumulation order and rounding (no FMA, no MFMA)',
                            'roofline': roofline_of(table_x, args.workload, batch, 'exact'), 'parity': par_x,
                            'layers_ms': {r['name']: round(r['ms'], 4) for r in table_x}}
            if not par_x['ok']:
                raise AssertionError('exact-mode parity failed: %s' % json.dumps(par_x))
        finally:
            knet.exact_mode(built_mode)

    def leg_float_key_parity():
        res['float_key_parity'] = float_key_parity(dev)

    def leg_exact_layers():
        # float-key workloads: the layers the contract keeps in the reference's order, AS TIMED (whole batch, the key-net's own activations as input), against the CPU oracle
        # on the canonical CSR of sampled output pixels (Conv2dTiledMatrix.rows_csr: a pixel pair hit by several taps is one stored entry, its terms summed in entry order)
        names = tuple(r['name'] for r in knet.contract_report()['layers'] if r['exact'] is True and r['calibration'] is not None and r['calibration'].get('decided') == 'exact')
        if names:
            res['exact_layers_parity'] = exact_parity(knet, x_cipher, n_img=8, n_pix=1, layers=names)


    single = world == 1 and replay is None
    leg('contract', leg_contract)
    if args.workload.startswith('vgg16'):
        leg('slots', leg_slots)
    if single and x_plain is not None:
        leg('end_to_end', leg_end_to_end)
    if args.graph_leg and single:
        leg('graph', leg_graph)
    if args.workload == 'vgg16' and single and not args.exact and not args.no_exact_leg:
        leg('exact', leg_exact)
        leg('float_key_parity', leg_float_key_parity)
    if args.exact_layers_parity and args.workload.startswith('vgg16-') and single and not args.exact:
        leg('exact_layers_parity', leg_exact_layers)
    if args.experimental and args.workload.startswith('vgg16') and single and not args.exact:
        leg('experimental_bf16x3', leg_bf16x3)

