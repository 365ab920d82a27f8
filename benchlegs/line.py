"""The ONE stdout line: compact (< 4 KB) so that a driver keeping a bounded stdout tail always sees the whole record; everything else (per-layer tables, plans,
contract evidence, child lines, experimental legs) goes to bench_detail.json and to stderr."""
import json
import os

from .common import ROOT, log, _num, _pick, _clip

LINE_LIMIT = 4096
DETAIL_FILE = 'bench_detail.json'


def _compact_roofline(r):
    if not isinstance(r, dict):
        return r
    out = _pick(r, ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'frac_of_l2_read_roof', 'traffic', 'traffic_ratio', 'algorithmic_bytes', 'algorithmic_flops', 'algorithmic_macs', 'ms_per_forward'))
    out['kernel'] = _clip(out.get('kernel'), 96)
    return out


def _compact_cpu(c):
    if not isinstance(c, dict):
        return c
    out = _pick(c, ('value', 'unit', 'cores', 'kind', 'engine', 'host'))
    out['engine'] = _clip(out.get('engine'), 72)
    if isinstance(c.get('all_cores'), dict):
        out['all_cores'] = _pick(c['all_cores'], ('value', 'cores'))
    out['sample'] = _clip(c.get('sample_short') or c.get('sample'), 200)
    return out


def _compact_secondary(s):
    if not isinstance(s, dict):
        return s
    if 'error' in s:
        return {'error': _clip(str(s['error']), 120)}
    roof = s.get('roofline') or {}
    par = s.get('parity') or {}
    cpu = s.get('cpu_baseline') or {}
    return {'images_per_gpu': s.get('images_per_gpu'), 'images_per_s': s.get('images_per_s'), 'ms_per_step': s.get('ms_per_step'), 'bound': roof.get('bound'), 'achieved': roof.get('achieved'),
            'peak': roof.get('peak'), 'unit': roof.get('unit'), 'frac': roof.get('frac'), 'frac_of_l2_read_roof': roof.get('frac_of_l2_read_roof'), 'kernel_ms': roof.get('ms_per_forward'),
            'bit_equal': par.get('bit_equal'),
            'cpu_images_per_s': cpu.get('value'), 'cpu_cores': cpu.get('cores')}


def _compact_collective(c):
    """What whoever runs the N > 1 tiers needs to validate the line on its own: backend, ranks really seen (rank, device index), the gathered message, the all-gather's
    own time, the slowest and fastest rank, and the two bit-level checks of the gathered block."""
    if not isinstance(c, dict):
        return c
    out = _pick(c, ('backend', 'ranks_seen', 'bytes_per_rank', 'every_rank_shard_bit_equal_to_its_local_forward'))
    out['gather_ms'] = c.get('ms_per_call')
    if isinstance(c.get('rank_images_per_s'), dict):
        out['rank_images_per_s'] = _pick(c['rank_images_per_s'], ('min', 'max'))
    out['ranks'] = [[r.get('rank'), r.get('device_index')] for r in (c.get('ranks') or []) if isinstance(r, dict)]      # [rank, device_index] per rank
    out['peer_shard_recomputed_on_rank0'] = c.get('peer_shard_recomputed_on_rank0')
    return out


def compact_record(res, detail_path=DETAIL_FILE):
    """The driver's line from the full record `res` (which is written to `detail_path`).  Keys and order follow the bench contract; every
    nested object is cut to the fields a reader needs to check the number (the rest is in the detail file, whose path the line carries)."""
    cfg = res.get('config') or {}
    line = {k: res.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')}
    line['metric'] = _clip(line['metric'], 120)
    line['config'] = _pick(cfg, ('workload', 'mode', 'headline_contract', 'default_contract_value', 'images_per_gpu', 'global_batch', 'nnz_per_image', 'parallelism'))
    line['config']['workload'] = _clip(line['config'].get('workload'), 160)
    line['config']['mode'] = _clip(line['config'].get('mode'), 160)
    line['roofline'] = _compact_roofline(res.get('roofline'))
    line['cpu_baseline'] = _compact_cpu(res.get('cpu_baseline')) if isinstance(res.get('cpu_baseline'), dict) else _clip(res.get('cpu_baseline'), 120)
    par = res.get('parity') or {}
    line['parity'] = _pick(par, ('ok', 'max_abs_err', 'atol'))
    if isinstance(res.get('oracle_parity'), dict):
        line['parity']['oracle_bit_equal'] = res['oracle_parity'].get('bit_equal')
    ex = res.get('exact')
    if isinstance(ex, dict):
        if 'error' in ex:
            line['exact'] = {'error': _clip(str(ex['error']), 120)}
        else:
            roof = ex.get('roofline') or {}
            epar = ex.get('parity') or {}
            line['exact'] = {'images_per_s': ex.get('images_per_s'), 'ms_per_step': ex.get('ms_per_step'), 'frac': roof.get('frac'), 'peak': roof.get('peak'), 'unit': roof.get('unit'),
                             'bit_equal': epar.get('ok'), 'oracle_checked_layers': [r.get('layer') for r in (epar.get('layers') or [])]}
    if isinstance(res.get('secondary'), dict):
        line['secondary'] = {k: _compact_secondary(v) for (k, v) in res['secondary'].items()}
    if isinstance(res.get('contract'), dict):
        line['contract'] = {'tolerance': res['contract'].get('tolerance'), 'layers_switched_to_exact': res['contract'].get('layers_switched_to_exact'),
                            'rescreened_every_forward': res['contract'].get('rescreened_every_forward')}
    if isinstance(res.get('end_to_end'), dict):
        line['end_to_end'] = _pick(res['end_to_end'], ('images_per_s', 'ms_per_step', 'encrypt_ms', 'error'))
    if isinstance(res.get('exact_layers_parity'), dict):
        line['exact_layers_parity'] = {'bit_equal': res['exact_layers_parity'].get('ok'), 'oracle_checked_layers': [r.get('layer') for r in (res['exact_layers_parity'].get('layers') or [])]}
    if isinstance((res.get('experimental') or {}).get('bf16x3'), dict):
        # --experimental only, never the headline (`value` and `dtype` above are the f32 path's): f32 products emulated on the bf16 matrix pipe, per layer inside the same gate
        e = res['experimental']['bf16x3']
        roof = e.get('roofline') or {}
        per = (e.get('parity') or {}).get('per_layer_vs_order_preserving_kernel') or {}
        ratios = [v['measured_bf16x3_vs_exact'] / v['tol'] for v in per.values() if v.get('decided') == 'bf16x3' and v.get('measured_bf16x3_vs_exact') is not None and v.get('tol')]
        line['experimental'] = {'bf16x3': {'images_per_s': e.get('images_per_s'), 'tf_equiv': roof.get('achieved'), 'frac_of_bf16_roof': roof.get('frac'),
                                           'worst_gate_ratio': max(ratios) if ratios else None, 'layers_on': len(e.get('layers_on_bf16x3') or []),
                                           'vs_source_network_max_abs_err': (e.get('parity') or {}).get('vs_source_network_max_abs_err')}}
    if res.get('collective') is not None:
        line['collective'] = _compact_collective(res['collective'])
    if res.get('errors'):
        line['errors'] = {k: _clip(str(v), 100) for (k, v) in list(res['errors'].items())[:6]}
    line['detail'] = detail_path
    line = _num(line)
    s = json.dumps(line, separators=(',', ':'))
    # belt and braces: if a pathological string still pushes the line over the limit, drop optional sections until it fits
    for k in ('experimental', 'end_to_end', 'exact_layers_parity', 'contract', 'secondary', 'exact', 'errors'):
        if len(s) < LINE_LIMIT:
            break
        line.pop(k, None)
        s = json.dumps(line, separators=(',', ':'))
    assert len(s) < LINE_LIMIT, 'bench line is %d chars' % len(s)
    return s


def write_detail(res, path=None):
    """Full record next to bench.py (and under gpurun_out/ when that scratch directory exists, so that a GPU-box run brings it home)."""
    paths = [path or os.path.join(ROOT, DETAIL_FILE)]
    if path is None and os.path.isdir(os.path.join(ROOT, 'gpurun_out')):
        paths.append(os.path.join(ROOT, 'gpurun_out', DETAIL_FILE))
    for p in paths:
        try:
            with open(p, 'w') as f:
                json.dump(res, f, indent=1, default=str)
        except OSError as e:
            log('[bench] could not write %s: %s' % (p, e))
