"""The driver keeps a bounded tail of bench.py's stdout: the ONE JSON line must stay under 4 KB whatever the legs produced
(round 3's line had grown to 27 KB and was cut: BENCH_r03.parsed = null).  CPU only: the records are canned."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402

REQUIRED = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def canned():
    """Round 3's full 27 KB record (profiles/r03_vgg16_b256_bench.json) -- the very line the driver could not parse."""
    return json.load(open(os.path.join(ROOT, 'profiles', 'r03_vgg16_b256_bench.json')))


def check(line):
    assert '\n' not in line
    assert len(line) < 4096, len(line)
    r = json.loads(line)
    for k in REQUIRED:
        assert k in r, k
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(r['roofline'])
    assert r['config']['workload'] and 'model' not in r['config']
    return r


def test_full_round3_record_fits_the_line():
    res = canned()
    assert len(json.dumps(res)) > 20000
    r = check(bench.compact_record(res))
    assert set(('value', 'unit', 'cores', 'kind', 'sample')) <= set(r['cpu_baseline'])
    assert len(r['cpu_baseline']['sample']) <= 200
    assert r['exact']['bit_equal'] is True and r['exact']['oracle_checked_layers'] == ['conv1_2', 'conv4_2']
    assert set(r['secondary']) == {'lenet', 'allconv'} and r['secondary']['lenet']['bit_equal'] is True
    assert r['detail'] == bench.DETAIL_FILE
    assert abs(r['value'] - res['value']) / res['value'] < 1e-5


@pytest.mark.parametrize('world', [2, 8])
def test_multi_gpu_record_fits_the_line(world):
    res = canned()
    res['n_gpus'] = world
    res['cpu_baseline'] = 'measured at N=1 only (the scipy baseline runs on rank 0 of a single-GPU run; see BENCH / profiles)'
    for k in ('secondary', 'exact', 'experimental', 'float_key_parity', 'end_to_end'):
        res.pop(k, None)
    res['collective'] = {'backend': 'nccl', 'ranks_seen': world,
                         'ranks': [{'rank': k, 'local_rank': k, 'device_index': k, 'device_name': 'AMD Instinct MI355X', 'pid': 100000 + k} for k in range(world)],
                         'op': 'all_gather_into_tensor of [256, 2622] f32 logits per rank', 'bytes_per_rank': 2684928, 'ms_per_call': 0.0731234, 'ms_per_call_wall': 0.08, 'calls_timed': 20,
                         'every_rank_shard_bit_equal_to_its_local_forward': True, 'rank0_shard_bit_equal': True, 'rank0_shard_sha256': 'ab' * 32,
                         'rank_images_per_s': {'min': 4400.123456789, 'max': 4512.3, 'all': [4400.123456789] * world, 'what': 'x' * 300},
                         'startup': {'keying_or_loading_s_rank0': 26.0},
                         'peer_shard_recomputed_on_rank0': {'peer_rank': world - 1, 'bit_equal': True}}
    r = check(bench.compact_record(res))
    c = r['collective']
    assert c['ranks_seen'] == world and c['ranks'] == [[k, k] for k in range(world)]
    assert c['backend'] == 'nccl' and c['bytes_per_rank'] == 2684928 and abs(c['gather_ms'] - 0.0731234) < 1e-6 and c['rank_images_per_s'] == {'min': 4400.12, 'max': 4512.3}
    assert c['peer_shard_recomputed_on_rank0'] == {'peer_rank': world - 1, 'bit_equal': True}
    assert 'pid' not in json.dumps(c) and 'device_name' not in json.dumps(c)


def test_pathological_strings_and_failed_legs_still_fit():
    res = canned()
    res['config']['workload'] = 'w' * 5000
    res['roofline']['kernel'] = 'k' * 5000
    res['cpu_baseline']['sample'] = 's' * 50000
    res['exact'] = {'error': 'e' * 9000}
    res['secondary'] = {'lenet': {'error': 'child exited with 1', 'stderr_tail': 'x' * 800}, 'allconv': copy.deepcopy(res['secondary']['allconv'])}
    res['errors'] = {'leg%d' % i: 'boom ' * 400 for i in range(12)}
    r = check(bench.compact_record(res))
    assert 'error' in r['exact'] and 'error' in r['secondary']['lenet']


def test_non_finite_numbers_do_not_break_json():
    res = canned()
    res['roofline']['traffic'] = float('nan')
    r = check(bench.compact_record(res))
    assert r['roofline']['traffic'] is None


def test_experimental_leg_is_reported_beside_the_headline_never_as_it():
    """`bench.py --experimental` (round-5 review, Next #7): the bf16x3 leg appears as experimental.bf16x3{images_per_s, tf_equiv, frac_of_bf16_roof, worst_gate_ratio,
    layers_on}; `value` and `dtype` stay the f32 path's.  The record is a real run's (profiles/r06_vgg16_experimental_bf16x3_detail.json)."""
    res = canned()
    real = json.load(open(os.path.join(ROOT, 'profiles', 'r06_vgg16_experimental_bf16x3_detail.json')))
    res['experimental'] = real['experimental']
    r = check(bench.compact_record(res))
    e = r['experimental']['bf16x3']
    assert set(e) >= {'images_per_s', 'tf_equiv', 'frac_of_bf16_roof', 'worst_gate_ratio', 'layers_on'}
    assert e['layers_on'] == 12 and 0 < e['worst_gate_ratio'] < 0.25 and 0.3 < e['frac_of_bf16_roof'] < 1
    assert r['dtype'] == 'f32' and abs(r['value'] - res['value']) / res['value'] < 1e-5 and e['images_per_s'] != r['value']
