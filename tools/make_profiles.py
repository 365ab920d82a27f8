#!/usr/bin/env python3
"""Condense a gpurun_out/<run>/ directory (bench.json/.log, stats/, pmc_fetch/, pmc_write/, pmc_mfma/, pmc_l2/ written by the
rocprofv3 commands listed in profiles/README.md) into the committed profiles/rNN_* artefacts.

    python3 tools/make_profiles.py gpurun_out/r1d r01
"""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

csv.field_size_limit(1 << 30)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    """The most recent match: gpurun merges a run's output INTO gpurun_out/, so a directory used twice holds files of both runs (other pids in the names)."""
    return max(glob.glob(pattern), key=os.path.getmtime)


def kernel_sources_sha():
    """sha256 over keynet_amd/csrc/*.{hip,h} (same function as bench.py): ties a PMC pass to the kernel sources it was taken on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'keynet_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()


def per_kernel(d, counter):
    f = newest(d + '/runc/*counter_collection.csv')
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return agg


def main(R, tag, out='profiles', forwards=None):
    os.makedirs(out, exist_ok=True)
    pre = '%s/%s_vgg16_b256_' % (out, tag)
    for name in ('bench.json', 'bench_under_rocprof.json', 'kernel_stats.csv', 'layers.log', 'per_layer_pmc.csv', 'per_layer_trace.csv', 'traffic.json'):     # only what this script writes
        if os.path.exists(pre + name):
            os.remove(pre + name)
    rows = list(csv.DictReader(open(newest(R + '/stats/runc/*kernel_stats.csv'))))
    with open(pre + 'kernel_stats.csv', 'w') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
        for r in rows:
            n = r['Name']
            w.writerow([n if len(n) < 150 else n[:147] + '...', r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
    shutil.copy(R + '/bench.json', pre + 'bench.json')
    if os.path.exists(R + '/bench_detail.json'):
        shutil.copy(R + '/bench_detail.json', pre + 'bench_detail.json')
    shutil.copy(R + '/stats_bench.json', pre + 'bench_under_rocprof.json')
    open(pre + 'layers.log', 'w').write(''.join(l for l in open(R + '/bench.log') if 'bench' in l))
    # roofline.traffic: the launches of ONE marked forward (tools/pmc_forward.py; bench.py --pmc-forward), nothing estimated
    tr = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, 'tools', 'pmc_forward.py'), R + '/pmc_fetch', R + '/pmc_write', R + '/forward.json']))
    assert tr['csrc_sha256'] == kernel_sources_sha(), 'the PMC passes were taken on other kernel sources than this tree'
    (tot_f, tot_w) = (tr['fetch_x2'], tr['write'])
    pm = {}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(newest(R + '/pmc_mfma/runc/*counter_collection.csv'))):
        if 'convtaps' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:70]][r['Counter_Name']] += float(r['Counter_Value'])
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(newest(R + '/pmc_mfma/runc/*kernel_trace.csv'))):
        if 'convtaps' in r['Kernel_Name']:
            dur[r['Kernel_Name'][:70]] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
    for (k, v) in agg.items():
        pm[k] = {'effective_clock_ghz': v['GRBM_GUI_ACTIVE'] / 8 / dur[k] / 1e9, 'mfma_busy_fraction_of_simd_cycles': v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * v['GRBM_GUI_ACTIVE'] / 8),
                 'lds_bank_conflict_cycles': v['SQ_LDS_BANK_CONFLICT'], 'duration_s_under_pmc': dur[k]}
    agg2 = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(newest(R + '/pmc_l2/runc/*counter_collection.csv'))):
        if 'convtaps' in r['Kernel_Name']:
            agg2[r['Kernel_Name'][:70]][r['Counter_Name']] += float(r['Counter_Value'])
    for (k, v) in agg2.items():
        pm.setdefault(k, {})['l2_hit_rate'] = v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum'])
    tr['convtaps_pmc'] = pm
    json.dump(tr, open(pre + 'traffic.json', 'w'), indent=1)
    if os.path.exists(R + '/per_layer.csv'):
        shutil.copy(R + '/per_layer.csv', pre + 'per_layer_pmc.csv')
    if os.path.exists(R + '/per_layer_trace.csv'):
        shutil.copy(R + '/per_layer_trace.csv', pre + 'per_layer_trace.csv')
    b = json.load(open(R + '/bench.json'))
    print('value %.1f img/s, %.2f ms/step, roofline frac %.4f (%.1f TF), conv traffic %.1f GB (fetch %.1f + write %.1f), alg bytes %.1f GB' %
          (b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['achieved'], (tot_f + tot_w) / 1e9, tot_f / 1e9, tot_w / 1e9, b['roofline']['algorithmic_bytes'] / 1e9))
    print(json.dumps(pm, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
