"""Summarise a tools/profile.sh output directory into profiles/<tag>_*.{csv,json} (committed evidence).

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so the read side is
doubled; WRITE_SIZE is exact for streaming stores.  Counters come from separate --pmc passes."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
kernel_key = sys.argv[2] if len(sys.argv) > 2 else 'iou_aligned'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', 'prof_' + tag)
dst = os.path.join(root, 'profiles')
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, 'trace', '*', '*_kernel_stats.csv'))[0]
shutil.copy(stats, os.path.join(dst, f'{tag}_kernel_stats.csv'))
summary = {'tag': tag, 'kernels': {}}
for r in csv.DictReader(open(stats)):
    if kernel_key in r['Name']:
        summary['kernels'][r['Name']] = {'calls': int(r['Calls']), 'avg_ns': float(r['AverageNs']),
                                         'min_ns': float(r['MinNs']), 'max_ns': float(r['MaxNs'])}
counters = collections.defaultdict(lambda: collections.defaultdict(list))
durations = collections.defaultdict(list)
meta = {}
for pas in ('pmc_fetch', 'pmc_write', 'pmc_sq'):
    files = glob.glob(os.path.join(src, pas, '*', '*_counter_collection.csv'))
    if not files:
        continue
    for r in csv.DictReader(open(files[0])):
        if kernel_key in r['Kernel_Name']:
            counters[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
            if pas == 'pmc_sq' and r['Counter_Name'] == 'SQ_ACTIVE_INST_VALU':
                durations[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
            meta[r['Kernel_Name']] = {'vgpr': int(r['VGPR_Count']), 'sgpr': int(r['SGPR_Count']),
                                      'lds': int(r['LDS_Block_Size']), 'scratch': int(r['Scratch_Size']),
                                      'grid': int(r['Grid_Size']), 'wg': int(r['Workgroup_Size'])}
for k, cs in counters.items():
    mean = {c: sum(v) / len(v) for c, v in cs.items()}
    entry = summary['kernels'].setdefault(k, {})
    entry['pmc_mean_per_launch'] = mean
    entry['resources'] = meta[k]
    if 'FETCH_SIZE' in mean and 'WRITE_SIZE' in mean:
        rd = mean['FETCH_SIZE'] * 1024 * 2   # gfx950: FETCH_SIZE counts 64 B per 128-B request
        wr = mean['WRITE_SIZE'] * 1024
        entry['hbm_read_bytes_per_launch'] = rd
        entry['hbm_write_bytes_per_launch'] = wr
        entry['hbm_bytes_per_launch'] = rd + wr
    if 'SQ_INSTS_VALU' in mean and 'SQ_WAVES' in mean:
        entry['valu_insts_per_wave'] = mean['SQ_INSTS_VALU'] / mean['SQ_WAVES']
    if 'SQ_ACTIVE_INST_VALU' in mean and entry.get('avg_ns'):
        # SQ_ACTIVE_INST_VALU (quad-cycles, summed over waves) x 4 / (1024 SIMDs x cycles of one launch at
        # the 2.4 GHz maximum clock; launch duration = the un-perturbed kernel-trace average, the counter passes
        # serialise launches and run them slower).  NB the counter sums per-wave "a VALU instruction is in flight" time
        # (~4 cycles per instruction whatever its kind), so it measures issue slots used, not a busy fraction of the pipe.
        entry['valu_active_frac'] = mean['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * entry['avg_ns'] * 2.4)
        if durations.get(k):
            entry['pmc_pass_avg_ns'] = sum(durations[k]) / len(durations[k])
json.dump(summary, open(os.path.join(dst, f'{tag}_summary.json'), 'w'), indent=1)
# bench.py reads the dominant kernel's measured traffic from here
dom = max(summary['kernels'].items(), key=lambda kv: kv[1].get('calls', 0))
cfg = {'pairs': int(os.environ.get('SPH2POB_PROFILE_PAIRS', 1000000)), 'variant': os.environ.get('SPH2POB_PROFILE_VARIANT', 'standard'),
       'arithmetic': os.environ.get('SPH2POB_PROFILE_ARITHMETIC', 'fast')}
json.dump({'iou_aligned': {'kernel': dom[0], 'hbm_bytes_per_launch': dom[1].get('hbm_bytes_per_launch'),
                           'valu_active_frac': dom[1].get('valu_active_frac'),
                           'avg_ns': dom[1].get('avg_ns'), 'source': f'profiles/{tag}_summary.json', 'config': cfg}},
          open(os.path.join(dst, 'pmc_summary.json'), 'w'), indent=1)
print(json.dumps(summary, indent=1)[:3000])
