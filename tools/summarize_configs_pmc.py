#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>/ (tools/profile_configs_pmc.sh) -> profiles/<tag>_configs_pmc.json + profiles/configs_pmc_summary.json.

Per kernel and grid size: average duration from the un-perturbed kernel-trace pass; HBM-side bytes per launch from the
FETCH_SIZE / WRITE_SIZE passes exactly as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes (KiB units; on gfx950
FETCH_SIZE counts 64 B per 128-B request: the read side is doubled — a correction the guide calibrates for wide coalesced
streaming reads only, so `fetch_calibrated` says whether the kernel's reads are of that kind); VALU figures from the SQ pass:
valu_issue_frac = SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x launch duration x 2.4 GHz) — issue slots used, an upper
bound of pipe occupancy (see tools/summarize_profile.py).  `binds` names the larger of the HBM and VALU-issue fractions.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK = 8.0e12
# algorithmic bytes per launch: (description, bytes as a function of the grid) — SURVEY §8d
M98, M392, K = 98208, 392832, 64
ALGO = {
    'loss_fwd_grad_kernel': ('1 M RBFoV pairs, one pass: 40 B read + 20 B grad_pred written (SURVEY §8d books 108 B/pair for separate fwd + bwd)',
                             lambda grid: 60.0e6),
    'iou_pairwise_compact_kernel': ('16 (m + n) + 4 m n (matrix written) or 16 (m + n) + partials (OUT = 2)', None),
    'assign_cols_kernel': ('4 k n read + 12 n written', lambda grid: 4.0 * K * grid + 12.0 * grid),
    'assign_finalize_kernel': ('4 k n re-read (assign-all) + 16 n', lambda grid: 4.0 * K * grid + 16.0 * grid),
}
STREAMING_16B = ('iou_aligned', 'iou_pairwise_compact', 'assign_cols', 'assign_finalize', 'nms_mask')   # 16-byte-per-lane or dword-coalesced row reads


def short(name):
    return name.split('(anonymous namespace)::')[-1].split('(')[0]


def load(pas, tag):
    files = glob.glob(os.path.join(ROOT, 'gpurun_out', 'pmc_' + tag, pas, '**', '*_counter_collection.csv'), recursive=True)
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            key = (short(r['Kernel_Name']), int(r['Grid_Size']))
            out[key][r['Counter_Name']].append(float(r['Counter_Value']))
            meta[key] = dict(vgpr=int(r['VGPR_Count']), sgpr=int(r['SGPR_Count']), lds=int(r['LDS_Block_Size']), scratch=int(r['Scratch_Size']),
                             wg=int(r['Workgroup_Size']))
    return out, meta


def main():
    tag = sys.argv[1]
    trace = glob.glob(os.path.join(ROOT, 'gpurun_out', 'pmc_' + tag, 'trace', '**', '*kernel_trace.csv'), recursive=True)[0]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if 'anonymous namespace' not in r['Kernel_Name']:
            continue
        grid = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
        dur[(short(r['Kernel_Name']), grid)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    fetch, _ = load('fetch', tag)
    write, _ = load('write', tag)
    sq, meta = load('sq', tag)
    kernels = {}
    for key, d in sorted(dur.items()):
        d = d[len(d) // 2:] if len(d) > 200 else d     # the settled half of the trace pass
        name, grid = key
        e = {'kernel': name, 'grid_threads': grid, 'launches_in_trace': len(d), 'avg_ns': sum(d) / len(d), 'min_ns': min(d)}
        mean = lambda t: {c: sum(v) / len(v) for c, v in t.get(key, {}).items()}   # noqa: E731
        f, w, s = mean(fetch), mean(write), mean(sq)
        if 'FETCH_SIZE' in f and 'WRITE_SIZE' in w:
            e['fetch_size_kib'], e['write_size_kib'] = f['FETCH_SIZE'], w['WRITE_SIZE']
            e['fetch_calibrated'] = any(k in name for k in STREAMING_16B)
            e['hbm_read_bytes'] = f['FETCH_SIZE'] * 1024 * 2
            e['hbm_write_bytes'] = w['WRITE_SIZE'] * 1024
            e['hbm_bytes'] = e['hbm_read_bytes'] + e['hbm_write_bytes']
            e['hbm_frac_counter_bytes'] = e['hbm_bytes'] / (e['avg_ns'] * 1e-9) / HBM_PEAK
        if s:
            e['sq'] = s
            if s.get('SQ_WAVES'):
                e['valu_insts_per_wave'] = s.get('SQ_INSTS_VALU', 0) / s['SQ_WAVES']
            if 'SQ_ACTIVE_INST_VALU' in s:
                e['valu_issue_frac'] = s['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * e['avg_ns'] * 2.4)
        for k2, (desc, fn) in ALGO.items():
            if name.startswith(k2) and fn is not None:
                e['algorithmic_bytes'] = fn(grid)
                e['algorithmic_bytes_note'] = desc
                e['hbm_frac_algorithmic'] = e['algorithmic_bytes'] / (e['avg_ns'] * 1e-9) / HBM_PEAK
        if key in meta:
            e['resources'] = meta[key]
        if 'valu_issue_frac' in e and 'hbm_frac_counter_bytes' in e:
            e['binds'] = 'valu-issue' if e['valu_issue_frac'] > e['hbm_frac_counter_bytes'] else 'hbm'
        kernels[f'{name} @ {grid}'] = e
    out = {'tag': tag, 'source': f'gpurun_out/pmc_{tag} (tools/profile_configs_pmc.sh)', 'hbm_peak': HBM_PEAK, 'kernels': kernels}
    json.dump(out, open(os.path.join(ROOT, 'profiles', f'{tag}_configs_pmc.json'), 'w'), indent=1)
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'configs_pmc_summary.json'), 'w'), indent=1)
    for k, e in kernels.items():
        print(f"{k[:78]:78s} {e['avg_ns'] / 1e3:8.2f} us  hbm {e.get('hbm_bytes', 0) / 1e6:8.2f} MB ({e.get('hbm_frac_counter_bytes', 0):.3f})"
              f"  valu-issue {e.get('valu_issue_frac', 0):.3f}  valu/wave {e.get('valu_insts_per_wave', 0):7.0f}  -> {e.get('binds', '?')}")


if __name__ == '__main__':
    main()
