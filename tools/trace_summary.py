#!/usr/bin/env python3
"""Per-kernel, per-grid-size durations out of a rocprofv3 --kernel-trace CSV (median / mean over the launches; a stats file
averages over every grid size a kernel was launched with, which hides what each configuration costs).

    python tools/trace_summary.py <dir-or-kernel_trace.csv> [name-substring ...]
"""
import collections
import csv
import glob
import os
import statistics
import sys


def main():
    path = sys.argv[1]
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True))[0]
    want = sys.argv[2:]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name']
        if want and not any(w in name for w in want):
            continue
        short = name.split('(anonymous namespace)::')[-1].split('(')[0][:70]
        agg[(short, int(r['Grid_Size_X']), int(r['Grid_Size_Y']))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for (name, gx, gy), v in sorted(agg.items()):
        if len(v) >= 20:
            print(f'{name:72s} grid {gx:>9d} x {gy:<3d} n {len(v):>6d}  median {statistics.median(v) / 1e3:8.2f} us  mean {statistics.mean(v) / 1e3:8.2f}'
                  f'  min {min(v) / 1e3:8.2f}')


if __name__ == '__main__':
    main()
