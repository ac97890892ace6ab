#!/usr/bin/env python3
"""Static instruction mix of one kernel from hipcc's `-S` device assembly, per basic block.

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S -o k.s sph2pob_iou.hip   (or sph2pob_{assign,loss,nms}.hip)
    python tools/isa_mix.py k.s 'iou_aligned_compact_kernel<0, 4, true, false>' [--blocks]

Cost weights (cycles per wave64 instruction with several waves per SIMD; tools/ubench/valu_rate*.hip and
MI355X_MICROARCH.md "vector-instruction ISSUE cost"): fp32 FMA / MUL / ADD / SUB 2, quarter-rate transcendentals
(v_rcp / rsq / sqrt / sin / cos / exp / log) 8, every other VALU instruction 4.
"""
import collections
import re
import subprocess
import sys

FMA = re.compile(r'^v_(fma|fmac|mul|add|sub|subrev|mac|mad)_f32')
TRANS = re.compile(r'^v_(rcp|rsq|sqrt|sin|cos|exp|log)_(f32|f16|iflag_f32)')
F64 = re.compile(r'^v_.*_f64')


def classify(op):
    if op.startswith('v_'):
        if F64.match(op):
            return 'valu_f64'
        if TRANS.match(op):
            return 'trans'
        if FMA.match(op):
            return 'fma'
        return 'valu_other'
    if op.startswith('s_'):
        if op.startswith('s_waitcnt'):
            return 'waitcnt'
        if op.startswith(('s_cbranch', 's_branch')):
            return 'branch'
        if op.startswith(('s_load', 's_buffer_load')):
            return 'smem'
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    return 'other'


WEIGHT = {'fma': 2, 'trans': 8, 'valu_other': 4, 'valu_f64': 8}


def demangled_names(path):
    syms = []
    with open(path) as f:
        for ln, line in enumerate(f):
            m = re.match(r'^(_Z[\w]+):', line)
            if m:
                syms.append((ln, m.group(1)))
    names = subprocess.run(['c++filt'], input='\n'.join(s for _, s in syms), capture_output=True, text=True).stdout.split('\n')
    return [(ln, s, n) for (ln, s), n in zip(syms, names)]


def main():
    path, pat = sys.argv[1], sys.argv[2]
    show_blocks = '--blocks' in sys.argv
    syms = demangled_names(path)
    hits = [(ln, s, n) for ln, s, n in syms if pat in n]
    if not hits:
        sys.exit('no kernel matches ' + pat)
    lines = open(path).read().split('\n')
    for ln, sym, name in hits:
        total = collections.Counter()
        blocks = []
        cur_name, cur = 'entry', collections.Counter()
        i = ln + 1
        while i < len(lines) and not lines[i].startswith('\t.section') and not lines[i].startswith('.Lfunc_end'):
            t = lines[i].strip()
            i += 1
            if not t or t.startswith((';', '.p2align', '.')) and not t.startswith('.LBB'):
                continue
            m = re.match(r'^(\.LBB\w+):', t)
            if m:
                blocks.append((cur_name, cur))
                cur_name, cur = m.group(1), collections.Counter()
                continue
            op = t.split()[0]
            c = classify(op)
            cur[c] += 1
            total[c] += 1
        blocks.append((cur_name, cur))
        cyc = sum(WEIGHT.get(k, 0) * v for k, v in total.items())
        print(f'== {name[:140]}')
        print('   total', dict(total), 'valu_cycles', cyc)
        # resource lines
        for j in range(i, min(i + 80, len(lines))):
            t = lines[j].strip()
            if re.match(r'^; (NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte)', t):
                print('   ' + t[2:])
        if show_blocks:
            for bn, c in blocks:
                n = sum(c.values())
                if n >= 8:
                    bc = sum(WEIGHT.get(k, 0) * v for k, v in c.items())
                    print(f'   {bn:14s} n={n:5d} cyc={bc:6d} ' + ' '.join(f'{k}={v}' for k, v in sorted(c.items())))


if __name__ == '__main__':
    main()
