#!/usr/bin/env python3
"""Print, per code region of an ISA dump (split at s_barrier), the highest VGPR index touched
and instruction class counts.  usage: vgpr_profile.py /tmp/kdump.s"""
import re, sys, collections
lines = open(sys.argv[1]).read().splitlines()
reg = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')
region, start = 0, 0
def flush(lo, hi, idx):
    mx = -1; cnt = collections.Counter()
    for l in lines[lo:hi]:
        t = l.strip().split()
        if not t or t[0].startswith((';', '.')) or t[0].endswith(':'): continue
        op = t[0]
        for m in reg.finditer(l):
            if m.group(1): mx = max(mx, int(m.group(1)))
            else: mx = max(mx, int(m.group(3)))
        if op.startswith('v_pk'): cnt['v_pk'] += 1
        elif op.startswith('v_'): cnt['valu'] += 1
        elif op.startswith('ds_'): cnt['ds'] += 1
        elif op.startswith('buffer_') or op.startswith('global_'): cnt['vmem'] += 1
        elif op.startswith('scratch_'): cnt['scratch'] += 1
        elif op.startswith('s_waitcnt'): cnt['wait'] += 1
        elif op.startswith('s_'): cnt['salu'] += 1
    print(f"region {idx:2d} lines {lo:5d}-{hi:5d} maxv={mx:3d} {dict(cnt)}")
for i, l in enumerate(lines):
    if 's_barrier' in l:
        flush(start, i, region); region += 1; start = i
flush(start, len(lines), region)
