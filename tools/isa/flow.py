#!/usr/bin/env python3
"""Basic blocks of one kernel in program order with their instruction classes and branch targets (loops show as backward
targets).  usage: flow.py file.s mangled_name [min_insts]"""
import re, sys
path, name = sys.argv[1], sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(name + ':'))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks, cur, lab = [], [], 'entry'
for l in lines[start + 1:end]:
    s = l.strip()
    if re.match(r'^\.LBB[0-9_]+:', s):
        blocks.append((lab, cur)); cur = []; lab = s.split(':')[0]
        continue
    if not s or s.startswith((';', '.')): continue
    if re.match(r'^[a-z_0-9]+(\s|$)', s): cur.append(s)
blocks.append((lab, cur))
order = {b[0]: i for i, b in enumerate(blocks)}
print('%5s %-14s %5s %5s %5s %5s %5s %5s %5s %5s  %s' % ('#', 'block', 'insts', 'vf64', 'lane', 'vothr', 'salu', 'lds', 'vmem', 'smem', 'branches'))
for i, (lab, ins) in enumerate(blocks):
    if len(ins) < mn: continue
    f64 = sum(1 for s in ins if s.startswith('v_') and 'f64' in s.split()[0] and not s.startswith('v_cvt'))
    lane = sum(1 for s in ins if s.startswith(('v_readlane', 'v_writelane')))
    v = sum(1 for s in ins if s.startswith('v_')) - f64 - lane
    sa = sum(1 for s in ins if s.startswith('s_') and not s.startswith(('s_waitcnt', 's_nop', 's_load')))
    lds = sum(1 for s in ins if s.startswith('ds_'))
    vm = sum(1 for s in ins if s.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))
    sm = sum(1 for s in ins if s.startswith('s_load'))
    br = []
    for s in ins:
        if s.startswith(('s_cbranch', 's_branch')):
            t = s.split()[-1]
            if t in order:
                br.append(('^' if order[t] <= i else 'v') + t.replace('.LBB', '') + '(%d)' % order[t])
    print('%5d %-14s %5d %5d %5d %5d %5d %5d %5d %5d  %s' % (i, lab, len(ins), f64, lane, v, sa, lds, vm, sm, ' '.join(br)))
