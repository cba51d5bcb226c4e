#!/usr/bin/env python3
"""Basic blocks of one kernel in hipcc's -S output, largest first: label, #instructions, VALU f64 / VALU other / SALU / LDS / VMEM.
usage: blocks.py file.s mangled_prefix [n]"""
import re, sys
path, name = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(name) and ':' in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks, cur, lab = [], [], 'entry'
for l in lines[start + 1:end]:
    s = l.strip()
    if re.match(r'^\.LBB[0-9_]+:', s):
        blocks.append((lab, cur)); cur = []; lab = s.split(':')[0]
        continue
    if not s or s.startswith((';', '.')): continue
    m = s.split()[0]
    if re.match(r'^[a-z_0-9]+$', m): cur.append(s)
blocks.append((lab, cur))
def stats(ins):
    f64 = sum(1 for s in ins if s.startswith('v_') and 'f64' in s.split()[0] and not s.startswith('v_cvt'))
    v = sum(1 for s in ins if s.startswith('v_'))
    sa = sum(1 for s in ins if s.startswith('s_') and not s.startswith(('s_waitcnt', 's_nop', 's_load')))
    return f64, v - f64, sa, sum(1 for s in ins if s.startswith('ds_')), sum(1 for s in ins if s.startswith(('global_', 'buffer_', 'flat_'))), \
        sum(1 for s in ins if s.startswith('s_waitcnt')), sum(1 for s in ins if s.startswith('s_nop')), sum(1 for s in ins if s.startswith('s_load'))
print('%-14s %6s %6s %6s %6s %5s %5s %5s %5s %5s' % ('block', 'insts', 'vf64', 'vother', 'salu', 'lds', 'vmem', 'wait', 'nop', 'smem'))
for lab, ins in sorted(blocks, key=lambda b: -len(b[1]))[:top]:
    print('%-14s %6d %6d %6d %6d %5d %5d %5d %5d %5d' % ((lab, len(ins)) + stats(ins)))
