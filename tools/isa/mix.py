#!/usr/bin/env python3
"""Instruction mix of one kernel (or one basic-block range of it) in hipcc's -S output.
usage: mix.py file.s mangled_prefix [first_label last_label]"""
import re, sys, collections
path, name = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(name) and l.rstrip().split(':')[0].startswith(name) and ':' in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
if len(sys.argv) > 4:
    a = next(i for i, l in enumerate(body) if l.startswith(sys.argv[3] + ':'))
    b = next(i for i, l in enumerate(body) if l.startswith(sys.argv[4] + ':'))
    body = body[a:b]
cnt = collections.Counter()
cat = collections.Counter()
def category(m):
    if m.startswith('v_'):
        if 'f64' in m and not m.startswith('v_cvt') and not m.startswith('v_cmp'):
            if m.startswith(('v_rcp', 'v_rsq', 'v_sqrt')): return 'valu_f64_trans'
            return 'valu_f64'
        if m.startswith('v_cmp') and 'f64' in m: return 'valu_f64_cmp'
        if m.startswith('v_cvt'): return 'valu_cvt'
        if m.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): return 'valu_lane'
        return 'valu_32'
    if m.startswith('s_load') or m.startswith('s_buffer_load'): return 'smem'
    if m.startswith('s_waitcnt'): return 's_waitcnt'
    if m.startswith('s_nop'): return 's_nop'
    if m.startswith('s_barrier'): return 's_barrier'
    if m.startswith(('s_cbranch', 's_branch')): return 's_branch'
    if m.startswith('s_'): return 'salu'
    if m.startswith('ds_'): return 'lds'
    if m.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    return 'other'
for l in body:
    l = l.strip()
    if not l or l.startswith((';', '.', '_')) or l.endswith(':'):
        continue
    m = l.split()[0]
    if not re.match(r'^[a-z_0-9]+$', m): continue
    cnt[m] += 1
    cat[category(m)] += 1
print('lines', len(body))
for k, v in sorted(cat.items(), key=lambda kv: -kv[1]): print('%-16s %d' % (k, v))
print()
for k, v in cnt.most_common(60): print('%-28s %d' % (k, v))
