#!/usr/bin/env python3
"""VGPRs / SGPR spills / scratch of every kernel of one translation unit (cross-compiled for gfx950):
   tools/kernel_resources.py csrc/ttm_band.hip [name filter] [-- extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if '--' in args:
    i = args.index('--'); extra = args[i + 1:]; args = args[:i]
src = args[0] if os.path.isabs(args[0]) else os.path.join(ROOT, 'triangular_transport_toolbox_amd', args[0])
flt = args[1] if len(args) > 1 else ''
cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-fPIC', '-DNDEBUG', '-x', 'hip', '-c', src,
       '--cuda-device-only', '-Rpass-analysis=kernel-resource-usage', '-o', '/tmp/kernel_resources.o'] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::', '').split('(')[0]
        rows[cur] = {}
        continue
    m = re.search(r'remark:\s+(VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)', line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print('%-70s vgpr %3d sgpr-spill %4d vgpr-spill %4d scratch %4d occ %d' % (k[-70:], v.get('VGPRs', -1), v.get('SGPRs Spill', -1), v.get('VGPRs Spill', -1),
                                                                        v.get('ScratchSize [bytes/lane]', -1), v.get('Occupancy [waves/SIMD]', -1)))
