#!/usr/bin/env python3
"""profiles/traffic.json from a PMC summary (tools/pmc_summary.py output): HBM bytes per launch of the two map kernels =
2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies the 128-byte requests of coalesced streams as 64 bytes,
MI355X_MICROARCH.md; calibrated here on k_import, whose read size is known).
    python tools/update_traffic.py gpurun_out/prof_pmc_summary.json C5 "note" """
import json
import os
import sys

src, workload, note = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
j = json.load(open(src))
out_path = os.path.join(ROOT, 'profiles', 'traffic.json')
out = json.load(open(out_path)) if os.path.exists(out_path) else {}
entry = {'note': note, 'raw': {}}
for name, v in j.items():
    base = name.split('<')[0].replace('ttm_band::', '')
    if 'FETCH_SIZE' not in v or 'WRITE_SIZE' not in v:
        continue
    if base in ('k_forward_hl', 'k_inverse_rt', 'k_import', 'k_import_flat', 'k_band_forward', 'k_band_inverse', 'k_band_inverse_ring', 'k_band_density'):
        if base == 'k_forward_hl' and '<true' in name:
            continue
        entry['raw'][name] = {'FETCH_SIZE': v['FETCH_SIZE'], 'WRITE_SIZE': v['WRITE_SIZE'], 'avg_ns': v.get('avg_ns')}
        if base not in ('k_import', 'k_import_flat'):
            entry['%s_hbm_bytes_per_launch' % base] = 2 * v['FETCH_SIZE'] + v['WRITE_SIZE']
out[workload] = entry
json.dump(out, open(out_path, 'w'), indent=1)
print(json.dumps(entry, indent=1))
