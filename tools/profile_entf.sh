#!/bin/bash
# Kernel trace of the device-resident filter (C4, N = 1e5, 80 cycles = 240 updates): GPU time per update by kernel.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/entf_trace --output-format csv -- python3 $R/tools/profile_entf.py > $R/gpurun_out/entf_trace.log 2>&1
cd $R
f=$(find gpurun_out/entf_trace -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then
  cp "$f" gpurun_out/entf_kernel_stats.csv
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
upd = 80 * 3
print('GPU busy per update: %.1f us over %d launches' % (tot / upd / 1e3, sum(int(r['Calls']) for r in rows) / upd))
for r in rows[:16]:
    print('  %-52s %5.1f launches  %6.1f us per update  (avg %5.1f us)' % (r['Name'][:52], int(r['Calls']) / upd, float(r['TotalDurationNs']) / upd / 1e3, float(r['AverageNs']) / 1e3))
PY
fi
rm -rf gpurun_out/entf_trace
