#!/bin/bash
# kernel + memory-copy trace of the un-cached C5 step: the launch sequence of one iteration with its gaps
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/tr_unc --output-format csv -- python3 $R/tools/time_uncached.py > $R/gpurun_out/tr_unc.log 2>&1
cd $R && python3 - <<'PY'
import csv, glob
rows = []
for path in glob.glob('gpurun_out/tr_unc/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-60:]))
for path in glob.glob('gpurun_out/tr_unc/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '') + ' ' + r.get('Size', '')))
rows.sort()
# the last 'fused 1' deferred block: find the last 8 occurrences of k_uform and print around the 4th from the end
idx = [i for i, r in enumerate(rows) if 'k_uform' in r[2]]
i0 = idx[-120]
t0 = rows[i0][0]
prev_end = None
for s, e, n in rows[i0 - 3:i0 + 22]:
    print('%9.1f us  +%6.1f  dur %7.1f  %s' % ((s - t0) / 1e3, 0 if prev_end is None else (s - prev_end) / 1e3, (e - s) / 1e3, n))
    prev_end = e
PY
rm -rf $R/gpurun_out/tr_unc
