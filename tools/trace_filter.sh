#!/bin/bash
# kernel + memory-copy trace of the device-resident filter: the launch sequence of one update with its gaps
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
TTM_FILTER_CYCLES=40 rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/tr_flt --output-format csv -- python3 $R/tools/time_filter_quick.py > $R/gpurun_out/tr_flt.log 2>&1
cd $R && python3 - <<'PY'
import csv, glob
rows = []
for path in glob.glob('gpurun_out/tr_flt/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-44:]))
for path in glob.glob('gpurun_out/tr_flt/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
rows.sort()
idx = [i for i, r in enumerate(rows) if 'k_lorenz63' in r[2]]
i0, i1 = idx[-3], idx[-2]          # one whole cycle
t0 = rows[i0][0]
prev_end = None
busy = 0
for s, e, n in rows[i0:i1]:
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    busy += (e - s) / 1e3
    if 'objective_sep' in n or 'reduce_partials_mark' in n:
        prev_end = e
        continue
    print('%9.1f us  +%6.1f  dur %7.1f  %s' % ((s - t0) / 1e3, gap, (e - s) / 1e3, n))
    prev_end = e
print('cycle wall %.1f us, GPU busy %.1f us, launches %d' % ((rows[i1][0] - t0) / 1e3, busy, i1 - i0))
PY
rm -rf $R/gpurun_out/tr_flt
