#!/bin/bash
# Kernel durations of the small-D configurations (C2b at N = 1e6, C3 at N = 5e5) from the kernel trace: the launches of
# these configurations are shorter than a Python call, so event timing around a Python loop measures the host.
# usage: tools/profile_small_d.sh [tag]      (environment, e.g. TTM_U_LOADER=1, is passed on)
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-sd}
cd /tmp && export TMPDIR=/tmp
SMALL_D_N=1e6 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_c2b --output-format csv -- python3 $R/tools/small_d.py C2b > $R/gpurun_out/${TAG}_c2b.log 2>&1
SMALL_D_N=5e5 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_c3 --output-format csv -- python3 $R/tools/small_d.py C3 > $R/gpurun_out/${TAG}_c3.log 2>&1
cd $R
for d in ${TAG}_c2b ${TAG}_c3; do
  f=$(find gpurun_out/$d -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then
    cp "$f" gpurun_out/$d.kernel_stats.csv
    python3 - "$f" $d <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print('%s  %-44s calls %6s avg %9.1f ns min %s' % (sys.argv[2], r['Name'][:44], r['Calls'], float(r['AverageNs']), r['MinNs']))
PY
  fi
  rm -rf gpurun_out/$d
done
