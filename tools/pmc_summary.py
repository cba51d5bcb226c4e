#!/usr/bin/env python3
"""
Summarise rocprofv3 counter-collection CSVs per kernel.

    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ ...

Prints one JSON object: {kernel: {counter: mean per dispatch, ..., 'dispatches': n, 'avg_ns': t}}.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; they are converted to bytes here.  The
gfx950 calibration of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests as
64 B for wide coalesced streams) is applied by the caller, against the k_import calibration kernel
whose byte count is known.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    return name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').strip()


def main(dirs):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    k = short(row['Kernel_Name'])
                    v = float(row['Counter_Value'])
                    if row['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
                        v *= 1024.0
                    acc[k][row['Counter_Name']].append(v)
                    acc[k]['_ns_' + row['Counter_Name']].append(float(row['End_Timestamp']) - float(row['Start_Timestamp']))
    out = {}
    for k, cs in acc.items():
        o = {}
        for c, vals in cs.items():
            if c.startswith('_ns_'):
                continue
            o[c] = sum(vals) / len(vals)
            o['dispatches'] = len(vals)
        ns = [v for c, vals in cs.items() if c.startswith('_ns_') for v in vals]
        o['avg_ns'] = sum(ns) / len(ns)
        out[k] = o
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == '__main__':
    main(sys.argv[1:])
