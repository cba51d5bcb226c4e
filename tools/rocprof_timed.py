#!/usr/bin/env python3
"""Average duration of the TIMED launches of a `bench.py` run under `rocprofv3 --kernel-trace` (not of the thousands of
pre-warm launches in front of them, which is what `--stats` averages).

    python tools/rocprof_timed.py <trace dir> <bench JSON line file> [--out profiles/r05_timed_launches.json]

bench.py counts its steps: `kernel_timing.timed_launches = {first_step, steps}` says which dispatches of the forward and of the
inverse kernel (one each per step, in stream order) lie between the contract's two barriers.  This tool sorts the
dispatches of each kernel in the trace by start time, takes that range and writes mean / min / max next to the `--stats`-style
average over ALL dispatches; bench.py prints the committed figures beside its HIP-event times (`roofline.rocprof_timed_avg_launch_ms`)."""
import csv
import glob
import json
import os
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    out = 'profiles/r05_timed_launches.json'
    if '--out' in sys.argv:
        out = sys.argv[sys.argv.index('--out') + 1]
        args = [a for a in args if a != out]
    trace_dir, line_file = args[0], args[1]
    line = None
    for ln in open(line_file):
        if ln.startswith('{"metric"'):
            line = json.loads(ln)
    if line is None:
        raise SystemExit('no bench line in %s' % line_file)
    kt = line['kernel_timing']['timed_launches']
    first, steps = int(kt['first_step']), int(kt['steps'])
    workload = line['config']['workload'].split(':')[0]
    names = {'forward': line['roofline']['forward_kernel'], 'inverse': line['roofline']['inverse_kernel']}
    files = glob.glob(os.path.join(trace_dir, '**', '*kernel_trace.csv'), recursive=True)
    if not files:
        raise SystemExit('no *kernel_trace.csv under %s' % trace_dir)
    rows = {'forward': [], 'inverse': []}
    for f in files:
        for r in csv.DictReader(open(f)):
            kn = r.get('Kernel_Name', '')
            for which, short in names.items():
                # (the trace carries the full template instantiation, e.g. "void ttm_band::k_band_forward<1, 2>(...)")
                if (short + '<') in kn or (short + '(') in kn:
                    rows[which].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
    res = {'steps': steps, 'first_step': first, 'bench_ms_per_step': line['ms_per_step'],
           'bench_forward_event_ms': line['kernel_timing']['forward_event_ms'],
           'bench_inverse_event_ms': line['kernel_timing']['inverse_event_ms'],
           'command': 'rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps %d --warmup %d' % (line['steps'], line['warmup'])}
    for which, rr in rows.items():
        rr.sort()
        sel = rr[first:first + steps]
        if len(sel) != steps:
            raise SystemExit('%s: %d dispatches in the trace, timed range [%d, %d)' % (which, len(rr), first, first + steps))
        d = [(b - a) * 1e-6 for a, b in sel]
        alld = [(b - a) * 1e-6 for a, b in rr]
        res.update({which + '_kernel': names[which], which + '_avg_ms': sum(d) / len(d), which + '_min_ms': min(d), which + '_max_ms': max(d),
                    which + '_all_dispatches': len(rr), which + '_all_avg_ms': sum(alld) / len(alld)})
    # the timed region by the trace's own clock: first forward start -> last inverse end
    t0 = sorted(rows['forward'])[first][0]
    t1 = sorted(rows['inverse'])[first + steps - 1][1]
    res['trace_ms_per_step'] = (t1 - t0) * 1e-6 / steps
    data = {}
    if os.path.exists(out):
        data = json.load(open(out))
    data[workload] = res
    json.dump(data, open(out, 'w'), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
