#!/usr/bin/env python3
"""profiles/fp64_counts.json and profiles/r03_fp64_pmc_summary.json from the PMC summaries of tools/fp64_counts.sh:
per workload and kernel the counted fp64 operations of a launch, 64 lanes x (2 FMA + ADD + MUL + TRANS [+ 256 per
fp64 MFMA 16x16x4]) wave-instructions, the VALU / SALU / LDS instruction counts and the wait / activity counters.
    python tools/fp64_counts.py gpurun_out C2a C3int C5int C5"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

src = sys.argv[1]
out, full = {}, {}
INTERESTING = ('k_forward', 'k_inverse_bisect', 'k_objective', 'k_gram', 'k_band', 'k_inverse_rt', 'k_forward_hl', 'k_basis', 'k_table', 'k_int_')


def launches_per_call(kernel):
    """The reference-sequence bisection is TWO launches per inversion: samples 1..N-1, then the replay of sample 0 with the
    largest iteration count as its cap (TM:3952).  Flops are counted per INVERSION: the sum over both launches."""
    k = kernel.replace(' ', '')
    if k.startswith('k_inverse_bisect<') and k.endswith(',false>'):
        return 2
    if k.startswith('k_int_root<') and k.endswith(',false>'):
        return 2
    return 1

for w in sys.argv[2:]:
    path = os.path.join(src, 'fp64_%s.json' % w)
    if not os.path.exists(path):
        continue
    j = json.load(open(path))
    N = bench.WORKLOADS[w][1]
    kernels = {}
    for name, v in j.items():
        short = name.replace('ttm_band::', '')
        if not short.startswith(INTERESTING) or 'SQ_INSTS_VALU_FMA_F64' not in v:
            continue
        flop = 64.0 * (2 * v['SQ_INSTS_VALU_FMA_F64'] + v['SQ_INSTS_VALU_ADD_F64'] + v['SQ_INSTS_VALU_MUL_F64'] + v['SQ_INSTS_VALU_TRANS_F64'])
        flop += 2.0 * 16 * 16 * 4 * v.get('SQ_INSTS_VALU_MFMA_F64', 0.0)          # v_mfma_f64_16x16x4: 1024 FMA per wave-instruction
        lpc = launches_per_call(short)
        kernels[short] = {'flop_per_launch': flop, 'launches_per_call': lpc, 'flop_per_call': flop * lpc,
                          'valu_per_call': v['SQ_INSTS_VALU'] * lpc, 'salu_per_call': (v.get('SQ_INSTS_SALU') or 0.0) * lpc,
                          'valu_per_launch': v['SQ_INSTS_VALU'], 'salu_per_launch': v.get('SQ_INSTS_SALU'),
                          'lds_per_launch': v.get('SQ_INSTS_LDS'), 'mfma_f64_per_launch': v.get('SQ_INSTS_VALU_MFMA_F64'),
                          'avg_ns_profiled': v.get('avg_ns'), 'dispatches': v.get('dispatches'),
                          'TFLOPs_profiled': flop / max(v.get('avg_ns', 1.0), 1.0) / 1e3,
                          'valu_active_frac': (v['SQ_ACTIVE_INST_VALU'] * 4.0 / v['SQ_BUSY_CYCLES'] / 4.0 if v.get('SQ_BUSY_CYCLES') else None),
                          'wait_any_frac': (v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES'] if v.get('SQ_WAVE_CYCLES') else None),
                          'wait_inst_frac': (v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES'] if v.get('SQ_WAVE_CYCLES') else None)}
        full.setdefault(w, {})[short] = v

    def pick(prefix):
        c = [k for k in kernels if k.startswith(prefix)]
        return max(c, key=lambda k: kernels[k]['flop_per_launch']) if c else None
    # (the bisection and the Newton search are the same template with a flag: the bisection is the one with more flops)
    inv = [k for k in kernels if k.startswith(('k_inverse_bisect', 'k_int_root'))]
    bis = [k for k in inv if launches_per_call(k) == 2]
    newt = [k for k in inv if launches_per_call(k) == 1]
    out[w] = {'N': N, 'kernels': kernels, 'forward_kernel': pick('k_int_forward') or pick('k_band_forward') or pick('k_forward'),
              'inverse_kernel': bis[0] if bis else (pick('k_band_inverse') or pick('k_inverse_rt')),
              'newton_kernel': newt[0] if newt else None,
              'definition': 'flop_per_call = 64 lanes x (2 FMA + ADD + MUL + TRANS) fp64 wave-instructions summed over the launches '
                            'of ONE call (a reference-sequence inversion is two launches: samples 1..N-1, then sample 0)'}
json.dump(out, open(os.path.join(ROOT, 'profiles', 'fp64_counts.json'), 'w'), indent=1, sort_keys=True)
json.dump(full, open(os.path.join(ROOT, 'profiles', 'r05_fp64_pmc_summary.json'), 'w'), indent=1, sort_keys=True)
for w, c in out.items():
    print(w, 'forward', c['forward_kernel'], 'inverse', c['inverse_kernel'], 'newton', c['newton_kernel'])
    for k, v in sorted(c['kernels'].items(), key=lambda kv: -kv[1]['flop_per_call'])[:8]:
        print('   %-60s %.3e flop  %.3e VALU  %8.1f us  %.2f TF/s  wait %.2f' % (k[:60], v['flop_per_call'], v['valu_per_call'],
                                                                               (v['avg_ns_profiled'] or 0) / 1e3, v['TFLOPs_profiled'], v['wait_any_frac'] or 0))
