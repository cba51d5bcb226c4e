"""bench.py's guardian of the JSON line (multi-rank runs): a child that exists before rank 0 touches the GPU, prints the
LAST complete line it was sent when rank 0's pipe closes - whether rank 0 finished, was stopped by its watchdog or died."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _guardian():
    return subprocess.Popen([sys.executable, BENCH, '--guardian'], stdin=subprocess.PIPE, stdout=subprocess.PIPE)


def test_guardian_prints_the_last_complete_line_only():
    g = _guardian()
    g.stdin.write(b'{"metric": "m", "multi_gpu_extras_error": "the process ended inside the multi-GPU extras"}\n')
    g.stdin.write(b'{"metric": "m", "strong_scaling": {}}\n')
    g.stdin.write(b'{"metric": "m", "trunc')              # (a writer that died in the middle of a line)
    g.stdin.close()
    out = g.stdout.read().decode().splitlines()
    assert g.wait(timeout=30) == 0
    assert len(out) == 1 and json.loads(out[0]) == {'metric': 'm', 'strong_scaling': {}}


def test_guardian_holds_the_headline_when_the_rank_dies():
    g = _guardian()
    g.stdin.write(b'{"metric": "m", "multi_gpu_extras_error": "the process ended inside the multi-GPU extras"}\n')
    g.stdin.flush()
    g.stdin.close()                                        # (rank 0 is gone: its end of the pipe closes)
    out = g.stdout.read().decode().splitlines()
    assert g.wait(timeout=30) == 0
    assert len(out) == 1 and 'multi_gpu_extras_error' in json.loads(out[0])


def test_no_fork_after_gpu_initialisation_in_bench():
    src = open(BENCH).read()
    assert 'os.fork' not in src


@pytest.mark.gpu
@pytest.mark.parametrize('abort', [False, True])
def test_two_rank_gloo_rehearsal_yields_one_json_line_with_the_node_level_keys(abort):
    """`bench.py --gpus 2 --backend gloo` on the one-GPU box: the never-GPU parent starts two ranks, rank 0's guardian
    prints exactly one JSON line; its node-level keys are pinned here.  With TTM_BENCH_ABORT_IN_EXTRAS=0 rank 0 dies
    inside the extras: still exactly one line (the headline, marked), and a non-zero exit code."""
    env = dict(os.environ)
    if abort:
        env['TTM_BENCH_ABORT_IN_EXTRAS'] = '0'
    res = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--backend', 'gloo', '--steps', '3', '--warmup', '1',
                          '--n', '70001', '--prewarm-seconds', '0', '--no-optimize', '--extras-timeout', '400'],
                         capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout[-2000:] + res.stderr[-2000:]
    js = json.loads(lines[0])
    assert js['n_gpus'] == 2 and js['config']['world_size'] == 2 and js['value'] > 0
    if abort:
        assert res.returncode != 0 and 'multi_gpu_extras_error' in js
        return
    assert res.returncode == 0, res.stderr[-2000:]
    assert 'multi_gpu_extras_error' not in js
    assert 'rccl_ranks' in js                              # (None under gloo: the communicator is RCCL-only)
    ss = js['strong_scaling']
    assert ss['N_total'] == 70001 and ss['value'] > 0 and ss['ms_per_step'] > 0
    es = js['entf_sample_sharded']
    assert es['allreduce_us'] > 0 and 'allreduce' in es and 'evaluations_last_update' in es
