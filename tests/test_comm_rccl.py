"""
The collective of the path against the real RCCL of the GPU box (`-m gpu`; include/ttm.h "C1", csrc/ttm_comm.cpp).

A 1-GPU box cannot run two RCCL ranks (RCCL refuses two ranks on one device), so what is covered here is
* the binding itself: run-time lookup of librccl, the NCCL 2 ABI (id size, dtype / op codes), a ONE-rank communicator
  reducing device vectors in place on a HIP stream;
* the rendezvous of `comm.get()` when communicator creation FAILS on the ranks (two ranks on the one GPU): every rank
  must agree on the fallback to torch.distributed instead of one raising while the other waits.
The reductions themselves are exercised with two ranks on CPU by tests/test_distributed_gloo.py (host test double).
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_communicator_reduces_device_vectors_in_place():
    import torch
    from triangular_transport_toolbox_amd import _capi
    lib = _capi.load()
    buf = ctypes.create_string_buffer(128)
    assert lib.ttm_comm_unique_id(buf) == 0, lib.ttm_comm_last_error().decode()
    assert any(b != 0 for b in buf.raw)
    handle = ctypes.c_void_p()
    assert lib.ttm_comm_create(ctypes.c_char_p(buf.raw), 0, 1, ctypes.byref(handle)) == 0, lib.ttm_comm_last_error().decode()
    try:
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        x = torch.arange(1, 42, dtype=torch.float64, device='cuda') * 0.125
        ref = x.clone()
        assert lib.ttm_allreduce_f64(handle, ctypes.c_void_p(x.data_ptr()), x.numel(), 0, st) == 0, lib.ttm_comm_last_error().decode()
        assert lib.ttm_allreduce_f64(handle, ctypes.c_void_p(x.data_ptr()), x.numel(), 1, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(x, ref)                              # one rank: sum and max are the identity, bit for bit
        i = torch.tensor([3, -7, 2 ** 31 - 1, 0], dtype=torch.int32, device='cuda')
        iref = i.clone()
        assert lib.ttm_allreduce_i32(handle, ctypes.c_void_p(i.data_ptr()), i.numel(), 1, st) == 0
        assert lib.ttm_allreduce_i32(handle, ctypes.c_void_p(i.data_ptr()), i.numel(), 0, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(i, iref)
        # argument errors are reported, not passed on to RCCL
        assert lib.ttm_allreduce_f64(handle, None, 4, 0, st) != 0 and lib.ttm_comm_last_error().decode()
        assert lib.ttm_allreduce_f64(handle, ctypes.c_void_p(x.data_ptr()), 4, 7, st) != 0
    finally:
        assert lib.ttm_comm_destroy(handle) == 0


_WORKER = r'''
import os, sys, warnings, json
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
rank = int(os.environ['RANK'])
dist.init_process_group('gloo', rank=rank, world_size=2)
torch.cuda.set_device(rank if os.environ.get('TTM_TEST_ONE_GPU_PER_RANK') else 0)
from triangular_transport_toolbox_amd import _capi, comm
lib = _capi.load()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    handle = comm.get(lib, force=True)
out = {'rank': rank, 'handle': handle is not None, 'warned': [str(x.message) for x in w]}
if handle is not None:                        # (an RCCL that accepts two ranks on one device: then it must reduce)
    x = torch.full((5,), float(rank + 1), dtype=torch.float64, device='cuda')
    import ctypes
    rc = lib.ttm_allreduce_f64(handle, ctypes.c_void_p(x.data_ptr()), 5, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    out['sum'] = [rc] + x.cpu().tolist()
    # the Gram-matrix size (m^2 = 400 doubles, SURVEY section 8a' C1) and the int32 maximum the bisection caps / the radix
    # select's bin counts go through (ttm_allreduce_i32)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.arange(400, dtype=torch.float64, device='cuda') * (rank + 1)
    rc = lib.ttm_allreduce_f64(handle, ctypes.c_void_p(g.data_ptr()), 400, 0, st)
    torch.cuda.synchronize()
    out['gram'] = [rc, float((g - 3.0 * torch.arange(400, dtype=torch.float64, device='cuda')).abs().max().item())]
    it = torch.tensor([7 + rank, 3 - rank, 100 * rank], dtype=torch.int32, device='cuda')
    rc = lib.ttm_allreduce_i32(handle, ctypes.c_void_p(it.data_ptr()), 3, 1, st)
    torch.cuda.synchronize()
    out['imax'] = [rc] + it.cpu().tolist()
    cnt = torch.full((256,), rank + 1, dtype=torch.int32, device='cuda')
    rc = lib.ttm_allreduce_i32(handle, ctypes.c_void_p(cnt.data_ptr()), 256, 0, st)
    torch.cuda.synchronize()
    out['isum'] = [rc, int(cnt.min().item()), int(cnt.max().item())]
# whatever happened, both ranks are still in step: a collective over the control plane completes
t = torch.tensor([rank + 1.0])
dist.all_reduce(t)
out['after'] = float(t.item())
print('RESULT ' + json.dumps(out), flush=True)
comm.destroy()
dist.destroy_process_group()
'''


def _run_two_ranks(tmp_path, extra_env):
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % {'root': ROOT})
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0', **extra_env)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=240)
            outs.append(o)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    import json
    res = []
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-2000:]
        line = [ln for ln in o.splitlines() if ln.startswith('RESULT ')]
        assert line, o[-2000:]
        res.append(json.loads(line[0][7:]))
    return res


def _check_extras(res):
    for r in res:
        assert r['gram'] == [0, 0.0], r                        # 400 doubles summed over the two ranks
        assert r['imax'] == [0, 8, 3, 100], r                  # int32 maximum
        assert r['isum'] == [0, 3, 3], r                       # int32 sum (the radix select's bin counts)


def test_two_ranks_on_one_gpu_agree_on_the_fallback(tmp_path):
    res = _run_two_ranks(tmp_path, {})
    print('communicator created:', res[0]['handle'], '|', res[0]['warned'])
    assert res[0]['handle'] == res[1]['handle']                # the ranks agree
    assert res[0]['after'] == res[1]['after'] == 3.0            # and are still in step afterwards
    if res[0]['handle']:
        assert res[0]['sum'] == res[1]['sum'] == [0] + [3.0] * 5
        _check_extras(res)
    else:
        assert all('communicator not available' in ' '.join(r['warned']) for r in res)


def test_two_ranks_on_two_gpus_reduce_over_rccl(tmp_path):
    """The real thing, on a box with at least two devices (skipped on the one-GPU boxes): one rank per GPU, the RCCL
    communicator of csrc/ttm_comm.cpp created through comm.get(), an in-place sum of a device vector over xGMI."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    res = _run_two_ranks(tmp_path, {'TTM_TEST_ONE_GPU_PER_RANK': '1'})
    assert res[0]['handle'] and res[1]['handle'], res
    assert res[0]['sum'] == res[1]['sum'] == [0] + [3.0] * 5
    _check_extras(res)
    assert res[0]['after'] == res[1]['after'] == 3.0
