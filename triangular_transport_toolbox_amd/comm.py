"""
Process-wide communicator for the one collective of the path (include/ttm.h "C1": ttm_allreduce_f64 / _i32 over
RCCL / xGMI).  torch.distributed is the control plane only: it carries the 128-byte rendezvous id from rank 0 to the
other ranks once; every data-path reduction afterwards is one C-ABI call on the caller's stream.

A communicator is created when the process group's backend is 'nccl' (= RCCL on ROCm, one rank per GPU).  With any
other backend (gloo rehearsals that put several ranks on one GPU - RCCL refuses duplicate devices) `get()` returns
None and the class falls back to torch.distributed for the same reductions.
"""
import ctypes

_state = {'key': None, 'comm': None, 'lib': None}


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None


def get(lib, force=False):
    """ctypes handle of this process's communicator, or None (single rank / non-RCCL backend)."""
    dist = _dist()
    if dist is None:
        return None
    key = (dist.get_rank(), dist.get_world_size(), id(lib))
    if _state['key'] == key:
        return _state['comm']
    if not force and dist.get_backend() != 'nccl':
        _state.update(key=key, comm=None, lib=lib)
        return None
    # Every rank must take the same branch at every step of the rendezvous (a rank that raises while the others wait in
    # a collective hangs the job): failures are agreed on through torch.distributed and turn into "no communicator" -
    # the class then reduces through torch.distributed, as it does for non-RCCL process groups.
    import warnings

    import torch
    buf = ctypes.create_string_buffer(128)
    ok = 1
    if dist.get_rank() == 0:
        ok = 1 if lib.ttm_comm_unique_id(buf) == 0 else 0
    box = [(ok, buf.raw)]
    dist.broadcast_object_list(box, src=0)
    ok, raw = box[0]
    handle = ctypes.c_void_p()
    if ok:
        ok = 1 if lib.ttm_comm_create(ctypes.c_char_p(raw), dist.get_rank(), dist.get_world_size(), ctypes.byref(handle)) == 0 else 0
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        if ok and handle:
            lib.ttm_comm_destroy(handle)
        warnings.warn('libttm communicator not available (%s): reductions go through torch.distributed'
                      % lib.ttm_comm_last_error().decode())
        _state.update(key=key, comm=None, lib=lib)
        return None
    _state.update(key=key, comm=handle, lib=lib)
    return handle


def destroy():
    if _state['comm'] is not None and _state['lib'] is not None:
        _state['lib'].ttm_comm_destroy(_state['comm'])
    _state.update(key=None, comm=None, lib=None)
