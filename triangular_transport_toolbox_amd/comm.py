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
    from . import _capi
    buf = ctypes.create_string_buffer(128)
    if dist.get_rank() == 0:
        rc = lib.ttm_comm_unique_id(buf)
        if rc != 0:
            raise _capi.TTMError('ttm_comm_unique_id: %d: %s' % (rc, lib.ttm_comm_last_error().decode()))
    box = [buf.raw]
    dist.broadcast_object_list(box, src=0)
    handle = ctypes.c_void_p()
    rc = lib.ttm_comm_create(ctypes.c_char_p(box[0]), dist.get_rank(), dist.get_world_size(), ctypes.byref(handle))
    if rc != 0:
        raise _capi.TTMError('ttm_comm_create: %d: %s' % (rc, lib.ttm_comm_last_error().decode()))
    _state.update(key=key, comm=handle, lib=lib)
    return handle


def destroy():
    if _state['comm'] is not None and _state['lib'] is not None:
        _state['lib'].ttm_comm_destroy(_state['comm'])
    _state.update(key=None, comm=None, lib=None)
