"""The library's own RCCL communicator (include/unetmi.h: unetmi_comm_*, unetmi_allreduce_f32) from Python.

``init_from_torch_distributed`` is for a process that already runs under ``torch.distributed.run``: rank 0 asks
the library for an id, the 128 bytes travel through the existing process group (any backend, host tensors), and
every rank joins with its GPU current.  After that ``all_reduce_sum_(t)`` is one ncclAllReduce enqueued on the
current stream, with no torch.distributed call on the data path.
"""

import ctypes

import torch

from . import _lib
from ._lib import check

ID_BYTES = 128


def unique_id():
    buf = ctypes.create_string_buffer(ID_BYTES)
    check(_lib.lib.unetmi_comm_unique_id(buf), "unetmi_comm_unique_id")
    return bytes(buf.raw)


def init(rank, world, id_bytes, device):
    if len(id_bytes) != ID_BYTES:
        raise ValueError("the communicator id is {} bytes".format(ID_BYTES))
    with torch.cuda.device(device):
        check(_lib.lib.unetmi_comm_init(rank, world, ctypes.c_char_p(id_bytes)), "unetmi_comm_init")


def init_from_torch_distributed(device, group=None):
    """Creates the communicator over the ranks of an initialised torch.distributed group (or a single-rank one
    when torch.distributed is not initialised).  Returns the world size."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        init(0, 1, unique_id(), device)
        return 1
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    init(rank, world, box[0], device)
    return world


def world():
    return _lib.lib.unetmi_comm_world()


def all_reduce_sum_(t):
    """In-place sum over the communicator's ranks of a contiguous fp32 device tensor, on the current stream."""
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError("all_reduce_sum_: contiguous fp32 device tensor expected")
    with torch.cuda.device(t.device):
        check(_lib.lib.unetmi_allreduce_f32(t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream),
              "unetmi_allreduce_f32")
    torch.autograd.graph.increment_version(t)
    return t


def destroy():
    check(_lib.lib.unetmi_comm_destroy(), "unetmi_comm_destroy")
