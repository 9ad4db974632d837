"""One process per GPU: starting the ranks and the few collectives the benchmarks need.

``self_launch`` is for a command started as a single process with ``--gpus N`` (N > 1) and no
``WORLD_SIZE`` in the environment: the parent, which must not have touched the GPU, starts
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` on the same script as a CHILD
process and exits with its code (it never replaces itself: a process image must not be swapped
once anything GPU-related is loaded).  Under ``torch.distributed.run`` the ranks call
``init_ranks``, which picks RCCL (backend "nccl") with one GPU per rank.

Rehearsal (a box with fewer GPUs than ranks, never a measurement): ``UNETMI_BENCH_REHEARSAL=1``
puts every rank on device 0 and uses gloo, because RCCL refuses two ranks on one GPU.
"""

import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def needs_self_launch(n_gpus):
    return n_gpus > 1 and "WORLD_SIZE" not in os.environ


def self_launch(script, argv, n_gpus, timeout=None):
    """Run ``script argv`` as n_gpus ranks under torch.distributed.run; returns the exit code.
    The children's stdout / stderr are this process's."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    return subprocess.run(cmd, env=env, timeout=timeout).returncode


class Ranks:
    """rank / world / device of this process and the collectives of the benchmarks."""

    def __init__(self, rank, world, local_rank, backend, device):
        self.rank, self.world, self.local_rank, self.backend, self.device = rank, world, local_rank, backend, device

    def _staged(self, t):
        # gloo moves host memory; the tensors exchanged here are a few numbers, so stage them explicitly
        return t.cpu() if self.backend == "gloo" else t

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()

    def ranks_seen(self):
        """Sum over ranks of a one: the number of ranks that took part in an all-reduce on the data device."""
        if self.world == 1:
            return 1
        import torch
        import torch.distributed as dist
        one = self._staged(torch.ones(1, dtype=torch.float32, device=self.device))
        dist.all_reduce(one)
        return int(round(one.item()))

    def gather_float(self, value):
        """[value of rank 0, ..., value of rank world-1] on every rank."""
        if self.world == 1:
            return [float(value)]
        import torch
        import torch.distributed as dist
        mine = self._staged(torch.tensor([float(value)], dtype=torch.float64, device=self.device))
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine)
        return [float(t.item()) for t in out]

    def shutdown(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()


def init_ranks(expect_world, use_gpu=True):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE (set by torch.distributed.run), binds this process to its GPU and
    creates the process group.  ``use_gpu=False`` is the host-only plumbing check (gloo, no device)."""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != expect_world:
        raise SystemExit("--gpus {} but WORLD_SIZE={}".format(expect_world, world))
    rehearsal = os.environ.get("UNETMI_BENCH_REHEARSAL") == "1"
    if not use_gpu:
        device, backend = torch.device("cpu"), "gloo"
    else:
        if rehearsal:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        backend = "gloo" if rehearsal else "nccl"
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    return Ranks(rank, world, local_rank, backend if world > 1 else "none", device)
