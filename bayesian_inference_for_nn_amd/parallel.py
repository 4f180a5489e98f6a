"""One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU
for tests).  What shards on this path:
  * HMC / SGLD / SGD / BBB chains: independent units, no data-path collective; results are merged
    once at the end (``merge_moment_chains`` / ``merge_sampled_chains``);
  * SVGD particles: rows sharded by rank, ONE all-gather of the particle matrix per step
    (``all_gather_rows``), Jacobi sweep.
"""

from __future__ import annotations

import os
from typing import List, Sequence, Tuple


def init_distributed(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, world, local_rank)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def world_info() -> Tuple[int, int]:
    """(rank, world) of the default group, (0, 1) when torch.distributed is not initialised."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


def shared_seed(seed=None) -> int:
    """A seed every rank agrees on: the given one, else rank 0's fresh entropy broadcast to all.
    Sharded SVGD needs it (one batch permutation and one particle initialisation for all ranks:
    SVGD.py:93-111 feeds every particle the same batch); independent chains derive theirs as seed + rank."""
    s = int(seed) if seed is not None else int.from_bytes(os.urandom(6), "little")
    rank, world = world_info()
    if world > 1:
        import torch.distributed as dist
        box = [s]
        dist.broadcast_object_list(box, src=0)
        s = int(box[0])
    return s


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """(first row, row count) of this rank's contiguous shard; n_items must divide evenly."""
    if n_items % world != 0:
        raise ValueError("the item count must be a multiple of the number of ranks")
    n_local = n_items // world
    return rank * n_local, n_local


def all_gather_rows(local, out, async_op: bool = False, force_collective: bool = False):
    """out (world * n_local, D) <- concatenation of every rank's `local` (n_local, D).
    async_op (RCCL only): the gather runs on the communicator's own stream, ordered after the work already
    queued on the current stream; returns a handle whose wait() makes the current stream wait for it -- what
    lies between the call and wait() overlaps with the exchange.  Otherwise returns None (complete).
    force_collective: issue the collective in a world of ONE rank as well (a single-GPU box then exercises the RCCL
    call, its stream and the handle's wait -- tests/test_gpu_multirank.py); needs an initialised process group."""
    import torch.distributed as dist
    rank, world = world_info()
    if world == 1 and not (force_collective and dist.is_available() and dist.is_initialized()):
        if out.data_ptr() != local.data_ptr():
            out.copy_(local)
        return None
    if dist.get_backend() == "gloo":           # CPU tests / several ranks sharing one GPU: staged through the host
        parts = [out[i * local.shape[0]:(i + 1) * local.shape[0]] for i in range(world)]
        tmp = [p.detach().cpu().clone() for p in parts]
        dist.all_gather(tmp, local.detach().cpu().contiguous())
        for p, t in zip(parts, tmp):
            p.copy_(t)
        return None
    work = dist.all_gather_into_tensor(out, local.contiguous(), async_op=async_op)
    return work if async_op else None


class PeerGather:
    """The exchange step of the sharded SVGD run WITHOUT a collective library ("verify a direct
    algorithm, else a peer-write fallback"): every rank writes its rows straight into every peer's gathered matrix --
    device memory of the peers mapped into this process through IPC handles, i.e. stores over xGMI -- and then its slot of
    the peer's flag array with the step number; the consumer parks ``pyz_wait_flags`` on the stream that reads the matrix.
    One hop per step, 7 x 5 MB in parallel over the 7 links of a GPU instead of a ring's 7 dependent hops.
    Two gathered matrices are used in turn: a rank sends its rows of step s + 1 only after it holds every rank's flag of
    step s (it needs them for its own update), and a rank raises that flag only after it has finished reading step s - 1 --
    so nobody still reads the matrix that step s + 1 overwrites.  Ranks must be processes of ONE node that see the same
    device numbering (torchrun's default).  Exercised with two processes sharing one GPU (tests/test_gpu_multirank.py);
    **not yet run between two devices**: opt-in (``SVGD(gather="p2p")`` / ``PYZ_SVGD_GATHER=p2p``)."""

    def __init__(self, n_rows: int, width: int, device=None):
        import torch
        import torch.distributed as dist
        from torch.multiprocessing.reductions import reduce_tensor
        self.rank, self.world = world_info()
        if self.world > 64:
            raise ValueError("PeerGather: at most 64 ranks")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.buf = torch.zeros((2, n_rows, width), device=dev)            # the two gathered matrices, used in turn
        self.flags = torch.zeros((2, 64), dtype=torch.int64, device=dev)  # [matrix][rank]: the last step whose rows have landed
        self.fail = torch.zeros(1, dtype=torch.int32, device=dev)
        self._seq_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.comm = torch.cuda.Stream(device=dev)
        self.copied = None
        torch.cuda.synchronize()
        self.peer_buf, self.peer_flags = [self.buf] * self.world, [self.flags] * self.world
        if self.world > 1:
            mine = (reduce_tensor(self.buf), reduce_tensor(self.flags))
            everyone = [None] * self.world
            dist.all_gather_object(everyone, mine)
            self.peer_buf, self.peer_flags = [], []
            for r, ((fb, ab), (ff, af)) in enumerate(everyone):
                self.peer_buf.append(self.buf if r == self.rank else fb(*ab))
                self.peer_flags.append(self.flags if r == self.rank else ff(*af))
            dist.barrier()                                                # every rank holds every mapping before the first write

    def post(self, local, row0: int, step: int):
        """Write `local` (this rank's rows, final on the current stream) into rows [row0, row0 + n) of everyone's matrix
        step & 1, then this rank's flag.  Runs on the exchange stream; `self.copied` marks the end of its reads of `local`."""
        import torch
        b, n = step & 1, local.shape[0]
        self.comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm):
            self._seq_dev.fill_(step)
            for r in range(self.world):                                   # (own matrix first in rank order: a plain copy)
                self.peer_buf[r][b, row0:row0 + n].copy_(local, non_blocking=True)
            self.copied = self.comm.record_event()
            for r in range(self.world):
                self.peer_flags[r][b, self.rank:self.rank + 1].copy_(self._seq_dev, non_blocking=True)
        return self.buf[b]

    def wait(self, step: int, stream=None, spin_limit: int = 1 << 22):
        """Park the wait for every rank's rows of `step` on `stream` (default: the current one)."""
        import ctypes as C
        import torch
        from . import _lib
        st = stream if stream is not None else torch.cuda.current_stream()
        _lib.check(_lib.load().pyz_wait_flags(C.c_void_p(self.flags[step & 1].data_ptr()), self.world, int(step), int(spin_limit),
                                              C.c_void_p(self.fail.data_ptr()), C.c_void_p(st.cuda_stream)))

    def check(self):
        if int(self.fail.item()) != 0:
            raise RuntimeError("PeerGather: a rank's rows did not arrive (pyz_wait_flags gave up)")


def sum_over_ranks(t):
    import torch.distributed as dist
    if world_info()[1] > 1:
        if dist.get_backend() == "gloo" and t.is_cuda:
            h = t.detach().cpu()
            dist.all_reduce(h)
            t.copy_(h)
        else:
            dist.all_reduce(t)
    return t


def max_over_ranks(value: float, device=None) -> float:
    """max over ranks of a host scalar (bench.py: the step time of the slowest rank)."""
    import torch
    import torch.distributed as dist
    if world_info()[1] == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def merge_moment_chains(mean, sq_mean, n: int):
    """Pooled running moments of independent SGLD/SGD chains: sum_c n_c m_c / sum_c n_c."""
    import torch
    import torch.distributed as dist
    if world_info()[1] == 1:
        return mean, sq_mean, n
    staged = dist.get_backend() == "gloo" and mean.is_cuda      # gloo reduces host tensors
    dev = "cpu" if staged else mean.device
    w = torch.tensor([float(n)], dtype=torch.float64, device=dev)
    m, s = mean.to(dev, torch.float64) * w, sq_mean.to(dev, torch.float64) * w
    for t in (m, s, w):
        dist.all_reduce(t)
    return (m / w).to(mean.device, mean.dtype), (s / w).to(mean.device, mean.dtype), int(round(float(w.item())))


def merge_sampled_chains(samples: Sequence, frequencies: Sequence[int]) -> Tuple[List, List[int]]:
    """Concatenate the (samples, frequencies) of independent HMC chains from every rank."""
    import torch.distributed as dist
    if world_info()[1] == 1:
        return list(samples), list(frequencies)
    gathered = [None] * world_info()[1]
    dist.all_gather_object(gathered, (list(samples), list(frequencies)))
    out_s, out_f = [], []
    for s, f in gathered:
        out_s += s
        out_f += f
    return out_s, out_f
