"""Multi-GPU plumbing: one process per GPU, trajectories sharded over ranks, NO data-path collective.

Independent trajectories (Monte-Carlo replicas) are closed systems (nothing in
src/replay_no_ros.py:269-482 couples two filter instances), so the only cross-rank traffic is the
benchmark's barrier and the max-over-ranks of the elapsed time.  That control traffic goes over
torch.distributed's gloo backend on the host; RCCL/xGMI are not involved.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional


def shard_trajectories(total: int, world_size: int, rank: int) -> List[int]:
    """Contiguous block partition of trajectory ids [0, total) over ranks; sizes differ by at most 1."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad rank / world size")
    base, extra = divmod(total, world_size)
    lo = rank * base + min(rank, extra)
    return list(range(lo, lo + base + (1 if rank < extra else 0)))


def split_banks(trajectory_ids: List[int], bank_size: int = 32) -> List[List[int]]:
    """A rank's trajectories as near-equal contiguous banks of at most `bank_size`, one EkfSlam handle (own stream) each.

    The covariance pass re-reads a bank's pending factors V / W (5.2 MB per trajectory at N = 2000 and 80 ranks) once per
    row slab; up to ~32 trajectories they stay in the 256 MB Infinity Cache, beyond that the re-reads go to HBM
    (profiles/r03_pass_vs_batch.txt: 64 trajectories as one bank 175 k steps/s, as two banks of 32 driven from one host
    thread 185 k -- the second bank's solve and panel launches also run under the first bank's pass)."""
    if bank_size < 1:
        raise ValueError("bank_size must be positive")
    ids = list(trajectory_ids)
    if not ids:
        return []
    banks = -(-len(ids) // bank_size)
    base, extra = divmod(len(ids), banks)
    out, lo = [], 0
    for k in range(banks):
        hi = lo + base + (1 if k < extra else 0)
        out.append(ids[lo:hi])
        lo = hi
    return out


def run_banks(banks, first: int, count: int, slice_steps: int = 50) -> None:
    """Enqueue steps [first, first + count) of the uploaded streams of several EkfSlam handles ("banks", see
    split_banks) from ONE host thread, `slice_steps` at a time and bank after bank, so that every bank's stream is fed
    from the start: the launches of a bank are asynchronous, and the solve / panel launches of one bank run under the
    covariance pass of another (tools/two_banks.py).  Nothing is synchronised here -- call sync() on the banks."""
    if slice_steps < 1:
        raise ValueError("slice_steps must be positive")
    for k in range(first, first + count, slice_steps):
        for f in banks:
            f.stream_run(k, min(slice_steps, first + count - k))


class RankGroup:
    """Rank bookkeeping from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""

    def __init__(self, env=None):
        env = os.environ if env is None else env
        self.rank = int(env.get("RANK", "0"))
        self.local_rank = int(env.get("LOCAL_RANK", "0"))
        self.world = int(env.get("WORLD_SIZE", "1"))
        self._dist = None
        if self.world > 1:
            import torch.distributed as dist
            if "MASTER_ADDR" not in env:
                os.environ["MASTER_ADDR"] = "127.0.0.1"
            if not dist.is_initialized():
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            self._dist = dist

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def max_over_ranks(self, x: float) -> float:
        if self._dist is None:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x: float) -> float:
        if self._dist is None:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t.item())

    def gather_over_ranks(self, x: float) -> List[float]:
        """Every rank's value, in rank order, on every rank (the benchmark reports each rank's own elapsed time beside
        the maximum, so that dispatch skew between the GPUs of a node is visible)."""
        if self._dist is None:
            return [float(x)]
        import torch
        mine = torch.tensor([float(x)], dtype=torch.float64)
        out = [torch.zeros(1, dtype=torch.float64) for _ in range(self.world)]
        self._dist.all_gather(out, mine)
        return [float(t.item()) for t in out]

    def close(self):
        if self._dist is not None:
            self._dist.barrier()
            self._dist.destroy_process_group()
            self._dist = None


def timed_region(group: RankGroup, run: Callable[[], None], sync: Callable[[], None], clock) -> float:
    """barrier + device sync on both sides of `run`; returns the MAX elapsed seconds over ranks."""
    sync()
    group.barrier()
    t0 = clock()
    run()
    sync()
    dt = clock() - t0
    group.barrier()
    timed_region.last_local_seconds = dt               # this rank's own elapsed time (see RankGroup.gather_over_ranks)
    return group.max_over_ranks(dt)


def aggregate_steps_per_second(units_this_rank: float, group: RankGroup, seconds_max: float) -> float:
    """Whole-job throughput: units processed by all ranks / max-over-ranks time."""
    return group.sum_over_ranks(units_this_rank) / seconds_max
