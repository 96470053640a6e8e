"""Env sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL).

The reference parallelises by running one env per MPI rank (`mpi_fork`, utils/mpi_tools.py:47-99;
seed += 10000 * rank, algs/iwpg/iwpg.py:124-127) and never communicates env data.  Here a global
batch of `total_envs` environments is cut into contiguous blocks, one per rank; the in-kernel RNG is
keyed by the GLOBAL env id (`env_id_base + local index`), so results do not depend on the number of
ranks.  No collective is on the step path.  `all_gather_obs` is the optional exchange for the layout
in which ONE policy consumes the whole batch (works with any backend: nccl (= RCCL) on GPUs, gloo
on CPU tensors in the tests).
"""
import torch
import torch.distributed as dist


def shard_range(total_envs, rank, world_size):
    """Contiguous block [start, stop) of global env ids owned by `rank` (sizes differ by <= 1)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, rem = divmod(int(total_envs), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def make_sharded(env_id, total_envs, rank=None, world_size=None, device=None, **kwargs):
    """`make()` for this rank's shard of a `total_envs` batch; sets num_envs and env_id_base."""
    from .envs import make
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    start, stop = shard_range(total_envs, rank, world_size)
    return make(env_id, num_envs=stop - start, env_id_base=start, device=device, **kwargs)


def all_gather_obs(obs, out=None, group=None):
    """Gather every rank's [n_r, D] observation block into one [sum n_r, D] tensor, ordered by
    global env id.  Equal shards use all_gather_into_tensor (one RCCL call); ragged shards fall
    back to all_gather on a list."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return obs
    world = dist.get_world_size(group)
    n = torch.tensor([obs.shape[0]], device=obs.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    if len(set(sizes)) == 1:
        if out is None:
            out = torch.empty((sum(sizes),) + tuple(obs.shape[1:]), dtype=obs.dtype, device=obs.device)
        try:
            dist.all_gather_into_tensor(out, obs.contiguous(), group=group)
            return out
        except (RuntimeError, NotImplementedError):
            pass
    # ragged shards (sizes differ by one): pad to the largest, gather, strip the padding rows
    m = max(sizes)
    padded = torch.zeros((m,) + tuple(obs.shape[1:]), dtype=obs.dtype, device=obs.device)
    padded[:obs.shape[0]] = obs
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], 0)


def max_over_ranks(value, device):
    """MAX-reduce a python float over ranks (bench.py: slowest rank defines the step time)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
