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


_SIZES = {}  # (group id, local rows) -> rows of every rank: the all_gather of the sizes happens once


def shard_sizes(rows, device, group=None):
    """Rows of every rank's shard, gathered once per (group, local size) and cached: a per-call gather of
    the sizes plus `.item()` would put a host synchronisation on every step of the gather path."""
    key = (id(group), int(rows))
    if key not in _SIZES:
        world = dist.get_world_size(group)
        n = torch.tensor([int(rows)], device=device, dtype=torch.int64)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n, group=group)
        _SIZES[key] = [int(x.item()) for x in sizes]
    return _SIZES[key]


def all_gather_obs(obs, out=None, group=None):
    """Gather every rank's [n_r, D] observation block into one [sum n_r, D] tensor, ordered by
    global env id.  Equal shards use all_gather_into_tensor (one RCCL call); ragged shards fall
    back to all_gather on a list.  No host synchronisation after the first call."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return obs
    world = dist.get_world_size(group)
    sizes = shard_sizes(obs.shape[0], obs.device, group)
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


class P2PObsGather:
    """All-gather of the observation shards by direct peer-to-peer STORES: every rank owns
    a full-size [sum n_r, D] buffer that its peers have opened through the HIP IPC handle of its
    allocation; after a step each rank copies its [n_r, D] block straight into the matching rows of every
    peer's buffer -- G-1 independent transfers that use all xGMI links of the GPU at once (the fabric is
    point to point: 7 links per MI355X), where a ring all-gather moves the same bytes over one link per
    hop.  Reference role: none of its own -- the reference's workers never exchange env data
    (utils/mpi_tools.py:117-187 only averages gradients and statistics); this is the optional layout in
    which ONE policy consumes the whole batch.

    Synchronisation (no device-wide sync on the host): the copies run on a side stream behind the
    producer's step; each rank then records an inter-process event; a host barrier on `sync_group`
    (a gloo group: microseconds on one node) guarantees that every record is enqueued before any rank
    enqueues `wait_event` on its peers' events; the consumer's stream waits for them on the GPU.  The
    buffers are coarse-grained device memory, coherent across devices at kernel boundaries, which is
    what the event wait provides.

    On CPU tensors (the world-2 gloo tests) the peer buffers are files under /dev/shm mapped by both
    processes and the events degenerate to the barrier: same slicing and hand-shake logic, no HIP.
    """

    def __init__(self, rows, width, device, dtype=torch.float32, group=None, sync_group=None, tag="pds_p2p"):
        if not dist.is_initialized():
            raise RuntimeError("P2PObsGather needs an initialised process group")
        self.group, self.sync_group = group, sync_group if sync_group is not None else group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device(device)
        self.sizes = shard_sizes(rows, self.device if self.device.type == "cuda" else torch.device("cpu"), group)
        self.offsets = [sum(self.sizes[:r]) for r in range(self.world)]
        self.total, self.width = sum(self.sizes), int(width)
        self._step = 0
        self._files = []
        if self.device.type == "cuda":
            self.out = torch.empty(self.total, self.width, dtype=dtype, device=self.device)
            handle = self.out.untyped_storage()._share_cuda_()
            handles = [None] * self.world
            dist.all_gather_object(handles, handle, group=self.sync_group)
            self.peers = []
            for r, h in enumerate(handles):
                if r == self.rank:
                    self.peers.append(self.out)
                    continue
                st = torch.UntypedStorage._new_shared_cuda(*h)
                self.peers.append(torch.empty(0, dtype=dtype, device=st.device).set_(st, 0, (self.total, self.width)))
            self.copy_stream = torch.cuda.Stream(self.device)
            self.events = [torch.cuda.Event(interprocess=True) for _ in range(2)]
            ev_handles = [None] * self.world
            dist.all_gather_object(ev_handles, (self.device.index, [e.ipc_handle() for e in self.events]),
                                   group=self.sync_group)
            # (an IPC event is re-opened on the device it was created on: the producer's)
            self.peer_events = [None if r == self.rank else
                                [torch.cuda.Event.from_ipc_handle(torch.device("cuda", ev_handles[r][0]), hh)
                                 for hh in ev_handles[r][1]]
                                for r in range(self.world)]
        else:
            import os
            import tempfile
            nbytes = self.total * self.width
            base = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir(), f"{tag}_{os.getppid()}")
            mine = f"{base}_{self.rank}"
            self.out = torch.from_file(mine, shared=True, size=nbytes, dtype=dtype).view(self.total, self.width)
            self._files.append(mine)
            dist.barrier(group=self.sync_group)  # every file exists
            self.peers = [self.out if r == self.rank else
                          torch.from_file(f"{base}_{r}", shared=True, size=nbytes, dtype=dtype).view(self.total, self.width)
                          for r in range(self.world)]

    def gather(self, obs):
        """Store `obs` [n_r, width] into rows [offset_r, offset_r + n_r) of every rank's buffer and return
        this rank's full [total, width] buffer once all peers' blocks have arrived (stream-ordered on GPUs)."""
        lo, n = self.offsets[self.rank], self.sizes[self.rank]
        assert obs.shape == (n, self.width), (tuple(obs.shape), n, self.width)
        if self.device.type == "cuda":
            cur = torch.cuda.current_stream(self.device)
            j = self._step & 1
            self.copy_stream.wait_stream(cur)  # the step that produced `obs`
            with torch.cuda.stream(self.copy_stream):
                for k in range(self.world):  # start with the next rank so that the G ranks do not all hit one target
                    r = (self.rank + k) % self.world
                    self.peers[r][lo:lo + n].copy_(obs, non_blocking=True)
                self.events[j].record(self.copy_stream)
            dist.barrier(group=self.sync_group)  # every rank has ENQUEUED its record (host only, no device sync)
            cur.wait_event(self.events[j])
            for r in range(self.world):
                if r != self.rank:
                    cur.wait_event(self.peer_events[r][j])
        else:
            for r in range(self.world):
                self.peers[r][lo:lo + n].copy_(obs)
            dist.barrier(group=self.sync_group)
        self._step += 1
        return self.out

    def release(self):
        """Call on every rank before the buffers go away: no peer may still be writing into them."""
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        dist.barrier(group=self.sync_group)
        self.peers = []
        import os
        for f in self._files:
            try:
                os.unlink(f)
            except OSError:
                pass
        self._files = []


def max_over_ranks(value, device):
    """MAX-reduce a python float over ranks (bench.py: slowest rank defines the step time)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
