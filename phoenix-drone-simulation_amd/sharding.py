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
    full-size [sum n_r, D] buffers that its peers have opened through the HIP IPC handle of the
    allocation; after a step each rank copies its [n_r, D] block straight into the matching rows of every
    peer's buffer -- G-1 independent transfers that use all xGMI links of the GPU at once (the fabric is
    point to point: 7 links per MI355X), where a ring all-gather moves the same bytes over one link per
    hop.  Reference role: none of its own -- the reference's workers never exchange env data
    (utils/mpi_tools.py:117-187 only averages gradients and statistics); this is the optional layout in
    which ONE policy consumes the whole batch.

    Synchronisation (no device-wide sync on the host), call number s, j = s & 1:
      * TWO buffers alternate (`out[j]`), so the blocks of step s land in a buffer nobody reads: the
        consumer of step s-1 works on `out[j ^ 1]`.
      * write-after-read across ranks: at the entry of call s this rank records the inter-process event
        `consumed[j ^ 1]` on its current stream -- every read of the buffer returned by call s-1 has been
        enqueued by then -- and a producer waits (on its copy stream, on the GPU) for the `consumed[j]`
        events its peers recorded during call s-1 before it overwrites their `out[j]`.
      * read-after-write: the copies run on a side stream behind the producer's step and are followed by
        the inter-process event `produced[j]`; the consumer's stream waits for its peers' events.
      * ONE host barrier per call on `sync_group` (a gloo group: microseconds on one node) orders the
        ENQUEUES: an event wait refers to the last record enqueued before it, so every rank's records of
        call s (and s-1) must be enqueued before any rank enqueues a wait on them.
    The buffers are coarse-grained device memory, coherent across devices at kernel boundaries, which is
    what the event waits provide.

    Validity of the result: `gather()` returns `out[j]`; it may be read by work enqueued on the current
    stream (or on streams that wait for it) until the NEXT `gather()` call on this rank -- clone what must
    live longer.

    On CPU tensors (the world-2 gloo tests) the peer buffers are files under /dev/shm mapped by both
    processes and the events degenerate to the barrier: same slicing, double buffering and hand-shake, no HIP.
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
            # one allocation, two halves: ONE IPC handle per rank
            self._both = torch.empty(2, self.total, self.width, dtype=dtype, device=self.device)
            self.bufs = [self._both[0], self._both[1]]
            handle = self._both.untyped_storage()._share_cuda_()
            handles = [None] * self.world
            dist.all_gather_object(handles, handle, group=self.sync_group)
            self.peers = []  # peers[r][j]: rank r's buffer j
            for r, h in enumerate(handles):
                if r == self.rank:
                    self.peers.append(self.bufs)
                    continue
                st = torch.UntypedStorage._new_shared_cuda(*h)
                both = torch.empty(0, dtype=dtype, device=st.device).set_(st, 0, (2, self.total, self.width))
                self.peers.append([both[0], both[1]])
            self.copy_stream = torch.cuda.Stream(self.device)
            self.produced = [torch.cuda.Event(interprocess=True) for _ in range(2)]
            self.consumed = [torch.cuda.Event(interprocess=True) for _ in range(2)]
            ev_handles = [None] * self.world
            dist.all_gather_object(ev_handles, (self.device.index, [e.ipc_handle() for e in self.produced],
                                                [e.ipc_handle() for e in self.consumed]), group=self.sync_group)

            # (an IPC event is re-opened on the device it was created on: the producer's)
            def _open(r, hs):
                return [torch.cuda.Event.from_ipc_handle(torch.device("cuda", ev_handles[r][0]), hh) for hh in hs]
            self.peer_produced = [None if r == self.rank else _open(r, ev_handles[r][1]) for r in range(self.world)]
            self.peer_consumed = [None if r == self.rank else _open(r, ev_handles[r][2]) for r in range(self.world)]
        else:
            import os
            import tempfile
            nbytes = 2 * self.total * self.width
            base = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir(), f"{tag}_{os.getppid()}")
            mine = f"{base}_{self.rank}"
            both = torch.from_file(mine, shared=True, size=nbytes, dtype=dtype).view(2, self.total, self.width)
            self.bufs = [both[0], both[1]]
            self._files.append(mine)
            dist.barrier(group=self.sync_group)  # every file exists
            self.peers = []
            for r in range(self.world):
                if r == self.rank:
                    self.peers.append(self.bufs)
                else:
                    pb = torch.from_file(f"{base}_{r}", shared=True, size=nbytes, dtype=dtype).view(2, self.total, self.width)
                    self.peers.append([pb[0], pb[1]])

    @property
    def out(self):
        """The buffer the last `gather()` returned (buffer 0 before the first call)."""
        return self.bufs[(self._step - 1) & 1] if self._step else self.bufs[0]

    def gather(self, obs):
        """Store `obs` [n_r, width] into rows [offset_r, offset_r + n_r) of every rank's buffer `j = call & 1`
        and return this rank's full [total, width] buffer j once all peers' blocks have arrived
        (stream-ordered on GPUs).  The result stays valid until the next `gather()` call."""
        lo, n = self.offsets[self.rank], self.sizes[self.rank]
        assert obs.shape == (n, self.width), (tuple(obs.shape), n, self.width)
        s, j = self._step, self._step & 1
        if self.device.type == "cuda":
            cur = torch.cuda.current_stream(self.device)
            if s >= 1:
                self.consumed[j ^ 1].record(cur)  # all reads of the previous result are enqueued before this point
            self.copy_stream.wait_stream(cur)  # the step that produced `obs` (and this rank's own reads of out[j])
            with torch.cuda.stream(self.copy_stream):
                if s >= 2:  # the peers' consumers of call s-2 (records enqueued before the barrier of call s-1)
                    for r in range(self.world):
                        if r != self.rank:
                            self.copy_stream.wait_event(self.peer_consumed[r][j])
                for k in range(self.world):  # start with the next rank so that the G ranks do not all hit one target
                    r = (self.rank + k) % self.world
                    self.peers[r][j][lo:lo + n].copy_(obs, non_blocking=True)
                self.produced[j].record(self.copy_stream)
            dist.barrier(group=self.sync_group)  # every rank has ENQUEUED its records (host only, no device sync)
            cur.wait_event(self.produced[j])
            for r in range(self.world):
                if r != self.rank:
                    cur.wait_event(self.peer_produced[r][j])
        else:
            # host copies are synchronous: a peer writes my out[j] in ITS call s, i.e. after the barrier of call
            # s-1, which I only pass once I am done with the result of call s-2 (the same buffer)
            for r in range(self.world):
                self.peers[r][j][lo:lo + n].copy_(obs)
            dist.barrier(group=self.sync_group)
        self._step += 1
        return self.bufs[j]

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
