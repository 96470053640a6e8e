"""The peer-to-peer STORE all-gather of observations (sharding.P2PObsGather) on real HIP
IPC memory: two processes (gloo for the hand-shake) that share the one GPU of the test box.  The
cross-process mechanics -- IPC memory handles, inter-process events, stream-ordered copies into the
peer's buffer -- are the ones a multi-GPU node uses; only the link the bytes travel on differs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
import phoenix_drone_simulation_amd as pds
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
total = {total}
kw = dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0, seed=3, max_episode_steps=6)
env = pds.make_sharded("DroneHoverSimpleEnv-v0", total, rank=rank, world_size=world, device="cuda:0", **kw)
full = pds.make("DroneHoverSimpleEnv-v0", num_envs=total, device="cuda:0", **kw)  # what ONE process would compute
gat = pds.P2PObsGather(env.num_envs, env.obs_dim, "cuda:0")
a, b = pds.shard_range(total, rank, world)
obs, _ = env.reset()
fobs, _ = full.reset()
g = torch.Generator(device="cuda:0"); g.manual_seed(1)
for step in range(12):
    got = gat.gather(obs)
    torch.cuda.current_stream().synchronize()
    assert got.shape == (total, env.obs_dim)
    assert torch.equal(got, fobs), (rank, step, (got != fobs).nonzero()[:4])
    dist.barrier()  # nobody overwrites a buffer that a peer is still comparing
    act = -0.1 + 0.3 * torch.randn(total, 4, generator=g, device="cuda:0")
    obs = env.step(act[a:b].contiguous())[0]
    fobs = full.step(act)[0]
gat.release()
env.close(); full.close()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


# The write-after-read case: a consumer that is still READING the gathered buffer on the GPU while its peer is
# already producing later steps.  No host synchronisation inside the loop; rank 1's consumer is delayed on the
# GPU (torch.cuda._sleep) so that rank 0's stores of step s+1 / s+2 are issued while rank 1's reader of step s
# has not run yet -- the `consumed` events and the second buffer are what keeps the rows intact.
_WORKER_SLOW = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
import phoenix_drone_simulation_amd as pds
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
total, T = {total}, 10
kw = dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0, seed=3, max_episode_steps=6)
env = pds.make_sharded("DroneHoverSimpleEnv-v0", total, rank=rank, world_size=world, device="cuda:0", **kw)
full = pds.make("DroneHoverSimpleEnv-v0", num_envs=total, device="cuda:0", **kw)
gat = pds.P2PObsGather(env.num_envs, env.obs_dim, "cuda:0")
a, b = pds.shard_range(total, rank, world)
g = torch.Generator(device="cuda:0"); g.manual_seed(1)
acts = -0.1 + 0.3 * torch.randn(T, total, 4, generator=g, device="cuda:0")
want = torch.empty(T, total, env.obs_dim, device="cuda:0")
seen = torch.empty(T, total, env.obs_dim, device="cuda:0")
fobs, _ = full.reset()
for t in range(T):  # what ONE process computes, recorded up front
    want[t].copy_(fobs)
    fobs = full.step(acts[t])[0]
torch.cuda.synchronize()
dist.barrier()
obs, _ = env.reset()
side = torch.cuda.Stream()
for t in range(T):
    got = gat.gather(obs)
    if rank == 1:
        torch.cuda._sleep(40_000_000)  # ~20 ms of GPU time before the consumer reads (the policy forward stands here)
    seen[t].copy_(got)                  # the consumer: reads the whole gathered buffer, stream-ordered
    obs = env.step(acts[t, a:b].contiguous())[0]
torch.cuda.synchronize()
bad = [(t, int((seen[t] != want[t]).any(dim=1).sum())) for t in range(T) if not torch.equal(seen[t], want[t])]
assert not bad, (rank, bad)
gat.release()
env.close(); full.close()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def _run_two(script, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o[-3000:]
        assert f"rank {r} ok" in o


def test_p2p_gather_slow_consumer_reads_without_host_syncs(tmp_path):
    script = tmp_path / "worker_slow.py"
    script.write_text(_WORKER_SLOW.format(root=ROOT, total=8192))
    _run_two(script, 29733)


@pytest.mark.parametrize("total", [4096, 1000])
def test_p2p_store_gather_two_processes_one_gpu(total, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, total=total))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29610 + total % 97), WORLD_SIZE="2",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o[-3000:]
        assert f"rank {r} ok" in o
