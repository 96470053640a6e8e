"""bench.py prints ONE JSON line with the fields the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "30", "--warmup", "5",
                          "--envs-per-gpu", "65536"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "env-steps/sec" and d["unit"] == "env-steps/s" and d["n_gpus"] == 1
    assert d["steps"] == 30 and d["warmup"] == 5 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert abs(d["value"] - 65536 * 30 / (d["ms_per_step"] * 30e-3)) < 1e-6 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["value_1_thread"] > 0
    assert "65536 envs" in c["sample"]  # timed at the GPU run's N
    assert 0 < r["frac_wall"] <= r["frac"] * 1.05


def test_bench_multi_rank_code_path_on_one_gpu():
    """`bench.py --gpus 2` under torch.distributed.run, both ranks on cuda:0 (--same-device, gloo control
    plane): the sharded launch, the max-over-ranks timing, per-rank times and the P2P-store observation
    gather run end to end; the figures of such a run are meaningless and not checked."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29733", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "4",
           "--envs-per-gpu", "65536", "--same-device", "--allgather-obs", "p2p"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["per_rank_ms_per_step"]) == 2 and d["allgather_ms"] > 0
    assert "cpu_baseline" not in d and "p2p" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 65536 * 20 / (d["ms_per_step"] * 20e-3)) < 1e-6 * d["value"]


def test_bench_eight_ranks_on_one_device_p2p_gather():
    """The multi-rank path at the node's real world size: `bench.py --gpus 8`, the eight ranks on
    cuda:0 (--same-device, gloo control plane), 8 192 envs each, the peer-to-peer observation gather with HIP IPC memory
    and event handles opened 8-way (bench.py checks the gathered rows against the owners' blocks before it times).
    What stays unmeasured without an 8-GPU node: the xGMI links themselves and RCCL at 8 ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "4",
           "--envs-per-gpu", "8192", "--same-device", "--allgather-obs", "p2p", "--no-cpu-baseline", "--no-traffic"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    d = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["ranks"] == 8 and len(d["per_rank_ms_per_step"]) == 8 and d["allgather_ms"] > 0
    assert d["collective_backend"] == "gloo" and "p2p" in d["config"]["parallelism"] and d["scaling"] == "weak"
    assert abs(d["value"] - 8 * 8192 * 20 / (d["ms_per_step"] * 20e-3)) < 1e-6 * d["value"]


def test_bench_gpus_n_without_a_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (no torch.distributed.run around it): bench.py starts
    the two ranks itself as a child launcher (the role of the reference's mpi_fork, utils/mpi_tools.py:47-99),
    relays the one JSON line and the exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "4",
           "--envs-per-gpu", "65536", "--same-device", "--allgather-obs", "p2p"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["per_rank_ms_per_step"]) == 2 and d["allgather_ms"] > 0
    # a failing rank's exit code comes back through the launcher
    bad = subprocess.run(cmd[:-2] + ["--allgather-obs", "rccl"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert bad.returncode != 0


@pytest.mark.parametrize("mode,steps,warmup", [("graph", 128, 64), ("stepk", 64, 16)])
def test_bench_diagnostic_modes(mode, steps, warmup):
    """--mode graph (64 captured steps per replay) / stepk (K = 8 steps per launch): same env-steps, fewer
    launches; the JSON line keeps its shape and says which mode produced it."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warmup),
                          "--envs-per-gpu", "65536", "--mode", mode, "--no-cpu-baseline", "--no-traffic"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][0])
    assert d["steps"] == steps and d["n_gpus"] == 1 and d["roofline"]["mode"] == mode
    assert abs(d["value"] - 65536 * steps / (d["ms_per_step"] * steps * 1e-3)) < 1e-6 * d["value"]
    assert 0 < d["roofline"]["frac"] < 1
    # a step count that the mode cannot time exactly is refused, not rounded
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--envs-per-gpu", "65536",
                          "--mode", mode, "--no-cpu-baseline", "--no-traffic"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert bad.returncode != 0 and "multiples" in (bad.stderr + bad.stdout)


def test_bench_total_envs_is_strong_scaling_same_device():
    """`--total-envs` splits a fixed job into contiguous global-id blocks (SURVEY 8e strong scaling): two ranks on one
    device (gloo control plane) each take half, the line says so."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "4",
           "--total-envs", "131072", "--same-device", "--no-cpu-baseline", "--no-traffic"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["scaling"] == "strong" and d["n_gpus"] == 2 and d["ranks"] == 2
    assert d["config"]["envs_per_gpu"] == 65536 and d["total_envs"] == 131072
    assert d["visible_devices"] >= 1 and d["collective_backend"] == "gloo"


@pytest.mark.skipif(__import__("torch").cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_bench_two_gpus_rccl_allgather_for_real():
    """config (5) in small: two ranks on two devices, RCCL all-gather of the observations (bench.py checks the
    gathered rows against the owners' blocks before it times the exchange).  Skipped on the 1-GPU test boxes."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "4",
           "--envs-per-gpu", "65536", "--allgather-obs", "rccl", "--no-cpu-baseline", "--no-traffic"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["collective_backend"] == "nccl" and d["visible_devices"] >= 2
    assert d["allgather_ms"] > 0 and d["scaling"] == "weak"


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def test_bench_one_rank_on_rccl_force_dist():
    """RCCL loaded and executed at last: `bench.py --force-dist --allgather-obs rccl` under
    torch.distributed.run with ONE rank -- `init_process_group("nccl", device_id=...)`, the barriers and the all_gather of the
    max-over-ranks timing, and `all_gather_into_tensor` of the observations after every step (checked against the owner's
    block) run on this ROCm's librccl.  Says nothing about xGMI or N > 1."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29741", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "4",
           "--envs-per-gpu", "65536", "--force-dist", "--allgather-obs", "rccl", "--no-traffic", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=_clean_env())
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["collective_backend"] == "nccl" and d["n_gpus"] == 1 and d["ranks"] == 1
    assert d["allgather_ms"] > 0 and len(d["per_rank_ms_per_step"]) == 1 and "rccl" in d["config"]["parallelism"]


_RCCL_TRAINER_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
import phoenix_drone_simulation_amd as pds
import phoenix_drone_simulation_amd.ppo as ppo
dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
calls = dict(all_reduce=0, broadcast=0)
for name in calls:
    def wrap(fn, name=name):
        def f(*a, **k):
            calls[name] += 1
            assert a[0].is_cuda
            return fn(*a, **k)
        return f
    setattr(dist, name, wrap(getattr(dist, name)))
ppo.FORCE_COLLECTIVES = True
res = {{}}
for fused in (True, False):
    env = pds.make_sharded("DroneHoverSimpleEnv-v0", 2048, rank=0, world_size=1, device=dev, seed=7)
    tr = ppo.PPOTrainer(env, rollout_len=16, epochs=4, train_pi_iterations=4, train_v_iterations=2, num_mini_batches=4, seed=7, fused=fused)
    before = dict(calls)
    info = tr.learn_one_epoch()
    torch.cuda.synchronize()
    n_ar = calls["all_reduce"] - before["all_reduce"]
    # 4 policy + 8 value gradient averages, 4 running-statistics reductions (mean / var of obs and returns), the NaN flag, the stats
    assert n_ar >= 4 + 8 + 4 + 2, (fused, n_ar)
    res[fused] = (info["loss_pi"], info["loss_v"], info["ep_len"])
    env.close()
assert calls["broadcast"] > 0  # sync_params
# the collectives over one rank change nothing: same epoch without them
ppo.FORCE_COLLECTIVES = False
env = pds.make_sharded("DroneHoverSimpleEnv-v0", 2048, rank=0, world_size=1, device=dev, seed=7)
tr = ppo.PPOTrainer(env, rollout_len=16, epochs=4, train_pi_iterations=4, train_v_iterations=2, num_mini_batches=4, seed=7, fused=True,
                    overlap_value_update=False)
info = tr.learn_one_epoch()
assert abs(info["loss_pi"] - res[True][0]) < 1e-6 and abs(info["loss_v"] - res[True][1]) < 1e-5 * abs(info["loss_v"]) and info["ep_len"] == res[True][2], (info, res)
dist.barrier()
dist.destroy_process_group()
print("rccl trainer ok", calls)
"""


def test_trainer_epoch_with_the_collectives_on_rccl_one_rank(tmp_path):
    """One `PPOTrainer.learn_one_epoch()` per path (fused kernels / PyTorch ops) with every collective of the multi-GPU trainer
    executed on the `nccl` (= RCCL) backend over one rank: sync_params broadcasts, the flattened gradient all-reduce of both
    networks, the running-statistics all-reduces, the NaN-flag and episode-statistics reductions (utils/mpi_tools.py:30-44,
    utils/online_mean_std.py:54-95).  Over one rank they are identities: the epoch's losses equal a run without them."""
    script = tmp_path / "rccl_trainer.py"
    script.write_text(_RCCL_TRAINER_WORKER.format(root=ROOT))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29743", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=_clean_env())
    assert out.returncode == 0 and "rccl trainer ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
