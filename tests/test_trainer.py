"""The caller of the hot path: on-device PPO pieces against golden
vectors produced by the reference's own trainer classes (oracle/refgen/gen_golden_trainer.py):
OnlineMeanStd, ActorCritic forward (state_dict interchange), PPO-clip / value losses, exploration
noise schedule on CPU; the HIP GAE kernel and a short end-to-end learning run on the GPU."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trainer.npz")
import golden_util as gu  # noqa: E402


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def test_online_mean_std_matches_reference(g):
    from phoenix_drone_simulation_amd.ppo import OnlineMeanStd
    oms = OnlineMeanStd(shape=(42,))
    for i in range(3):
        oms.update(torch.from_numpy(g[f"oms_batch{i}"]))
        np.testing.assert_allclose(oms.mean.numpy(), g[f"oms_mean{i}"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(oms.std.numpy(), g[f"oms_std{i}"], rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(oms.count.numpy(), g[f"oms_count{i}"])
    x = torch.from_numpy(g["oms_fwd_in"])
    np.testing.assert_allclose(oms(x).numpy(), g["oms_fwd_out"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(oms(x, subtract_mean=False, clip=True).numpy(), g["oms_fwd_clip"], rtol=1e-6, atol=1e-6)
    r = OnlineMeanStd(shape=(1,))
    for i in range(2):
        r.update(torch.from_numpy(g[f"roms_batch{i}"]))
        np.testing.assert_allclose(r.mean.numpy(), g[f"roms_mean{i}"], rtol=1e-6)
        np.testing.assert_allclose(r.std.numpy(), g[f"roms_std{i}"], rtol=1e-6)


def _load_ac(g):
    from phoenix_drone_simulation_amd.ppo import ActorCritic
    ac = ActorCritic(42, 4)
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd__")}
    assert set(sd) == set(ac.state_dict())  # the reference's state_dict keys, one for one
    ac.load_state_dict(sd)
    return ac


def test_actor_critic_forward_and_losses_match_reference(g):
    from phoenix_drone_simulation_amd.ppo import ppo_loss, value_loss
    ac = _load_ac(g)
    ac.eval()
    a, v, _ = ac.step(torch.from_numpy(g["ac_obs"]))
    np.testing.assert_allclose(a.numpy(), g["ac_mean_action"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v.numpy(), g["ac_value"], rtol=1e-5, atol=1e-6)
    obs = ac.obs_oms(torch.from_numpy(g["ac_obs"]))
    data = dict(obs=obs, act=torch.from_numpy(g["loss_act"]), log_p=torch.from_numpy(g["loss_old_logp"]),
                adv=torch.from_numpy(g["loss_adv"]))
    with torch.no_grad():
        _, logp = ac.pi(obs, data["act"])
        loss, info = ppo_loss(ac, data, clip_ratio=0.2, entropy_coef=0.0)
        lv = value_loss(ac, obs, torch.from_numpy(g["loss_target_v"]))
    np.testing.assert_allclose(logp.numpy(), g["loss_logp_act"], rtol=1e-5, atol=1e-5)
    assert float(loss) == pytest.approx(float(g["loss_pi"]), rel=1e-5)
    assert float(lv) == pytest.approx(float(g["loss_v"]), rel=1e-5)
    assert float(info["kl"]) == pytest.approx(float(g["loss_kl"]), rel=1e-5)
    assert float(info["ent"]) == pytest.approx(float(g["loss_ent"]), rel=1e-5)
    assert float(info["ratio"]) == pytest.approx(float(g["loss_ratio"]), rel=1e-5)


def test_exploration_noise_schedule(g):
    from phoenix_drone_simulation_amd.ppo import ActorCritic
    ac = ActorCritic(42, 4)
    for f, s in zip(g["anneal_frac"], g["anneal_std"]):
        ac.pi.set_log_std(float(f))
        assert float(torch.exp(ac.pi.log_std[0])) == pytest.approx(float(s), rel=1e-6)


def _gae_numpy(rew, val, term, trunc, fval, last_val, gamma, lam, scale, clip):
    """Host restatement of Buffer.finish_path per path (algs/core.py:461-533) for [T, N] arrays."""
    T, N = rew.shape
    adv, tv, dr = np.zeros_like(rew), np.zeros_like(rew), np.zeros_like(rew)
    for n in range(N):
        start = 0
        for t in range(T):
            end = term[t, n] or trunc[t, n] or t == T - 1
            if not end:
                continue
            b = fval[t, n] if trunc[t, n] else (0.0 if term[t, n] else last_val[n])  # (cut wins: iwpg.py:374-379)
            r = np.append(rew[start:t + 1, n], b).astype(np.float64)
            v = np.append(val[start:t + 1, n], b).astype(np.float64)
            disc = np.zeros(len(r)); acc = 0.0
            for k in range(len(r) - 1, -1, -1):
                acc = r[k] + gamma * acc; disc[k] = acc
            rs = np.clip(r * scale, -clip, clip) if scale > 0 else r
            deltas = rs[:-1] + gamma * v[1:] - v[:-1]
            a = np.zeros(len(deltas)); acc = 0.0
            for k in range(len(deltas) - 1, -1, -1):
                acc = deltas[k] + gamma * lam * acc; a[k] = acc
            adv[start:t + 1, n], tv[start:t + 1, n], dr[start:t + 1, n] = a, a + v[:-1], disc[:-1]
            start = t + 1
    return adv, tv, dr


@pytest.mark.gpu
def test_gae_kernel_matches_reference_buffer(g):
    """pds_gae on the exact trajectories the reference Buffer processed (4 paths: terminated,
    bootstrapped one-step path, ...), with and without reward scaling."""
    from phoenix_drone_simulation_amd.ppo import gae
    dev = torch.device("cuda")
    for tag in ("plain", "scaled"):
        rew, val = g[f"buf_{tag}_rew"], g[f"buf_{tag}_val"]
        lens, lvs = g[f"buf_{tag}_lens"], g[f"buf_{tag}_last_vals"]
        T = len(rew)
        term = np.zeros(T, np.uint8); trunc = np.zeros(T, np.uint8); fval = np.zeros(T, np.float32)
        k = 0
        for L, lv in zip(lens, lvs):
            k += int(L)
            if lv == 0.0:
                term[k - 1] = 1
            else:
                trunc[k - 1] = 1; fval[k - 1] = lv
        scale = float(1.0 / (g["buf_ret_std"][0] + 1e-5)) if tag == "scaled" else 0.0
        col = lambda x, dt=torch.float32: torch.tensor(x, dtype=dt, device=dev).reshape(T, 1)  # noqa: E731
        adv, tv, dr = gae(col(rew), col(val), col(term, torch.uint8), col(trunc, torch.uint8), col(fval),
                          torch.zeros(1, device=dev), 0.99, 0.95, scale, 10.0)
        np.testing.assert_allclose(adv.cpu().numpy()[:, 0], g[f"buf_{tag}_adv"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(tv.cpu().numpy()[:, 0], g[f"buf_{tag}_target_v"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(dr.cpu().numpy()[:, 0], g[f"buf_{tag}_discounted_ret"], rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
def test_gae_kernel_random_rollout():
    from phoenix_drone_simulation_amd.ppo import gae
    rs = np.random.RandomState(3)
    T, N = 37, 300
    rew = rs.normal(-1, 2, (T, N)).astype(np.float32); val = rs.normal(-3, 2, (T, N)).astype(np.float32)
    term = (rs.rand(T, N) < 0.05).astype(np.uint8); trunc = ((rs.rand(T, N) < 0.03) & (term == 0)).astype(np.uint8)
    fval = rs.normal(-3, 2, (T, N)).astype(np.float32); last = rs.normal(-3, 2, N).astype(np.float32)
    dev = torch.device("cuda")
    for scale in (0.0, 0.37):
        out = gae(*(torch.tensor(x, device=dev) for x in (rew, val, term, trunc, fval, last)), 0.99, 0.95, scale, 10.0)
        ref = _gae_numpy(rew, val, term, trunc, fval, last, 0.99, 0.95, scale, 10.0)
        for a, b in zip(out, ref):
            np.testing.assert_allclose(a.cpu().numpy(), b, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_ppo_learns_hover_end_to_end(fused):
    """Learning-curve check: PPO on DroneHoverSimpleEnv-v0 (the reference's default env config: sensor
    noise, 10 % DR, thrust noise) improves the mean episode return and length within a few epochs --
    with the fused MFMA kernels (csrc/pds_mlp.hip, the default on the GPU) and with the PyTorch op
    chains they replace."""
    import phoenix_drone_simulation_amd as pds
    from phoenix_drone_simulation_amd.ppo import PPOTrainer
    env = pds.make("DroneHoverSimpleEnv-v0", num_envs=2048, seed=1)
    tr = PPOTrainer(env, rollout_len=64, epochs=12, train_pi_iterations=40, seed=1, fused=fused)
    assert tr.fused == fused
    tr.learn()
    first, last = tr.log[0], tr.log[-1]
    assert all(np.isfinite(e["loss_pi"]) and np.isfinite(e["loss_v"]) for e in tr.log)
    assert last["ep_len"] > 1.5 * first["ep_len"], (first, last)
    assert last["ep_ret"] / last["ep_len"] > first["ep_ret"] / first["ep_len"], (first, last)
    assert last["noise_std"] < first["noise_std"]
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("task,H,ac_kwargs", [
    ("DroneCircleSimpleEnv-v0", 2, None),
    # round 6: 8 192 envs x 16 steps = 131 072 samples -- the size from which the policy gradient runs five of its GEMMs on split-bf16 MFMAs
    # (csrc/pds_mlp.hip PDS_SPLIT_BF16; "big": num_envs 8 192 instead of 1 024)
    ("DroneHoverSimpleEnv-v0", 2, "big"),
    # round 6: more than 64 network inputs (csrc/pds_mlp_wide.hip) -- the widths and layer sizes of the reference's
    # experiments/04_history_of_state_action_inputs/04_train_with_history.py:34-42 (H = 4 / 6 / 8; policy 32-32 / 48-48 / 64-64)
    ("DroneHoverSimpleEnv-v0", 4, None),
    ("DroneCircleSimpleEnv-v0", 4, {"pi": {"hidden_sizes": (32, 32), "activation": "relu"}, "val": {"hidden_sizes": (64, 64), "activation": "tanh"}}),
    ("DroneCircleSimpleEnv-v0", 6, {"pi": {"hidden_sizes": (48, 48), "activation": "relu"}, "val": {"hidden_sizes": (64, 64), "activation": "tanh"}}),
    ("DroneHoverSimpleEnv-v0", 8, None),
    ("DroneCircleSimpleEnv-v0", 8, {"pi": {"hidden_sizes": (64, 64), "activation": "relu"}, "val": {"hidden_sizes": (64, 64), "activation": "tanh"}}),
    ("DroneTakeOffSimpleEnv-v0", 8, None),
])
def test_fused_update_matches_the_autograd_update(task, H, ac_kwargs):
    """One policy iteration and one value mini-batch on real rollout data: the fused kernels write the
    same gradients into .grad as loss.backward() of the PyTorch path (ppo_loss / value_loss)."""
    import phoenix_drone_simulation_amd as pds
    from phoenix_drone_simulation_amd.ppo import PPOTrainer, gae, ppo_loss, value_loss
    num_envs = 8192 if ac_kwargs == "big" else 1024
    ac_kwargs = None if ac_kwargs == "big" else ac_kwargs
    env = pds.make(task, num_envs=num_envs, seed=2, observation_history_size=H)
    assert env.obs_dim == {"DroneHoverSimpleEnv-v0": 17, "DroneCircleSimpleEnv-v0": 20, "DroneTakeOffSimpleEnv-v0": 24}[task] * H
    tr = PPOTrainer(env, rollout_len=16, epochs=4, seed=2, fused=True, ac_kwargs=ac_kwargs)
    tr.roll_out()
    ac, T, N = tr.ac, tr.T, tr.N
    adv, target_v, _ = gae(tr.rew_buf, tr.val_buf, tr.term_buf, tr.trunc_buf, tr.fval_buf, tr.last_val, 0.99, 0.95, 0.0, 10.0)
    obs = ac.obs_oms(tr.obs_buf.reshape(T * N, -1)).contiguous()
    data = dict(obs=obs, act=tr.act_buf.reshape(T * N, -1), adv=adv.reshape(-1), log_p=tr.logp_buf.reshape(-1) + 0.1,
                target_v=target_v.reshape(-1))
    # the rollout's own value / log-prob bookkeeping agrees with the PyTorch modules
    with torch.no_grad():
        d, logp = ac.pi(obs, data["act"])
        assert torch.allclose(logp, tr.logp_buf.reshape(-1), atol=2e-4)
        assert torch.allclose(ac.v(obs), tr.val_buf.reshape(-1), atol=1e-4)
    tr.fm_pi.ppo_grad(obs, data["act"].contiguous(), data["adv"].contiguous(), data["log_p"].contiguous(), ac.pi.log_std, 0.2)
    got_pi = tr.fm_pi.flat_grad.clone()
    idx = torch.randperm(T * N, device=obs.device)[:1000]
    tr.fm_v.value_grad(obs, data["target_v"].contiguous(), index=idx)
    got_v = tr.fm_v.flat_grad.clone()
    for p in list(ac.pi.net.parameters()) + list(ac.v.parameters()):
        p.grad = None
    ppo_loss(ac, data, 0.2, 0.0)[0].backward()
    value_loss(ac, obs[idx], data["target_v"][idx]).backward()
    want_pi = torch.cat([p.grad.reshape(-1) for p in ac.pi.net.parameters()])
    want_v = torch.cat([p.grad.reshape(-1) for p in ac.v.parameters()])
    assert torch.allclose(got_pi, want_pi, rtol=1e-3, atol=1e-6 * max(1.0, float(want_pi.abs().max())))
    assert torch.allclose(got_v, want_v, rtol=1e-3, atol=1e-6 * max(1.0, float(want_v.abs().max())))
    env.close()


_DDP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from phoenix_drone_simulation_amd.ppo import avg_grads, OnlineMeanStd, ActorCritic
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(0)
ac = ActorCritic(42, 4)
for i, p in enumerate(ac.pi.net.parameters()):
    p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
avg_grads(ac.pi.net)     # mpi_avg_grads: mean over ranks, one flattened all-reduce
for i, p in enumerate(ac.pi.net.parameters()):
    assert torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1))), (rank, i)
# OnlineMeanStd over two ranks == single-process update on the concatenated batch
torch.manual_seed(1)
full = torch.randn(64, 42) * 3 + 1
oms = OnlineMeanStd(shape=(42,)); oms.update(full[rank * 32:(rank + 1) * 32])
ref = OnlineMeanStd(shape=(42,))
dist.destroy_process_group()
ref.update(full)
assert torch.allclose(oms.mean, ref.mean, atol=1e-5) and torch.allclose(oms.std, ref.std, atol=1e-4), rank
assert float(oms.count) == 64.0
print("rank", rank, "ok")
"""


def test_gradient_averaging_and_running_stats_world2_gloo(tmp_path):
    """The N>1 path of the trainer on CPU: 2 processes over gloo (the reference's mpi_avg_grads /
    MPI-averaged OnlineMeanStd, utils/mpi_tools.py:30-36, utils/online_mean_std.py:76-83)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(_DDP_WORKER.format(root=root))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


@pytest.mark.gpu
def test_save_checkpoint_roundtrip(tmp_path):
    """save_checkpoint writes a reference-layout model.pt and the firmware JSON; both reload and drive
    the same deterministic actions as the trained ActorCritic."""
    import phoenix_drone_simulation_amd as pds
    from phoenix_drone_simulation_amd.policy_io import load_network_json
    from phoenix_drone_simulation_amd.ppo import ActorCritic, PPOTrainer
    env = pds.make("DroneCircleSimpleEnv-v0", num_envs=512, seed=4)
    tr = PPOTrainer(env, rollout_len=16, epochs=3, train_pi_iterations=5, seed=4)
    tr.learn(2)
    path = tr.save_checkpoint(str(tmp_path))
    tr.write_progress_csv(str(tmp_path / "progress.csv"))
    rows = open(tmp_path / "progress.csv").read().strip().split("\n")
    assert rows[0].startswith("Epoch,EpRet/Mean,EpLen/Mean,Loss/Pi,Loss/Value") and len(rows) == 3
    assert rows[2].split(",")[0] == "2" and float(rows[2].split(",")[10]) == 2 * 16 * 512
    sd = torch.load(path, map_location="cpu", weights_only=True)
    assert set(sd) == set(tr.ac.state_dict())
    ac2 = ActorCritic.from_reference_state_dict(sd).to(env.device)
    pol = load_network_json(str(tmp_path / "model.json")).to(env.device)
    obs, _ = env.reset()
    tr.ac.eval(); ac2.eval()
    a1 = tr.ac.step(obs)[0]
    assert torch.allclose(ac2.step(obs)[0], a1, atol=1e-6)
    assert torch.allclose(pol(obs), a1, atol=1e-5)
    env.close()


@pytest.mark.gpu
def test_graph_captured_rollout_equals_eager_rollout_bitwise():
    """The whole rollout (env step, actor + critic inference, sampling, bookkeeping: 7 launches per step)
    captured into ONE hipGraph and replayed per epoch gives bit for bit what the eager loop gives, epoch
    after epoch (updates in between): the env's tick / action-ring parity and the sampler's call counter
    live in device memory, the running statistics and the exploration noise are updated in place."""
    import phoenix_drone_simulation_amd as pds
    from phoenix_drone_simulation_amd.ppo import PPOTrainer
    mk = lambda g: PPOTrainer(pds.make("DroneHoverSimpleEnv-v0", num_envs=1024, seed=3, max_episode_steps=40),
                              rollout_len=16, epochs=6, train_pi_iterations=4, train_v_iterations=2, seed=3, fused=True,
                              graph_rollout=g)
    a, b = mk(True), mk(False)
    assert a.graph_rollout and not b.graph_rollout
    for ep in range(4):
        for tr in (a, b):
            tr.ac.update(frac=ep / 6)
        sa, sb = a.roll_out(), b.roll_out()
        torch.cuda.synchronize()
        for name in ("obs_buf", "act_buf", "logp_buf", "val_buf", "fval_buf", "rew_buf", "term_buf", "trunc_buf"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (ep, name)
        assert torch.equal(a.last_val, b.last_val) and torch.equal(a.obs, b.obs), ep
        # finished-episode statistics: block partials are added with float atomics (order varies run to run)
        assert float(sa[2]) == float(sb[2]) > 0 and float(sa[1]) == float(sb[1]), ep
        assert abs(float(sa[0]) - float(sb[0])) <= 1e-5 * abs(float(sb[0])), ep
        torch.manual_seed(100 + ep)  # the value mini-batches are drawn from torch's global generator
        ia = a.update()
        torch.manual_seed(100 + ep)
        ib = b.update()
        assert ia["loss_pi"] == ib["loss_pi"] and ia["loss_v"] == ib["loss_v"], (ep, ia, ib)
    assert a._sample_calls == b._sample_calls == 64
    assert a.env.sync_tick() == b.env.tick
    a.env.close(); b.env.close()


@pytest.mark.gpu
def test_value_update_on_a_second_stream_equals_the_sequential_update_bitwise():
    """PPOTrainer(overlap_value_update=True) runs the value net's mini-batch steps on a second HIP stream next to the
    policy net's full-batch steps: same launches on disjoint state, so every parameter and both optimiser states come
    out bit for bit as from the sequential order, epoch after epoch."""
    import phoenix_drone_simulation_amd as pds
    from phoenix_drone_simulation_amd.ppo import PPOTrainer
    mk = lambda o: PPOTrainer(pds.make("DroneHoverSimpleEnv-v0", num_envs=2048, seed=5), rollout_len=16, epochs=5,
                              train_pi_iterations=12, train_v_iterations=3, seed=5, fused=True, overlap_value_update=o)
    a, b = mk(True), mk(False)
    for ep in range(3):
        ia, ib = a.learn_one_epoch(), b.learn_one_epoch()
        torch.cuda.synchronize()
        assert a._side_stream is not None and b._side_stream is None
        for (ka, pa), (kb, pb) in zip(a.ac.state_dict().items(), b.ac.state_dict().items()):
            assert ka == kb and torch.equal(pa, pb), (ep, ka)
        for oa, ob in ((a.pi_opt, b.pi_opt), (a.vf_opt, b.vf_opt)):
            for sa, sb in zip(oa.state.values(), ob.state.values()):
                for k in sa:
                    assert torch.equal(torch.as_tensor(sa[k]), torch.as_tensor(sb[k])), (ep, k)
        assert ia["loss_pi"] == ib["loss_pi"] and ia["loss_v"] == ib["loss_v"], (ep, ia, ib)
    a.env.close(); b.env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("task,kw,n", [
    ("DroneHoverSimpleEnv-v0", dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0), 512),
    ("DroneHoverSimpleEnv-v0", dict(), 1000),                                   # reference defaults, ragged last tile
    ("DroneCircleSimpleEnv-v0", dict(use_motor_dynamics=True), 256),            # PT1 + DR + noise
    ("DroneTakeOffSimpleEnv-v0", dict(), 192),
    ("DroneTakeOffSimpleEnv-v0", dict(use_ground_effect=True), 200),            # round 6: the ground-effect extension (BASELINE config 4)
    ("DroneTakeOffSimpleEnv-v0", dict(use_ground_effect=True, observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0), 64 * 259),
    # round 4: the PID control modes (what the reference's exp-07 trains, 4 / 8 physics sub-steps) and the latency ring
    ("DroneHoverSimpleEnv-v0", dict(control_mode="AttitudeRate", aggregate_phy_steps=4), 320),
    ("DroneCircleSimpleEnv-v0", dict(control_mode="Attitude", aggregate_phy_steps=2, use_motor_dynamics=True), 192),
    ("DroneHoverSimpleEnv-v0", dict(use_latency=True, latency=0.02), 256),
    ("DroneHoverSimpleEnv-v0", dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0, use_latency=True,
                                    latency=0.035, control_mode="PWM"), 130),
    # more tiles than CUs: two tiles (teams) per block; an odd tile count leaves the last block with one
    ("DroneHoverSimpleEnv-v0", dict(), 64 * 313 - 7),
    ("DroneCircleSimpleEnv-v0", dict(control_mode="AttitudeRate", aggregate_phy_steps=2, use_motor_dynamics=True), 64 * 258),
    ("DroneTakeOffSimpleEnv-v0", dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0), 64 * 300),
    # round 5: what else the reference trains -- the Kalman hold (observation_frequency < sim_freq, envs/hover.py:134-156),
    # partial noise settings (the ctor arguments are independent, envs/base.py:26-48), the latency ring with a PID mode
    ("DroneHoverSimpleEnv-v0", dict(observation_frequency=50), 200),
    ("DroneCircleSimpleEnv-v0", dict(observation_frequency=50, aggregate_phy_steps=2, use_motor_dynamics=True,
                                     domain_randomization=-1, motor_thrust_noise=0), 130),
    ("DroneTakeOffSimpleEnv-v0", dict(observation_frequency=25), 100),
    ("DroneHoverSimpleEnv-v0", dict(motor_thrust_noise=0), 200),                                   # sensor noise + DR only
    ("DroneHoverSimpleEnv-v0", dict(observation_noise=-1), 200),                                   # DR + thrust noise only
    ("DroneCircleSimpleEnv-v0", dict(domain_randomization=-1, use_motor_dynamics=True), 130),      # both noises, no DR
    ("DroneCircleSimpleEnv-v0", dict(observation_noise=-1, motor_thrust_noise=0), 130),            # DR only
    ("DroneTakeOffSimpleEnv-v0", dict(domain_randomization=-1, motor_thrust_noise=0), 100),        # sensor noise only
    ("DroneHoverSimpleEnv-v0", dict(observation_noise=-1, domain_randomization=-1), 64 * 260),     # thrust noise only, two teams
    ("DroneHoverSimpleEnv-v0", dict(use_latency=True, latency=0.02, control_mode="AttitudeRate", aggregate_phy_steps=2), 200),
    ("DroneCircleSimpleEnv-v0", dict(use_latency=True, latency=0.03, control_mode="Attitude", aggregate_phy_steps=2,
                                     use_motor_dynamics=True, observation_noise=-1, domain_randomization=-1,
                                     motor_thrust_noise=0), 130),
    # round 6: observation histories other than 2 (pds_rollout_history, csrc/pds_rollout_hist.h: actor + env + history update in
    # the kernel, the critic in one pass after it) -- every input-tile count the kernel is built for (<= 64 / 96 / 128 / 192
    # inputs), H = 1 (no shift), 16-byte aligned halves (Circle 20, TakeOff 24) and unaligned ones (Hover 17)
    ("DroneHoverSimpleEnv-v0", dict(observation_history_size=4), 200),                                              # 68 inputs
    ("DroneCircleSimpleEnv-v0", dict(observation_history_size=8, use_motor_dynamics=True), 130),                    # 160
    ("DroneTakeOffSimpleEnv-v0", dict(observation_history_size=8, observation_noise=-1, domain_randomization=-1,
                                      motor_thrust_noise=0), 100),                                                  # 192
    ("DroneHoverSimpleEnv-v0", dict(observation_history_size=1, observation_noise=-1, domain_randomization=-1,
                                    motor_thrust_noise=0), 130),                                                    # 17
    ("DroneCircleSimpleEnv-v0", dict(observation_history_size=3), 64 * 5 - 3),                                      # 60
    ("DroneHoverSimpleEnv-v0", dict(observation_history_size=6, use_motor_dynamics=True, observation_noise=-1,
                                    domain_randomization=-1, motor_thrust_noise=0), 64 * 300),                      # 102, > 256 tiles
    ("DroneCircleSimpleEnv-v0", dict(observation_history_size=6), 70),                                              # 120
])
def test_fused_rollout_equals_per_step_rollout_bitwise(task, kw, n):
    """pds_rollout (ONE launch for the T closed-loop steps: both networks on the matrix cores, Gaussian sampling, env
    step with the state in registers, V(final_obs), episode bookkeeping; csrc/pds_rollout.h) against the per-step
    kernels of round 2 (7 launches per step): every rollout buffer bit for bit, over two consecutive rollouts with an
    update of the running statistics in between, short episodes so that resets and TimeLimit truncations occur."""
    import phoenix_drone_simulation_amd as pds
    from phoenix_drone_simulation_amd.ppo import PPOTrainer
    T = 12
    tr = []
    for fused_rollout in (True, False):
        env = pds.make(task, num_envs=n, seed=3, max_episode_steps=9, **kw)
        tr.append(PPOTrainer(env, rollout_len=T, epochs=4, seed=5, fused=True, graph_rollout=False, fused_rollout=fused_rollout))
    a, b = tr
    for rnd in range(2):
        sa, sb = a.roll_out(), b.roll_out()
        torch.cuda.synchronize()
        assert a.fused_rollout is True and b.fused_rollout is False
        for name in ("obs_buf", "act_buf", "logp_buf", "val_buf", "rew_buf", "term_buf", "trunc_buf", "ep_ret", "ep_len", "last_val"):
            x, y = getattr(a, name), getattr(b, name)
            assert torch.equal(x, y), (rnd, name, (x != y).nonzero()[:4])
        assert torch.equal(a.obs, b.obs)
        if kw.get("observation_history_size", 2) != 2:  # the env's own history is what the next step() would start from
            assert torch.equal(a.env._hist.reshape(n, -1), b.env._hist.reshape(n, -1))
        done = (a.term_buf | a.trunc_buf).bool()
        assert int(done.sum()) >= n  # max_episode_steps = 9 < T
        # V(final_obs) where pds_gae reads it: episodes the TimeLimit cut, terminated on that step or not (one that only
        # terminated bootstraps with 0, and the one-launch rollout does not evaluate its row) -- and, for the caller that
        # mirrors the reference's epoch-end cut (reset_each_rollout), every env that finished on the LAST step
        cut = a.trunc_buf.bool().clone()  # (includes `terminated AND cut` where a config produces it)
        cut[T - 1] |= a.term_buf[T - 1].bool()
        assert int(cut.sum()) > 0 and torch.equal(a.fval_buf[cut], b.fval_buf[cut])
        # ... so what the update sees is the same, bit for bit: advantages, value targets, discounted returns
        from phoenix_drone_simulation_amd.ppo import gae
        ga, gb = (gae(t_.rew_buf, t_.val_buf, t_.term_buf, t_.trunc_buf, t_.fval_buf, t_.last_val, 0.99, 0.95, 0.37, 10.0) for t_ in tr)
        for x, y in zip(ga, gb):
            assert torch.equal(x, y)
        torch.testing.assert_close(sa, sb, rtol=1e-5, atol=1e-3)  # (sums over all envs: atomics, order differs)
        for f in ("pos", "vel", "rpy", "omega", "last_action", "step_count"):
            assert torch.equal(a.env.get_state(f), b.env.get_state(f)), f
        assert a.env.tick == b.env.tick and a._sample_calls == b._sample_calls
        # move the running statistics / weights identically on both before the second rollout
        for t_ in tr:
            t_.ac.obs_oms.update(t_.obs_buf.reshape(-1, t_.obs_buf.shape[-1])) if t_.ac.obs_oms is not None else None
            with torch.no_grad():
                for p_ in t_.ac.parameters():
                    p_.mul_(1.01)
    for t_ in tr:
        t_.env.close()


@pytest.mark.gpu
def test_fused_rollout_refuses_what_it_is_not_built_for_and_the_trainer_falls_back():
    """What is left without a rollout kernel, e.g. the opt-in ground-effect extension on Hover (round 6 built it for TakeOff, the
    task it matters for).  A refused call leaves the handle as it was (support is decided first)."""
    import phoenix_drone_simulation_amd as pds
    from phoenix_drone_simulation_amd.ppo import PPOTrainer
    env = pds.make("DroneHoverSimpleEnv-v0", num_envs=128, seed=3, use_ground_effect=True)
    tr = PPOTrainer(env, rollout_len=4, epochs=2, seed=5, fused=True)
    before = {f: env.get_state(f).clone() for f in ("pos", "step_count", "noisy_obs")}
    tick = env.tick
    tr.roll_out()
    assert tr.fused_rollout is False  # PDS_EUNSUPPORTED -> per-step kernels
    env2 = pds.make("DroneHoverSimpleEnv-v0", num_envs=128, seed=3, use_ground_effect=True)
    with pytest.raises(NotImplementedError):
        PPOTrainer(env2, rollout_len=4, epochs=2, seed=5, fused=True, fused_rollout=True).roll_out()
    assert env2.tick == tick and env2.sync_tick() == tick
    for f, v in before.items():
        assert torch.equal(env2.get_state(f), v), f
    env.close(); env2.close()


def _reference_learning_curves(name="learning_curve.json"):
    import json
    ref = json.load(open(os.path.join(os.path.dirname(GOLD), name)))
    seeds = [str(s_) for s_ in ref["seeds"]]
    return ref, {k: np.array([ref["curves"][s_][k] for s_ in seeds]) for k in ("EpRet/Mean", "EpLen/Mean")}


def test_learning_curve_fixture_is_a_sample_and_the_comparison_is_calibrated_on_it():
    """tests/golden/learning_curve.json is a statistical SAMPLE of the reference trainer's run distribution, not a known answer:
    it says so, holds >= 24 seeds, and the
    two-sample comparison the GPU test applies (golden_util.compare_learning_curves) accepts the reference against itself --
    its own first half of seeds against the second -- while it rejects a copy shifted by the offset round 4's band could not
    see (+10 steps of episode length from epoch 24 on)."""
    ref, cur = _reference_learning_curves()
    assert "STATISTICAL SAMPLE" in ref["what"]
    n = len(ref["seeds"])
    assert n >= 24 and (ref["epochs"], ref["steps_per_epoch"], ref["env_id"]) == (40, 32000, "DroneHoverSimpleEnv-v0")
    for key, x in cur.items():
        fails, _ = gu.compare_learning_curves(x[: n // 2], x[n // 2:])
        assert not fails, (key, fails)
    # power: what the late-phase test can see is set by the seed-to-seed spread of the late level (SD ~13 steps): the standard
    # error of the difference of two n-seed means is 13 sqrt(2 / n), and a shift of 3.2 of them is rejected at p = 0.01 -- the
    # pool against itself + that shift fails, as does a run distribution with the early rise two epochs late
    x = cur["EpLen/Mean"]
    lvl = x[:, 23:].mean(axis=1)
    shift = 3.2 * lvl.std(ddof=1) * np.sqrt(2.0 / n)
    assert shift < 14.0, shift  # (24 runs: ~13 steps.  Round 4 saw +8..12 against the mean of SIX runs, which itself sits 4-5
    #                              steps below the mean of either 24-run sample of the reference)
    shifted = x.copy(); shifted[:, 23:] += shift
    fails, rep = gu.compare_learning_curves(x + np.random.default_rng(0).normal(0, 1e-3, x.shape), shifted)
    assert any(f[0].startswith("late") for f in fails), rep
    late = np.concatenate([x[:, :1], x[:, :1], x[:, :-2]], axis=1)
    fails, rep = gu.compare_learning_curves(x, late)
    assert fails, rep


def _train_runs_side_by_side(env_id, seeds, num_envs, steps_per_epoch, epochs, kw, threads=12):
    """PPOTrainer runs (PPO defaults == the reference's hyper-parameters, asserted by the callers) for `seeds`, `threads` at a
    time on one GPU, in a CHILD process (profiles/tools/learning_threads.py: one Python thread and one HIP stream per run) that
    starts with GPU_MAX_HW_QUEUES=16 -- the runtime's default of 4 hardware queues serialises streams that share one.  A rollout
    of one env is one block on one of 256 CUs, so runs at the reference's layout overlap: 12.4 s -> 1.9 s per run, bit-identical
    to sequential runs (the tool checks that when run by hand).  -> {seed: (EpRet/Mean [epochs], EpLen/Mean [epochs])}"""
    import json
    import subprocess
    import sys
    import tempfile
    assert (steps_per_epoch, epochs) == (32000, 40) and seeds == list(range(seeds[0], seeds[0] + len(seeds))) and not kw
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "runs.json")
        cmd = [sys.executable, os.path.join(root, "profiles", "tools", "learning_threads.py"), "--seeds", str(len(seeds)),
               "--first-seed", str(seeds[0]), "--threads", str(threads), "--envs", str(num_envs), "--env-id", env_id,
               "--out", out, "--no-check"]
        r = subprocess.run(cmd, env=dict(os.environ, GPU_MAX_HW_QUEUES="16"), capture_output=True, text=True, timeout=1500, cwd=root)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        runs = json.load(open(out))
    return {int(k): (v[1], v[0]) for k, v in runs.items()}


def test_circle_learning_curve_fixture_is_a_sample_of_the_reference():
    """tests/golden/learning_curve_circle.json: 24 runs of the reference's learn() on DroneCircleSimpleEnv-v0 (same generator,
    `--env`); halves of it pass the comparison against each other."""
    ref, cur = _reference_learning_curves("learning_curve_circle.json")
    n = len(ref["seeds"])
    assert "STATISTICAL SAMPLE" in ref["what"] and n >= 24 and ref["obs_dim"] == 40
    assert (ref["epochs"], ref["steps_per_epoch"], ref["env_id"]) == (40, 32000, "DroneCircleSimpleEnv-v0")
    for key, x in cur.items():
        fails, _ = gu.compare_learning_curves(x[: n // 2], x[n // 2:])
        assert not fails, (key, fails)
    assert cur["EpLen/Mean"][:, :3].mean() < 40 and cur["EpLen/Mean"][:, -3:].mean() > 150  # the task is learned


@pytest.mark.gpu
def test_ppo_learning_curve_matches_the_reference_trainers_run_distribution():
    """End-to-end pin of the caller ("Hover return vs epochs"): tests/golden/learning_curve.json holds the
    per-epoch log of the REFERENCE's own ProximalPolicyOptimizationAlgorithm.learn() (algs/ppo/ppo.py:50-63,
    algs/iwpg/iwpg.py:259-485, defaults algs/ppo/defaults.py:6-19) on its own DroneHoverSimpleEnv-v0 with the env's default
    sensor noise / domain randomisation / thrust noise: 24 seeds x 40 epochs x 32 000 steps (oracle/refgen/
    gen_golden_learning.py, 20-28 minutes per seed) -- a SAMPLE of its run distribution.  PPOTrainer on the HIP envs runs the
    same configuration AT THE REFERENCE'S LAYOUT -- ONE env x 32 000 steps per epoch (IWPGAlgorithm.roll_out: one env per MPI
    rank, one epoch-end cut), 40 epochs (the exploration-noise and learning-rate schedules span exactly them), the same
    hyper-parameters, env.reset() at the start of every rollout -- under its own randomness, 24 seeds, twelve at a time.
    The two samples are compared seed-wise (golden_util.compare_learning_curves): Welch's t-test on the per-seed level of
    four phases of the curve -- the late one is the test that sees the one-sided offset round 4's min/max band hid -- and
    per epoch with a Bonferroni factor; p > 0.01 everywhere, for EpLen and EpRet.
    Round 5's bisection (profiles/r05_learning_curve.txt, DESIGN 8b: 24 seeds each of 1 x 32 000 and of 8 x 4 000, 12 each of
    the per-step kernels, the PyTorch-op path, 32 x 1 000, 64 x 500, and this trainer's logic on the REFERENCE's envs) found
    the round-4 offset to be sampling noise: the six reference runs of round 4 average 89.2 steps in epochs 24-40, the 24 of
    this fixture 93.3 +- 2.9 (run-to-run SD 14.2), an independent earlier sample of 24 runs 94.3 +- 2.6 (its own second dozen
    sits 7.6 above its first); HIP 1 x 32 000: 93.3 +- 2.9 (-0.01 standard errors of the difference), HIP 8 x 4 000:
    95.8 +- 2.4 (+0.67).  What the layout DOES change is the first peak (epochs 9-16): 88.9 (reference) / 85.1 (1 env) / 82.4
    (8 envs) / 71.2 (32 envs x 1 000 steps) -- presumably because every env's episode is cut and bootstrapped at the rollout end
    -- which is why the pin runs the reference's layout.
    The trainer is deterministic for fixed seeds (also side by side), so this test does not flake: it fails when the code
    changes the numbers."""
    ref, rcur = _reference_learning_curves()
    E, spe = ref["epochs"], ref["steps_per_epoch"]
    assert (E, spe, ref["env_id"]) == (40, 32000, "DroneHoverSimpleEnv-v0") and len(ref["seeds"]) >= 24
    hyper = ref["hyper"]  # what the reference ran with == PPOTrainer's defaults
    kw = dict(gamma=hyper["gamma"], lam=hyper["lam"], pi_lr=hyper["pi_lr"], vf_lr=hyper["vf_lr"],
              train_pi_iterations=hyper["train_pi_iterations"], train_v_iterations=hyper["train_v_iterations"],
              num_mini_batches=hyper["num_mini_batches"], use_kl_early_stopping=hyper["use_kl_early_stopping"],
              use_linear_lr_decay=hyper["use_linear_lr_decay"], use_exploration_noise_anneal=hyper["use_exploration_noise_anneal"],
              use_reward_scaling=hyper["use_reward_scaling"], use_standardized_obs=hyper["use_standardized_obs"],
              use_max_grad_norm=hyper["use_max_grad_norm"], use_entropy=hyper["use_entropy"])
    from phoenix_drone_simulation_amd.ppo import PPOTrainer
    import inspect
    defaults = {k: v.default for k, v in inspect.signature(PPOTrainer.__init__).parameters.items()}
    assert all(defaults[k] == v for k, v in kw.items()), [(k, defaults[k], v) for k, v in kw.items() if defaults[k] != v]
    seeds = list(range(100, 124))
    runs = _train_runs_side_by_side(ref["env_id"], seeds, 1, spe, E, {})
    curves = {"EpRet/Mean": np.array([runs[s_][0] for s_ in seeds]), "EpLen/Mean": np.array([runs[s_][1] for s_ in seeds])}
    for key, mine in curves.items():
        fails, report = gu.compare_learning_curves(mine, rcur[key])
        assert not fails, (key, fails, report)
    # ... and it is the reference's curve: rise, dip while the noise anneals, rise
    ln = curves["EpLen/Mean"].mean(axis=0)
    assert ln[:3].mean() < 16 and ln[9:13].mean() > 70 and ln[17:21].mean() < ln[9:13].mean() - 4 and ln[35:].mean() > ln[17:21].mean() + 10


@pytest.mark.gpu
@pytest.mark.parametrize("fixture,corr_early,corr_peak", [("learning_curve.json", 0.75, 0.2), ("learning_curve_seeds100.json", 0.75, 0.2),
                                                          ("learning_curve_circle.json", 0.7, 0.7)],
                         ids=["hover", "hover_seeds100", "circle"])
def test_ppo_learning_curve_seed_by_seed_at_the_references_own_seeds(fixture, corr_early, corr_peak):
    """Round 6: the same comparison PAIRED by seed.  A run's seed fixes its initial networks (torch.manual_seed before the
    actor-critic is built, algs/iwpg/iwpg.py:83-90 / PPOTrainer.__init__), and the initialisation decides most of the early
    learning speed: the reference's own two samples at seeds 0-23 (the fixture and round 5's runs with the unseeded env
    constructor) correlate 0.98 over seeds in epochs 4-8 and 0.70 at the first peak.  PPOTrainer on the HIP envs AT THOSE SEEDS
    starts from the same networks under different env and sampling randomness, so (i) its per-seed phase levels must correlate
    with the fixture's -- measured 0.91 (EpLen) / 0.85 (EpRet) in epochs 4-8, 0.49 / 0.37 at the first peak: the networks are
    the reference's for the same seed -- and (ii) the PAIRED difference is a far sharper test than two independent samples:
    its standard error in epochs 4-8 is 0.5 steps (1.4 %) where Welch's test on independent seeds has 1.7.  Measured: +0.04
    +- 0.51 (epochs 4-8), -0.9 +- 2.0 (first peak, epochs 9-16), +2.1 +- 2.4 (dip), -0.9 +- 3.4 (late).  This is also what
    explains the "one-sidedly low first peak" of rounds 5-6 (85-86 at other seeds against the reference's 88.8): seeds 0-23
    are lucky initialisations -- HIP runs at seeds 0-23 reach 88.0, at seeds 24-255 85.5 +- 0.6, at seeds 1000-1191 86.1 +- 0.7
    (profiles/r06_learning_curve_paired.txt).  Bars: |paired mean difference| <= 3 standard errors in every phase,
    correlation over seeds >= 0.75 in epochs 4-8 and >= 0.2 at the first peak.  Deterministic for fixed seeds.
    Circle (learning_curve_circle.json, seeds 0-23 as well): correlations 0.85 / 0.83 in epochs 4-8, 0.88 / 0.88 in epochs 9-16,
    0.68 / 0.64 in 17-23, 0.65 / 0.50 late (EpLen / EpRet); paired differences +0.09 +- 0.13 steps (0.4 %) in epochs 4-8, +0.6 +-
    0.6 (9-16), +1.8 +- 1.8 (17-23), +11.7 +- 9.3 (late: runs spread 106 .. 288 there), all within 1.3 standard errors, which
    are 2.5-3 x smaller than those of two independent samples; bars: correlation >= 0.7 in both early phases.
    learning_curve_seeds100.json: 24 more runs of the reference's learn() at seeds 100-123, generated in round 6 to test the
    explanation above on the reference itself (gen_golden_learning.py --first-seed 100): they peak at 84.8 +- 1.6, not 88.8;
    HIP at those seeds: correlation 0.94 / 0.90 (epochs 4-8), 0.74 / 0.75 (first peak); paired differences +1.0 +- 0.6 steps
    (epochs 4-8, the largest: 1.8 standard errors; EpRet -1.2 +- 0.5, 2.1), -1.3 +- 1.2 (first peak), +0.1 +- 1.9, +4.7 +- 2.8."""
    from scipy import stats
    ref, rcur = _reference_learning_curves(fixture)
    seeds = [int(s_) for s_ in ref["seeds"]][:24]
    assert seeds == list(range(seeds[0], seeds[0] + 24))
    runs = _train_runs_side_by_side(ref["env_id"], seeds, 1, ref["steps_per_epoch"], ref["epochs"], {})
    for key, col in (("EpRet/Mean", 0), ("EpLen/Mean", 1)):
        mine, theirs = np.array([runs[s_][col] for s_ in seeds], dtype=np.float64), np.asarray(rcur[key], dtype=np.float64)[:24]
        for name, sl in gu.LC_PHASES.items():
            x, y = mine[:, sl].mean(axis=1), theirs[:, sl].mean(axis=1)
            d = x - y
            se = d.std(ddof=1) / np.sqrt(len(d))
            corr = float(np.corrcoef(x, y)[0, 1])
            assert abs(d.mean()) <= 3.0 * se, (key, name, d.mean(), se, stats.ttest_1samp(d, 0.0).pvalue)
            if sl.stop <= 8:
                assert corr >= corr_early, (key, name, corr)
            elif sl.stop <= 16:
                assert corr >= corr_peak, (key, name, corr)


@pytest.mark.gpu
def test_ppo_learning_curve_on_circle_matches_the_reference_trainers_run_distribution():
    """The same pin on the second task: DroneCircleSimpleEnv-v0 (other reward, termination and observation; env defaults), the
    reference's own learn() for 24 seeds x 40 epochs x 32 000 steps (tests/golden/learning_curve_circle.json) against 24
    PPOTrainer runs at the reference's layout, compared seed-wise like the Hover runs (p > 0.01 on the four phases and per
    epoch, EpLen and EpRet).  Circle runs spread widely -- late level 106 .. 288 steps over the reference's 24, SD 54; 96 HIP
    runs (profiles/r05_circle_hip_runs.json): 202.0 +- 6.5 against the reference's 194.7 +- 11.1, Kolmogorov-Smirnov p = 0.91
    -- so eight runs are not enough: seeds 100-107 alone average 136 and fail the late phase at p = 2e-4."""
    ref, rcur = _reference_learning_curves("learning_curve_circle.json")
    assert (ref["epochs"], ref["steps_per_epoch"], ref["env_id"]) == (40, 32000, "DroneCircleSimpleEnv-v0") and len(ref["seeds"]) >= 24
    seeds = list(range(100, 124))
    runs = _train_runs_side_by_side(ref["env_id"], seeds, 1, ref["steps_per_epoch"], ref["epochs"], {})
    for key, col in (("EpRet/Mean", 0), ("EpLen/Mean", 1)):
        mine = np.array([runs[s_][col] for s_ in seeds])
        fails, report = gu.compare_learning_curves(mine, rcur[key])
        assert not fails, (key, fails, report)


@pytest.mark.gpu
def test_ppo_learning_curve_with_eight_envs_keeps_the_late_level():
    """The same configuration as 8 envs x 4 000 steps (2 s per run), round 4's layout: eight rollout cuts per epoch instead of
    one make the early rise a little faster and the first peak a little lower (measurably so with 24 + 24 seeds: p = 0.006 /
    0.03 -- a property of the layout, the 1-env runs above show neither); the dip and the LATE level -- what round 4's
    one-sided offset was about -- are the reference's: Welch p > 0.01 on those two phases, 24 seeds against the reference's 24."""
    ref, rcur = _reference_learning_curves()
    seeds = list(range(100, 124))
    runs = _train_runs_side_by_side(ref["env_id"], seeds, 8, ref["steps_per_epoch"], ref["epochs"], {}, threads=4)
    for key, col in (("EpRet/Mean", 0), ("EpLen/Mean", 1)):
        mine = np.array([runs[s_][col] for s_ in seeds])
        _, report = gu.compare_learning_curves(mine, rcur[key])
        for phase in ("dip (17-23)", "late (24-40)"):
            assert report[phase]["p"] > 0.01, (key, phase, report)
        for phase in ("early (epochs 4-8)", "first peak (9-16)"):
            assert report[phase]["p"] > 1e-4, (key, phase, report)  # (shifted a little, not different in kind)


@pytest.mark.gpu
def test_trainer_stays_fused_up_to_192_inputs():
    """observation_history_size = 4 .. 8 (experiments/04_*: the reference trains H = 1 .. 8, envs/base.py:303-319) gives 68 .. 192
    network inputs: since round 6 both networks stay on the fused MFMA kernels (csrc/pds_mlp_wide.hip: the first layer K-tiled;
    rounds 1-5 fell back to PyTorch ops above 64 inputs), and the rollout is one launch (pds_rollout_history).  Beyond 192 inputs
    the trainer says so and uses PyTorch ops on the HIP envs with pds_history_advance; asked for explicitly (fused=True) it
    refuses."""
    import warnings
    import phoenix_drone_simulation_amd as pds
    from phoenix_drone_simulation_amd.ppo import PPOTrainer
    for task, H, D in (("DroneHoverSimpleEnv-v0", 4, 68), ("DroneCircleSimpleEnv-v0", 8, 160), ("DroneTakeOffSimpleEnv-v0", 8, 192)):
        env = pds.make(task, num_envs=256, seed=3, observation_history_size=H)
        assert env.obs_dim == D
        with warnings.catch_warnings():
            warnings.simplefilter("error")  # (no fallback warning)
            tr = PPOTrainer(env, rollout_len=16, epochs=3, train_pi_iterations=4, train_v_iterations=1, seed=5)
        assert tr.fused is True and tr.graph_rollout is False
        ref = PPOTrainer(pds.make(task, num_envs=256, seed=3, observation_history_size=H), rollout_len=16, epochs=3,
                         train_pi_iterations=4, train_v_iterations=1, seed=5, fused=False)
        info, info_ref = tr.learn_one_epoch(), ref.learn_one_epoch()
        assert tr.fused_rollout is True  # (pds_rollout_history: one launch per rollout for H != 2 as well)
        assert np.isfinite(info["loss_pi"]) and np.isfinite(info["loss_v"])
        assert info["episodes"] > 0 or task != "DroneHoverSimpleEnv-v0"  # (TakeOff only ends episodes by the 500-step limit)
        # same seeds, same initial networks: the first epoch's value loss (before the update) agrees with the PyTorch-op
        # trainer up to the two samplers' different action draws
        assert abs(info["loss_v"] - info_ref["loss_v"]) < 0.2 * abs(info_ref["loss_v"]) + 1e-3, (info, info_ref)
        env.close(); ref.env.close()
    env = pds.make("DroneTakeOffSimpleEnv-v0", num_envs=64, seed=3, observation_history_size=10)
    assert env.obs_dim == 240
    with pytest.warns(RuntimeWarning, match="fused MFMA kernels do not cover"):
        tr = PPOTrainer(env, rollout_len=8, epochs=3, train_pi_iterations=2, train_v_iterations=1, seed=5)
    assert tr.fused is False and tr.graph_rollout is False and tr.fm_pi is None
    assert np.isfinite(tr.learn_one_epoch()["loss_pi"])
    with pytest.raises(ValueError):
        PPOTrainer(env, rollout_len=8, epochs=3, seed=5, fused=True)
    env.close()


# ---- one full update() of the reference's trainer, replayed (tests/golden/update.npz) ------------------------------------------
class _ReplayEnv:
    """Stands in for a DroneVecEnv where only the trainer's update is exercised (CPU test): one env, no stepping."""

    def __init__(self, obs_dim, device):
        self.num_envs, self.obs_dim, self.act_dim, self.device = 1, int(obs_dim), 4, torch.device(device)
        self.env_id_base = 0

    def reset(self):
        return torch.zeros(1, self.obs_dim, device=self.device), {}


def _replay_reference_updates(env, fused, rtol, atol):
    """Two consecutive update()s of the reference's own PPO run (oracle/refgen/gen_golden_update.py: its rollouts, path ends,
    bootstrap values and np.random.shuffle sequences recorded) through PPOTrainer.update(): GAE / value targets / discounted
    returns against the reference Buffer's, Loss/Pi and Loss/Value before the update, and EVERY entry of the ActorCritic
    state_dict (both networks, obs / return statistics, log_std) after each update -- Adam's state and the LambdaLR schedule
    carry over from the first update into the second (algs/iwpg/iwpg.py:282-485, algs/ppo/ppo.py:22-40)."""
    import phoenix_drone_simulation_amd.ppo as ppo
    g = np.load(os.path.join(os.path.dirname(GOLD), "update.npz"))
    T, D = int(g["steps"]), int(g["obs_dim"])
    tr = ppo.PPOTrainer(env, rollout_len=T, epochs=int(g["epochs_total"]), gamma=float(g["gamma"]), lam=float(g["lam"]),
                        clip_ratio=float(g["clip_ratio"]), pi_lr=float(g["pi_lr"]), vf_lr=float(g["vf_lr"]),
                        train_pi_iterations=int(g["train_pi_iterations"]), train_v_iterations=int(g["train_v_iterations"]),
                        num_mini_batches=int(g["num_mini_batches"]), seed=0, fused=fused, graph_rollout=False)
    dev = env.device
    with torch.no_grad():
        for k, p_ in tr.ac.state_dict().items():
            p_.copy_(torch.as_tensor(g["sd_init__" + k], device=dev))
    for e in range(2):
        d = gu.load_update_epoch(g, e, dev)
        assert abs(tr.pi_opt.param_groups[0]["lr"] - float(g[f"e{e}_lr"])) < 1e-12
        tr.ac.update(frac=e / tr.epochs)
        assert torch.allclose(tr.ac.pi.log_std.cpu(), torch.as_tensor(g[f"e{e}_log_std"], dtype=torch.float32), rtol=0, atol=1e-7)
        tr.obs_buf.copy_(d["obs"]); tr.act_buf.copy_(d["act"]); tr.rew_buf.copy_(d["rew"]); tr.val_buf.copy_(d["val"])
        tr.logp_buf.copy_(d["logp"]); tr.term_buf.copy_(d["term"]); tr.trunc_buf.copy_(d["trunc"]); tr.fval_buf.copy_(d["fval"])
        tr.last_val = d["last_val"]
        # the Buffer's own outputs (algs/core.py:461-533): reward scaling by the running return std of the epoch before
        scale = float(1.0 / (tr.ac.ret_oms.std.item() + tr.ac.ret_oms.eps))
        adv, tv, dr = ppo.gae(tr.rew_buf, tr.val_buf, tr.term_buf, tr.trunc_buf, tr.fval_buf, tr.last_val, tr.gamma, tr.lam,
                              scale, float(tr.ac.ret_oms.bound))
        for got, want, what in ((adv, d["adv"], "adv"), (tv, d["target_v"], "target_v"), (dr, d["disc_ret"], "discounted_ret")):
            gu.assert_close(got.reshape(-1).cpu().numpy(), want.cpu().numpy(), 2e-5, 2e-5, f"epoch {e} {what}")
        shuffles = iter(d["shuffles"])
        tr.perm_fn = lambda B: next(shuffles)
        info = tr.update()
        assert next(shuffles, None) is None  # every recorded shuffle was consumed: train_v_iterations of them
        assert abs(info["loss_pi"] - float(g[f"e{e}_loss_pi"])) < 1e-4 * max(1.0, abs(float(g[f"e{e}_loss_pi"]))), (e, info)
        assert abs(info["loss_v"] - float(g[f"e{e}_loss_v"])) < 1e-4 * max(1.0, abs(float(g[f"e{e}_loss_v"]))), (e, info)
        for k, p_ in tr.ac.state_dict().items():
            gu.assert_close(p_.detach().cpu().numpy(), g[f"e{e}_sd_after__" + k], rtol, atol, f"epoch {e} after update: {k}")
        tr.scheduler.step()  # learn_one_epoch
        tr.epoch += 1
    return tr


def test_two_reference_updates_replayed_through_the_torch_path_on_cpu(monkeypatch):
    """PPOTrainer.update() (PyTorch-op path, CPU tensors, the GAE kernel restated with torch ops) against two recorded updates
    of the reference's own trainer: same parameters after each to 1e-5 relative."""
    import phoenix_drone_simulation_amd.ppo as ppo
    monkeypatch.setattr(ppo, "gae", gu.gae_torch)
    g = np.load(os.path.join(os.path.dirname(GOLD), "update.npz"))
    _replay_reference_updates(_ReplayEnv(int(g["obs_dim"]), "cpu"), False, 1e-5, 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_two_reference_updates_replayed_on_the_gpu(fused):
    """The same replay on the HIP device: pds_gae, the fused MFMA gradient kernels with the Adam step riding on them and the
    value steps on the second stream (fused=True), or PyTorch ops on the device (fused=False)."""
    import phoenix_drone_simulation_amd as pds
    env = pds.make("DroneHoverSimpleEnv-v0", num_envs=1, seed=0)
    tr = _replay_reference_updates(env, fused, 2e-4, 2e-6)
    assert tr.fused is fused
    env.close()


# ---- the reference's ROLLOUTS replayed step for step (tests/golden/rollout.npz) -------------------------------------------------
class _RecordedEnv:
    """The reference env of oracle/refgen/check_rollout_logic.py as recorded transitions, behind the surface `PPOTrainer` drives
    (one env, auto-reset, final_obs, the env's `terminated` and the trainer's own `truncated = ep_len == max_ep_len`,
    algs/iwpg/iwpg.py:371).  step() checks the action it is handed against the one the reference took there."""

    def __init__(self, g, prefix):
        self.g, self.p = g, prefix
        self.num_envs, self.act_dim, self.device, self.env_id_base = 1, 4, torch.device("cpu"), 0
        self.obs_dim, self.T, self.max_ep_len = int(g[prefix + "obs_dim"]), int(g[prefix + "steps"]), int(g[prefix + "max_ep_len"])
        self.e, self.t, self.ep_len, self.dry, self.worst_act = -1, 0, 0, True, 0.0

    def _f(self, x, dt=torch.float32):
        return torch.as_tensor(np.asarray(x), dtype=dt)[None]

    def reset(self):
        if self.dry:  # PPOTrainer's constructor
            return torch.zeros(1, self.obs_dim), {}
        self.e += 1
        k = lambda name: self.g[f"{self.p}e{self.e}_{name}"]  # noqa: E731
        self.obs, self.act, self.rew, self.term = k("obs_buf"), k("act_buf"), k("rew_buf"), k("terminated")
        self.end_of = {int(end) - 1: j for j, end in enumerate(k("path_end"))}
        self.end_obs, self.reset_at, self.reset_obs = k("end_obs"), k("reset_at"), k("reset_obs")
        assert self.reset_at[0] == 0 and self.reset_at[-1] == self.T  # one at the top of roll_out, one behind the cut
        self.t, self.ep_len = 0, 0
        return self._f(self.reset_obs[0]), {}

    def step(self, a):
        t = self.t
        self.worst_act = max(self.worst_act, float(np.max(np.abs(a.numpy()[0] - self.act[t]))))
        self.ep_len += 1
        self.t += 1
        te, tr = bool(self.term[t]), self.ep_len == self.max_ep_len
        if t in self.end_of:                      # a path of the reference ended here
            fin = self.end_obs[self.end_of[t]]
            assert te or tr or self.t == self.T
            if tr or self.t == self.T:            # the action `self.ac(o)` samples and drops at a cut (iwpg.py:376): 4 draws
                torch.normal(torch.zeros(4), torch.ones(4))
            o = self.reset_obs[1 + self.end_of[t]] if self.t < self.T else fin
            self.ep_len = 0
        else:
            assert not (te or tr)
            fin = o = self.obs[t + 1]
        return self._f(o), self._f(self.rew[t]), self._f(te, torch.bool), self._f(tr, torch.bool), {"final_obs": self._f(fin)}


@pytest.mark.parametrize("scenario", ["limit500", "limit12", "limit500_term_at_cut"])
def test_reference_rollouts_replayed_step_for_step_through_the_trainer(scenario, monkeypatch):
    """`PPOTrainer.roll_out` (the per-step PyTorch path every other rollout path is held to) against `IWPGAlgorithm.roll_out`
    (algs/iwpg/iwpg.py:350-385), deterministically: three consecutive epochs of the reference's own run, its env as recorded
    transitions, torch's generator in the state the reference had in front of epoch 0.  `Normal.sample()` then draws the same
    standard normals, so the trainer must take the reference's actions (checked at every step), store the reference's buffers,
    end every path where the reference ended it with the flags and the bootstrap value the reference used -- the TimeLimit
    cut, `terminated AND cut` taking V(o), the epoch-end cut taking V(o) also for a path that terminated on the last step --
    log the same episodes, and, after the update with the recorded shuffles, hold the reference's parameters; the generator
    state in front of epochs 1 and 2 is the reference's too (nothing else drew from it).  oracle/refgen/check_rollout_logic.py
    runs the same comparison against the LIVE reference env in the build container (profiles/r06_rollout_logic.txt)."""
    import phoenix_drone_simulation_amd.ppo as ppo
    monkeypatch.setattr(ppo, "gae", gu.gae_torch)
    g = np.load(os.path.join(os.path.dirname(GOLD), "rollout.npz"))
    p = scenario + "__"
    T, E = int(g[p + "steps"]), int(g[p + "epochs_run"])
    env = _RecordedEnv(g, p)
    tr = ppo.PPOTrainer(env, rollout_len=T, epochs=int(g[p + "epochs_total"]), gamma=float(g[p + "gamma"]), lam=float(g[p + "lam"]),
                        clip_ratio=float(g[p + "clip_ratio"]), pi_lr=float(g[p + "pi_lr"]), vf_lr=float(g[p + "vf_lr"]),
                        train_pi_iterations=int(g[p + "train_pi_iterations"]), train_v_iterations=int(g[p + "train_v_iterations"]),
                        num_mini_batches=int(g[p + "num_mini_batches"]), seed=0, fused=False, reset_each_rollout=True)
    env.dry = False
    tr.obs, _ = env.reset()
    with torch.no_grad():
        for k, p_ in tr.ac.state_dict().items():
            p_.copy_(torch.as_tensor(g[p + "sd_init__" + k]))
    saved = torch.get_rng_state()
    try:
        torch.set_rng_state(torch.as_tensor(g[p + "e0_torch_rng"]))
        seen_both, seen_term_at_cut = 0, 0
        for e in range(E):
            k = lambda name: g[f"{p}e{e}_{name}"]  # noqa: E731
            assert np.array_equal(torch.get_rng_state().numpy(), k("torch_rng")), f"epoch {e}: generator state"
            assert abs(tr.pi_opt.param_groups[0]["lr"] - float(k("lr"))) < 1e-12
            tr.ac.update(frac=tr.epoch / tr.epochs)
            stats = tr.roll_out().tolist()
            assert env.t == T and env.worst_act < 2e-5, env.worst_act
            flat = lambda x: x.reshape(T, -1).squeeze(-1).numpy()  # noqa: E731
            for name, buf, tol in (("obs_buf", tr.obs_buf, 0.0), ("act_buf", tr.act_buf, 2e-5), ("rew_buf", tr.rew_buf, 0.0),
                                   ("val_buf", tr.val_buf, 2e-5), ("logp_buf", tr.logp_buf, 2e-5)):
                gu.assert_close(flat(buf), k(name), 0, tol, f"{scenario} epoch {e} {name}")
            # every path ends where the reference's ended, with the reference's bootstrap value
            term, trunc = flat(tr.term_buf).astype(bool), flat(tr.trunc_buf).astype(bool)
            ends = (np.nonzero(term | trunc)[0] + 1).tolist()
            if not ends or ends[-1] != T:
                ends.append(T)
            assert ends == k("path_end").tolist()
            boot = [float(tr.fval_buf[t - 1, 0]) if trunc[t - 1] else (0.0 if term[t - 1] else float(tr.last_val[0])) for t in ends]
            gu.assert_close(boot, k("path_last_val"), 0, 2e-5, f"{scenario} epoch {e} bootstrap values")
            seen_both += int(np.sum(k("terminated")[np.array(ends) - 1].astype(bool) & (k("path_last_val") != 0)))
            seen_term_at_cut += int(k("terminated")[-1])
            # the episodes the logger received: finished ones only (iwpg.py:381-382)
            assert stats[2] == len(k("ep_len")) and stats[1] == int(k("ep_len").sum())
            assert abs(stats[0] - float(k("ep_ret").sum())) < 1e-5 * abs(float(k("ep_ret").sum()))
            # the Buffer's outputs and the update
            scale = float(1.0 / (tr.ac.ret_oms.std.item() + tr.ac.ret_oms.eps))
            adv, tv, dr = ppo.gae(tr.rew_buf, tr.val_buf, tr.term_buf, tr.trunc_buf, tr.fval_buf, tr.last_val, tr.gamma, tr.lam,
                                  scale, float(tr.ac.ret_oms.bound))
            for got, name in ((adv, "adv_buf"), (tv, "target_val_buf"), (dr, "discounted_ret_buf")):
                gu.assert_close(flat(got), k(name), 2e-5, 2e-5, f"{scenario} epoch {e} {name}")
            shuffles = iter(torch.as_tensor(x) for x in k("shuffles"))
            tr.perm_fn = lambda B: next(shuffles)
            info = tr.update()
            assert abs(info["loss_v"] - float(k("loss_v"))) < 1e-4 * max(1.0, abs(float(k("loss_v"))))
            for name, p_ in tr.ac.state_dict().items():
                gu.assert_close(p_.numpy(), k("sd_after__" + name), 1e-4, 2e-5, f"{scenario} epoch {e} after update: {name}")
            tr.scheduler.step()
            tr.epoch += 1
        assert (seen_both > 0) == (scenario != "limit500")                # `terminated AND cut` (12-step limit; seed 8's last step)
        assert (seen_term_at_cut > 0) == (scenario == "limit500_term_at_cut")
    finally:
        torch.set_rng_state(saved)


_GPU_DDP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
import phoenix_drone_simulation_amd as pds
from phoenix_drone_simulation_amd.ppo import PPOTrainer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
env = pds.make_sharded("DroneHoverSimpleEnv-v0", 4096, rank=rank, world_size=world, device="cuda:0", seed=7)
assert env.num_envs == 4096 // world and env.env_id_base == rank * (4096 // world)
tr = PPOTrainer(env, rollout_len=16, epochs=4, train_pi_iterations=6, train_v_iterations=2, num_mini_batches=4, seed=7, fused=True)
for _ in range(2):
    info = tr.learn_one_epoch()
torch.cuda.synchronize()
# every rank holds the same parameters and the same running statistics after two epochs of averaged gradients
flat = torch.cat([p.detach().reshape(-1).float().cpu() for p in tr.ac.state_dict().values()])
both = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(both, flat)
for r in range(world):
    assert torch.equal(both[r], both[0]), (rank, r, float((both[r] - both[0]).abs().max()))
assert bool(torch.isfinite(flat).all()) and info["episodes"] > 0
assert info["total_env_steps"] == 2 * 16 * 4096  # the whole job's steps
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok", info["ep_len"])
"""


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_multi_rank_trainer_on_one_device_keeps_the_ranks_in_step(world, tmp_path):
    """The trainer's multi-rank path with the FUSED kernels (one process per GPU; mpi_avg_grads / sync_params /
    the MPI-averaged OnlineMeanStd of utils/mpi_tools.py:30-44, utils/online_mean_std.py:76-83): two / eight ranks, each with its
    shard of the envs, all on cuda:0 with gloo as the collective backend (the 1-GPU stand-in for RCCL): parameter broadcast, one
    flattened gradient all-reduce per optimiser step, all-reduced batch moments, the MAX-reduced non-finite guard -- after two
    epochs every rank holds bit-identical parameters and statistics."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(_GPU_DDP_WORKER.format(root=root))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29670 + world), WORLD_SIZE=str(world),
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o[-3000:]
        assert f"rank {r} ok" in o
