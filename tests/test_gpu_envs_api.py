"""API conformance and smoke tests in the shape of the reference's own tests/test_envs.py:175-268
(stable-baselines env_checker + one full episode per registered id), for the batched envs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N = 300


def _ids():
    import phoenix_drone_simulation_amd as pds
    return [i for i in pds.registry if "Drone" in i]


def _sample_actions(env, g):
    # action_space.sample(): uniform in the Box (tests/test_envs.py:246)
    lo = torch.as_tensor(env.action_space.low, device=env.device)
    hi = torch.as_tensor(env.action_space.high, device=env.device)
    return lo + (hi - lo) * torch.rand(env.num_envs, 4, generator=g, device=env.device)


def test_registry_has_the_three_simple_ids():
    assert set(_ids()) == {"DroneHoverSimpleEnv-v0", "DroneCircleSimpleEnv-v0", "DroneTakeOffSimpleEnv-v0"}


@pytest.mark.parametrize("env_id", ["DroneHoverSimpleEnv-v0", "DroneCircleSimpleEnv-v0", "DroneTakeOffSimpleEnv-v0"])
def test_gym_api(env_id):
    """_check_spaces / _check_returned_values (tests/test_envs.py:120-171): spaces exist and are Boxes,
    reset -> (obs, info), step -> 5-tuple with the right shapes / types, info is a dict."""
    import phoenix_drone_simulation_amd as pds
    env = pds.make(env_id, num_envs=N, seed=42)
    assert hasattr(env, "observation_space") and hasattr(env, "action_space")
    osp, asp = env.observation_space, env.action_space
    assert osp.shape == (env.obs_dim,) and asp.shape == (4,)
    assert osp.dtype == np.float32 and asp.dtype == np.float32
    assert np.all(osp.low == -1000) and np.all(osp.high == 1000)          # envs/base.py:147-150
    assert np.all(asp.low == -1) and np.all(asp.high == 1)                # symmetric, normalised
    assert env.unwrapped is env and env._max_episode_steps == 500
    out = env.reset(seed=42)
    assert isinstance(out, tuple) and len(out) == 2
    obs, info = out
    assert isinstance(info, dict)
    assert obs.shape == (N,) + osp.shape and obs.dtype == torch.float32 and obs.is_cuda
    assert bool(torch.isfinite(obs).all()) and float(obs.abs().max()) <= 1000
    g = torch.Generator(device=env.device); g.manual_seed(0)
    data = env.step(_sample_actions(env, g))
    assert len(data) == 5, "The `step()` method must return five values: obs, reward, terminated, truncated, info"
    obs, reward, terminated, truncated, info = data
    assert obs.shape == (N,) + osp.shape and obs.dtype == torch.float32
    assert reward.shape == (N,) and reward.dtype == torch.float32          # "The reward ... must be a float"
    assert terminated.shape == (N,) and terminated.dtype == torch.bool     # "The `terminated` signal must be a boolean"
    assert truncated.shape == (N,) and truncated.dtype == torch.bool
    assert isinstance(info, dict) and info["cost"].shape == (N,) and info["final_obs"].shape == obs.shape
    env.close()


@pytest.mark.parametrize("env_id", ["DroneHoverSimpleEnv-v0", "DroneCircleSimpleEnv-v0", "DroneTakeOffSimpleEnv-v0"])
def test_all_envs_one_episode(env_id):
    """TestEnvs.check_env (tests/test_envs.py:239-253): run every env until its first episode ends, with
    sampled actions; every env of the batch must finish within the 500-step limit."""
    import phoenix_drone_simulation_amd as pds
    env = pds.make(env_id, num_envs=N, seed=42, domain_randomization=-1)  # (TakeOff + DR overflows by design: DESIGN 5)
    env.reset(seed=42)
    g = torch.Generator(device=env.device); g.manual_seed(1)
    done = torch.zeros(N, dtype=torch.bool, device=env.device)
    rewards = torch.zeros(N, device=env.device); costs = torch.zeros(N, device=env.device)
    steps = 0
    while not bool(done.all()):
        x, r, terminated, truncated, info = env.step(_sample_actions(env, g))
        steps += 1
        rewards += torch.where(done, torch.zeros_like(r), r)
        costs += torch.where(done, torch.zeros_like(r), info["cost"])
        done |= terminated | truncated
        assert steps <= 500
    assert bool(torch.isfinite(rewards).all())
    if env_id == "DroneTakeOffSimpleEnv-v0":
        assert steps == 500  # never terminates (envs/takeoff.py:100): TimeLimit only
    env.close()


def test_errors_are_loud():
    import phoenix_drone_simulation_amd as pds
    with pytest.raises(AssertionError):
        pds.make("DroneHoverSimpleEnv-v0", num_envs=4, control_mode="Bogus")        # envs/agents.py:70-71
    with pytest.raises(KeyError):
        pds.make("DroneHoverBulletEnv-v0", num_envs=4)                               # Bullet envs: out of scope
    env = pds.make("DroneHoverSimpleEnv-v0", num_envs=4)
    with pytest.raises(ValueError, match="before pds_reset"):
        env.step(torch.zeros(4, 4, device=env.device))                               # step before reset
    env.reset()
    with pytest.raises(ValueError):
        env.step(torch.zeros(5, 4, device=env.device))                               # wrong batch size
    env.close()


def test_latency_and_hold_with_the_ground_effect_extension():
    """Round 3: use_latency / set_latency and the Kalman-hold branch are built together with the opt-in ground-effect
    extension (control_mode PWM); with a PID mode the combination is still refused."""
    import phoenix_drone_simulation_amd as pds
    env = pds.make("DroneHoverSimpleEnv-v0", num_envs=256, seed=1, use_ground_effect=True, observation_noise=-1)
    env.reset()
    env.set_latency(0.02)
    assert env.latency_steps == 2
    a = torch.zeros(256, 4, device=env.device)
    for _ in range(3):
        o, r, te, tr, info = env.step(a)
    assert bool(torch.isfinite(o).all())
    env.close()
    env = pds.make("DroneHoverSimpleEnv-v0", num_envs=256, seed=1, use_ground_effect=True, observation_frequency=50)
    env.reset()
    o, *_ = env.step(a)
    assert o.shape == (256, 34) and bool(torch.isfinite(o).all())
    env.close()
    with pytest.raises(NotImplementedError):
        pds.make("DroneHoverSimpleEnv-v0", num_envs=16, use_ground_effect=True, use_latency=True, latency=0.02, control_mode="AttitudeRate")
