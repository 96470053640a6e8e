"""pds_step_k (K steps per launch, open-loop replay) and hipGraph capture of pds_step: both must be
bitwise identical to the eager single-step path (same kernels' device functions, -ffp-contract=on)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ENV_ID = {"hover": "DroneHoverSimpleEnv-v0", "circle": "DroneCircleSimpleEnv-v0",
          "takeoff": "DroneTakeOffSimpleEnv-v0"}
DET = dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0)

VARIANTS = [
    ("hover", dict(DET)),                                                       # merged reset
    ("circle", dict(DET, use_motor_dynamics=True, domain_randomization=0.1)),   # inline reset in the K-step kernel
    ("takeoff", dict(DET, use_ground_effect=True)),
    ("hover", dict()),                                                          # reference defaults: noise + DR
    ("circle", dict(domain_randomization=-1, motor_thrust_noise=0.05)),
    ("hover", dict(DET, aggregate_phy_steps=2)),
    ("hover", dict(DET, use_latency=True, latency=0.025, use_motor_dynamics=True)),
    ("circle", dict(DET, control_mode="AttitudeRate")),                         # no K-step kernel: loop of pds_step
    ("hover", dict(observation_frequency=50)),                                  # Kalman-hold variant, noise + DR
    ("takeoff", dict(observation_frequency=25, domain_randomization=-1, use_motor_dynamics=True)),
    ("circle", dict(use_latency=True, latency=0.02)),                           # latency + noise + DR (inline reset)
]


def _actions(K, N, dev, seed=0, center=-0.1, scale=0.3):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return (center + scale * torch.randn(K, N, 4, generator=g, device=dev)).contiguous()


@pytest.mark.parametrize("task,kw", VARIANTS)
@pytest.mark.parametrize("N", [1000, 4096])
def test_step_k_equals_k_single_steps_bitwise(task, kw, N):
    """Short episodes (max_episode_steps=9) so that auto-resets, TimeLimit truncations and final_obs
    rows occur inside the K-step launches."""
    import phoenix_drone_simulation_amd as pds
    K, rounds = 7, 3
    mk = lambda: pds.make(ENV_ID[task], num_envs=N, seed=11, max_episode_steps=9, **kw)
    e1, ek = mk(), mk()
    o1, _ = e1.reset()
    ok, _ = ek.reset()
    assert torch.equal(o1, ok)
    nfin = 0
    for r in range(rounds):
        acts = _actions(K, N, e1.device, seed=r)
        obs_k, rew_k, term_k, trunc_k, info_k = ek.step_k(acts)
        for s in range(K):
            o, rw, te, tr, info = e1.step(acts[s])
            w = f"{task} round {r} step {s}"
            assert torch.equal(te, term_k[s]) and torch.equal(tr, trunc_k[s]), w
            assert torch.equal(o, obs_k[s]), w
            assert torch.equal(rw, rew_k[s]) and torch.equal(info["cost"], info_k["cost"][s]), w
            fin = te | tr
            nfin += int(fin.sum())
            assert torch.equal(info["final_obs"][fin], info_k["final_obs"][s][fin]), w
    assert nfin > N  # every env finished at least once on average
    assert e1.tick == ek.tick == 1 + K * rounds
    for name in ("pos", "rpy", "vel", "omega", "last_action", "prev_action", "step_count", "quat_sign", "ref_offset"):
        assert torch.equal(e1.get_state(name), ek.get_state(name)), name
    for name, on in (("motor_x", kw.get("use_motor_dynamics")), ("params", kw.get("domain_randomization", 0.1) > 0),
                     ("ou", kw.get("motor_thrust_noise", 0.05) > 0), ("gyro_bias", kw.get("observation_noise", 1) > 0),
                     ("gyro_lpf", kw.get("observation_noise", 1) > 0), ("noisy_obs", kw.get("observation_noise", 1) > 0),
                     ("action_buffer", kw.get("use_latency")), ("action_idx", kw.get("use_latency")),
                     ("pid", kw.get("control_mode", "PWM") != "PWM")):
        if on:
            assert torch.equal(e1.get_state(name), ek.get_state(name)), name
    # the two envs continue identically on the single-step path
    a = _actions(1, N, e1.device, seed=99)[0]
    assert torch.equal(e1.step(a)[0], ek.step(a)[0])
    e1.close(); ek.close()


@pytest.mark.parametrize("task,kw", [VARIANTS[3], VARIANTS[4], VARIANTS[10], ("hover", dict(aggregate_phy_steps=2, use_motor_dynamics=True))])
def test_step_k_between_single_steps_bitwise(task, kw):
    """pds_step and pds_step_k interleaved.  The single-step kernels of the observation-noise variants do not keep the noisy
    o(k) in memory (it is regenerated), the K-step kernel reads it from oh0-2: pds_step_k materialises it first
    (materialize_oh_kernel) for the envs a pds_step has left unflagged, and a pds_step after a pds_step_k finds every env
    flagged.  Either way the trajectory is the one of single steps, bit for bit."""
    import phoenix_drone_simulation_amd as pds
    N = 1500
    mk = lambda: pds.make(ENV_ID[task], num_envs=N, seed=21, max_episode_steps=7, **kw)
    e1, em = mk(), mk()
    e1.reset(); em.reset()
    plan = [1, 1, 5, 1, 4, 3, 1, 1, 6]  # 1 = pds_step, K > 1 = pds_step_k
    for r, K in enumerate(plan):
        acts = _actions(K, N, e1.device, seed=50 + r)
        if K == 1:
            o, rw, te, tr, info = em.step(acts[0])
            got = (o[None], rw[None], te[None], tr[None], info["cost"][None], info["final_obs"][None])
        else:
            o, rw, te, tr, info = em.step_k(acts)
            got = (o, rw, te, tr, info["cost"], info["final_obs"])
        for s in range(K):
            o1, rw1, te1, tr1, info1 = e1.step(acts[s])
            w = f"{task} call {r} step {s}"
            fin = te1 | tr1
            assert torch.equal(o1, got[0][s]) and torch.equal(rw1, got[1][s]), w
            assert torch.equal(te1, got[2][s]) and torch.equal(tr1, got[3][s]) and torch.equal(info1["cost"], got[4][s]), w
            assert torch.equal(info1["final_obs"][fin], got[5][s][fin]), w
        for name in ("pos", "rpy", "omega", "gyro_bias", "gyro_lpf", "noisy_obs", "step_count"):
            assert torch.equal(e1.get_state(name), em.get_state(name)), (r, name)
    assert e1.tick == em.tick
    e1.close(); em.close()


def test_step_k_ragged_and_no_autoreset():
    import phoenix_drone_simulation_amd as pds
    for N in (1, 63, 321):
        e1 = pds.make(ENV_ID["hover"], num_envs=N, seed=3, auto_reset=False, **DET)
        ek = pds.make(ENV_ID["hover"], num_envs=N, seed=3, auto_reset=False, **DET)
        e1.reset(); ek.reset()
        acts = _actions(5, N, e1.device, seed=1, scale=0.05)
        obs_k = ek.step_k(acts)[0]
        for s in range(5):
            assert torch.equal(e1.step(acts[s])[0], obs_k[s]), (N, s)
        e1.close(); ek.close()


@pytest.mark.parametrize("task,kw", [VARIANTS[0], VARIANTS[1], VARIANTS[3]])
def test_hipgraph_replay_equals_eager_bitwise(task, kw):
    """The tick and the action-ring parity live in device memory (one word per tile), so a captured
    sequence of pds_step launches replays correctly: two replays of a T-step graph == 2 T eager steps."""
    import phoenix_drone_simulation_amd as pds
    N, T = 4096, 6
    mk = lambda: pds.make(ENV_ID[task], num_envs=N, seed=5, max_episode_steps=8, **kw)
    ee, eg = mk(), mk()
    dev = ee.device
    acts = _actions(T, N, dev, seed=4)
    ee.reset(); eg.reset()
    rec = torch.zeros(T, N, ee.obs_dim, device=dev)
    rew = torch.zeros(T, N, device=dev)
    fin = torch.zeros(T, N, dtype=torch.bool, device=dev)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for s in range(T):
            o, r, te, tr, info = eg.step(acts[s])
            rec[s].copy_(o); rew[s].copy_(r); fin[s].copy_(te | tr)
    nfin = 0
    for rep in range(2):
        graph.replay()
        torch.cuda.synchronize()
        for s in range(T):
            o, r, te, tr, info = ee.step(acts[s])
            assert torch.equal(o, rec[s]) and torch.equal(r, rew[s]) and torch.equal(te | tr, fin[s]), (rep, s)
            nfin += int(fin[s].sum())
    assert nfin > 0
    assert eg.sync_tick() == ee.tick == 1 + 2 * T
    a = acts[0]
    assert torch.equal(ee.step(a)[0], eg.step(a)[0])  # eager again after the replays
    ee.close(); eg.close()


def test_entry_points_keep_the_callers_current_device():
    import phoenix_drone_simulation_amd as pds
    env = pds.make(ENV_ID["hover"], num_envs=256, **DET)
    before = torch.cuda.current_device()
    env.reset(); env.step(torch.zeros(256, 4, device=env.device)); env.get_state("pos"); env.count_nonfinite()
    assert torch.cuda.current_device() == before
    env.close()


@pytest.mark.parametrize("H,auto_reset", [(4, True), (1, True), (6, False)])
def test_step_k_with_observation_history_other_than_two(H, auto_reset):
    """observation_history_size != 2 (experiments/04_*, envs/base.py:303-319) through pds_step_k: the K rows of the launch run
    through pds_history_advance one after the other -- bit for bit what K step() calls return, final histories included."""
    import phoenix_drone_simulation_amd as pds
    n, K = 3000, 7
    kw = dict(seed=9, observation_history_size=H, max_episode_steps=5, auto_reset=auto_reset)
    a, b = (pds.make("DroneHoverSimpleEnv-v0", num_envs=n, **kw) for _ in range(2))
    oa, _ = a.reset(); ob, _ = b.reset()
    assert torch.equal(oa, ob) and oa.shape == (n, H * 17)
    g = torch.Generator(device=a.device); g.manual_seed(0)
    for rnd in range(2):
        acts = (-0.11 + 0.1 * torch.randn(K, n, 4, generator=g, device=a.device)).contiguous()
        ko, kr, kt, ku, kinfo = a.step_k(acts)
        for k in range(K):
            o, r, t, u, info = b.step(acts[k])
            assert torch.equal(ko[k], o) and torch.equal(kr[k], r) and torch.equal(kt[k], t) and torch.equal(ku[k], u), (rnd, k)
            done = t | u
            if auto_reset and bool(done.any()):
                assert torch.equal(kinfo["final_obs"][k][done], info["final_obs"][done]), (rnd, k)
    a.close(); b.close()


def test_captured_step_k_with_history_carries_it_from_replay_to_replay():
    """step_k with observation_history_size != 2 used to REBIND the env's history tensor per call, so a hipGraph
    that captured a call re-read the address of the capture at every replay.  The history now lives in one env-owned buffer:
    two replays of a captured step_k equal two eager calls on a twin env, bit for bit; capturing before the buffer was adopted
    (no eager step_k call yet) is refused; and a masked reset() after step_k copies the rows of the launch's LAST step for the
    envs outside the mask."""
    import phoenix_drone_simulation_amd as pds
    n, K, H = 1000, 4, 4
    kw = dict(seed=4, observation_history_size=H, max_episode_steps=6)
    a, b, c = (pds.make("DroneHoverSimpleEnv-v0", num_envs=n, **kw) for _ in range(3))
    for e in (a, b, c):
        e.reset()
    g = torch.Generator(device=a.device); g.manual_seed(1)
    acts = (-0.11 + 0.1 * torch.randn(K, n, 4, generator=g, device=a.device)).contiguous()
    with pytest.raises(RuntimeError, match="once eagerly"):
        with torch.cuda.graph(torch.cuda.CUDAGraph()):
            c.step_k(acts)
    c.close()
    ref = [tuple(x.clone() for x in b.step_k(acts)[:4]) for _ in range(3)]
    first = a.step_k(acts)                        # eager: adopts the history buffer
    assert torch.equal(first[0], ref[0][0])
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = a.step_k(acts)
    for rep in (1, 2):                            # (capture does not execute: the first replay is the second step_k call)
        graph.replay()
        torch.cuda.synchronize()
        for x, y in zip(out[:4], ref[rep]):
            assert torch.equal(x, y), rep
    a.sync_tick()
    # masked reset after step_k: rows outside the mask are the launch's last observation, not an older step()'s
    mask = torch.zeros(n, dtype=torch.bool, device=a.device); mask[::3] = True
    o, _ = a.reset(mask=mask)
    assert torch.equal(o[~mask], ref[2][0][K - 1][~mask])
    a.close(); b.close()
