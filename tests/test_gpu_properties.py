"""Size-independent properties at BASELINE.json's full sizes (N = 2^20, configs 2-4): things the
domain guarantees regardless of the state values, checked on the HIP path at scale where the oracle
is too slow to run."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DET = dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0)
ENV_ID = {"hover": "DroneHoverSimpleEnv-v0", "circle": "DroneCircleSimpleEnv-v0",
          "takeoff": "DroneTakeOffSimpleEnv-v0"}
O = {"hover": 17, "circle": 16, "takeoff": 20}


def _actions(n, dev, seed, shift=0.0):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return (-1.0 + 2.0 / 2.25 + shift) + 0.1 * torch.randn(n, 4, generator=g, device=dev)


@pytest.mark.parametrize("task,n,kw", [
    ("hover", 1 << 20, {}),
    ("circle", 262144, dict(use_motor_dynamics=True, domain_randomization=0.10)),
    ("takeoff", 1 << 20, dict(use_ground_effect=True)),
])
def test_history_shift_and_invariants_full_size(task, n, kw):
    """obs(k+1)[:, first half] == obs(k)[:, second half's o] for envs that did not reset
    (envs/base.py:303-319 history layout); unit quaternions; z >= 0 (physics.py:182); TimeLimit
    truncation exactly at step 500 for survivors; final_obs carries the terminal observation."""
    import phoenix_drone_simulation_amd as pds
    base = dict(DET)
    base.update(kw)
    env = pds.make(ENV_ID[task], num_envs=n, seed=3, **base)
    dev = env.device
    o = O[task]
    D = env.obs_dim
    assert D == 2 * (o + 4)
    obs, _ = env.reset()
    assert torch.equal(obs[:, :o + 4], obs[:, o + 4:])  # [o0,u0,o0,u0]
    prev = obs.clone()
    steps_alive = torch.zeros(n, dtype=torch.int32, device=dev)
    saw_trunc = False
    for k in range(502):
        a = _actions(n, dev, k, 0.2 if task == "takeoff" else 0.0)
        obs, rew, term, trunc, info = env.step(a)
        done = term | trunc
        # NaN != NaN: shift/equality checks only on rows that are finite (see the TakeOff note below)
        finite = torch.isfinite(obs).all(dim=1) & torch.isfinite(prev).all(dim=1) & torch.isfinite(info["final_obs"]).all(dim=1)
        keep = ~done & finite
        # second half of the previous obs becomes the first half (o(k) then u(k-1) -> o(k), u(k-2) slot
        # holds u(k-2) == previous row's u(k-1) slot shifted): compare the o part and the action part
        # bitwise for envs past their first step; right after a reset the stored quaternion is
        # Q(sampled, unwrapped yaw) while the rebuilt one is +-Q(wrapped yaw): equal to rounding
        old = keep & (steps_alive >= 1)
        assert torch.equal(obs[old][:, :o], prev[old][:, o + 4:2 * o + 4])
        assert torch.allclose(obs[keep][:, :o], prev[keep][:, o + 4:2 * o + 4], rtol=0, atol=1e-6)
        assert torch.equal(obs[keep][:, o:o + 4], prev[keep][:, 2 * o + 4:])
        if task != "circle":  # o(k+1) ends with last_action = u(k)  (agents.py:339-348)
            assert torch.equal(obs[keep][:, o + 4 + 13:o + 4 + 17], a[keep])
        # u(k-1) slot of the new row == the action of the previous step, which the previous row
        # carried as last_action (hover/takeoff)
        q = obs[:, o + 4 + 3:o + 4 + 7]
        if task != "takeoff":
            assert torch.allclose(q.norm(dim=1), torch.ones(n, device=dev), atol=2e-6)
            assert bool((obs[:, o + 4 + 2] >= 0).all())
            assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
        else:
            # TakeOff never terminates (takeoff.py:100) and the reference's ground-effect height clip
            # is 3.5e-6 m (agents.py:153 with MAX_RPM == 1), so an env that falls back onto z = 0
            # receives a ~1e7 x thrust kick and can overflow fp32 until its TimeLimit reset: the
            # formula is reproduced as is.  Everything that is finite obeys the invariants, and a
            # reset always restores a finite state.
            fin = torch.isfinite(obs).all(dim=1)
            assert float(fin.float().mean()) > 0.5
            assert torch.allclose(q[fin].norm(dim=1), torch.ones(int(fin.sum()), device=dev), atol=2e-6)
            assert bool((obs[fin][:, o + 4 + 2] >= 0).all())
            assert bool(fin[done].all())
        # reset rows are [o0,u0,o0,u0]; their terminal observation went to final_obs
        if done.any():
            r = obs[done]
            assert torch.equal(r[:, :o + 4], r[:, o + 4:])
            fo = info["final_obs"][done & finite]
            assert torch.allclose(fo[:, :o], prev[done & finite][:, o + 4:2 * o + 4], rtol=0, atol=1e-6)
            od = done & finite & (steps_alive >= 1)
            assert torch.equal(info["final_obs"][od][:, :o], prev[od][:, o + 4:2 * o + 4])
        steps_alive = torch.where(done, torch.zeros_like(steps_alive), steps_alive + 1)
        if trunc.any():
            saw_trunc = True
            assert k == 499 or task != "takeoff"  # takeoff never terminates: all truncate at step 500
        if task == "takeoff":
            assert not term.any()
            assert bool(trunc.all()) == (k % 500 == 499)
        if task == "circle":
            assert float(info["cost"].abs().max()) == 0.0
        prev = obs.clone()
    assert saw_trunc or task != "takeoff"  # hover/circle envs rarely survive 500 random steps
    env.close()


def test_determinism_and_shard_invariance():
    """Same seed -> bitwise identical outputs; splitting the batch over two handles with
    env_id_base (the multi-GPU sharding) reproduces the single-handle run exactly."""
    import phoenix_drone_simulation_amd as pds
    n = 1 << 18
    kw = dict(DET, seed=11, max_episode_steps=37)
    one = pds.make(ENV_ID["hover"], num_envs=n, **kw)
    two = pds.make(ENV_ID["hover"], num_envs=n, **kw)
    lo = pds.make(ENV_ID["hover"], num_envs=n // 2, env_id_base=0, **kw)
    hi = pds.make(ENV_ID["hover"], num_envs=n // 2, env_id_base=n // 2, **kw)
    o1, _ = one.reset(); o2, _ = two.reset(); ol, _ = lo.reset(); oh, _ = hi.reset()
    assert torch.equal(o1, o2) and torch.equal(o1, torch.cat([ol, oh]))
    for k in range(80):
        a = _actions(n, o1.device, 100 + k)
        r1 = one.step(a); r2 = two.step(a)
        rl = lo.step(a[:n // 2].contiguous()); rh = hi.step(a[n // 2:].contiguous())
        for x, y, zl, zh in zip(r1[:4], r2[:4], rl[:4], rh[:4]):
            assert torch.equal(x, y)
            assert torch.equal(x, torch.cat([zl, zh]))
    for e in (one, two, lo, hi):
        e.close()


@pytest.mark.parametrize("kw", [dict(), dict(use_motor_dynamics=True, control_mode="AttitudeRate", aggregate_phy_steps=2)])
def test_eight_way_ragged_sharding_equals_one_handle_bitwise(kw):
    """The node's real world size: the reference-default (noisy, domain-randomised) batch cut into
    EIGHT contiguous global-id blocks by sharding.shard_range -- ragged: 8 x 1031 + 5 envs, no block a multiple of the
    64-env tile -- reproduces the single handle bit for bit over resets, steps, in-kernel auto-resets and truncations,
    because every random draw is keyed by (seed, GLOBAL env id, tick)."""
    import phoenix_drone_simulation_amd as pds
    from phoenix_drone_simulation_amd.sharding import shard_range
    world, n = 8, 8 * 1031 + 5
    base = dict(seed=17, max_episode_steps=23)
    base.update(kw)
    one = pds.make(ENV_ID["hover"], num_envs=n, **base)
    spans = [shard_range(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n and len({b - a for a, b in spans}) == 2
    shards = [pds.make(ENV_ID["hover"], num_envs=b - a, env_id_base=a, **base) for a, b in spans]
    o1, _ = one.reset()
    assert torch.equal(o1, torch.cat([s.reset()[0] for s in shards]))
    for k in range(40):
        a = _actions(n, o1.device, 300 + k)
        r1 = one.step(a)
        rs = [s.step(a[lo:hi].contiguous()) for s, (lo, hi) in zip(shards, spans)]
        for j in range(4):
            assert torch.equal(r1[j], torch.cat([r[j] for r in rs])), (k, j)
        assert torch.equal(r1[4]["final_obs"], torch.cat([r[4]["final_obs"] for r in rs]))
    for e in [one] + shards:
        e.close()


def test_reset_distribution_statistics():
    """In-kernel Philox reset distribution of Hover (envs/hover.py:201-229): ranges and moments."""
    import phoenix_drone_simulation_amd as pds
    n = 1 << 20
    env = pds.make(ENV_ID["hover"], num_envs=n, seed=21, use_motor_dynamics=True, **DET)
    obs, _ = env.reset()
    pos = env.get_state("pos"); vel = env.get_state("vel"); rpy = env.get_state("rpy")
    mx = env.get_state("motor_x"); u0 = env.get_state("last_action")
    d = pos - torch.tensor([0., 0., 1.], device=pos.device)
    assert float(d.abs().max()) <= 0.25 + 1e-6 and abs(float(d.mean())) < 1e-3
    assert abs(float(d.std()) - 0.5 / np.sqrt(12)) < 1e-3
    assert float(vel.abs().max()) <= 0.1 + 1e-6
    assert float(rpy[:, :2].abs().max()) <= np.pi / 6 + 1e-6 and float(rpy[:, 2].abs().max()) <= np.pi + 1e-5
    assert abs(float(mx.mean()) - np.sqrt(1 / 2.25)) < 1e-4 and abs(float(mx.std()) - 0.02) < 2e-4
    assert abs(float(u0.mean()) - (2 / 2.25 - 1)) < 1e-4 and abs(float(u0.std()) - 0.02) < 2e-4
    # The first observation carries the quaternion as Bullet's pose read-back returns it (row a17:
    # btMatrix3x3::setRotation -> getRotation, envs/agents.py:443): w > 0 where 4 w^2 > 1, else the largest of
    # x, y, z positive.  It differs in sign from Q(wrapped rpy) -- what the next step writes, physics.py:179 --
    # where the wrapped yaw lies below about -2 pi / 3: 17.1 % of the Hover resets (numpy, 4 M samples of
    # envs/hover.py:207-209's distribution: 0.17107; exactly 1/6 for level attitudes, e.g. TakeOff).
    sign = env.get_state("quat_sign").float().mean().item()
    assert 0.168 < sign < 0.174, sign
    q = obs[:, 3:7]
    big_w = 4 * q[:, 3] ** 2 > 1
    assert bool((q[big_w, 3] > 0).all())
    lead = q[:, :3].gather(1, q[:, :3].abs().argmax(1, keepdim=True))[:, 0]
    assert bool((lead[~big_w] > 0).all()) and int((~big_w).sum()) > n // 4
    qs = env.get_state("quat")  # +-Q(rpy) with the stored sign bit == the observation's quaternion (f32 rounding of the wrap)
    assert float((qs - q).abs().max()) < 2e-6
    # independent streams: neighbouring envs and successive ticks are uncorrelated
    x = d[:, 0]
    assert abs(float((x[:-1] * x[1:]).mean()) / float(x.var())) < 5e-3
    obs2, _ = env.reset()
    d2 = env.get_state("pos")[:, 0] - 0.0
    assert abs(float(((d2 - d2.mean()) * (x - x.mean())).mean()) / float(x.var())) < 5e-3
    env.close()


def test_masked_reset_only_touches_selected_envs():
    import phoenix_drone_simulation_amd as pds
    n = 4096
    env = pds.make(ENV_ID["circle"], num_envs=n, seed=2, **DET)
    env.reset()
    for k in range(5):
        env.step(_actions(n, env.device, k))
    before = env.get_state("pos").clone()
    steps = env.get_state("step_count").clone()
    mask = (torch.arange(n, device=env.device) % 3 == 0)
    env.reset(mask=mask)
    after = env.get_state("pos")
    assert torch.equal(after[~mask], before[~mask])
    assert not torch.equal(after[mask], before[mask])
    s2 = env.get_state("step_count")
    assert bool((s2[mask] == 0).all()) and torch.equal(s2[~mask], steps[~mask])
    env.close()


@pytest.mark.parametrize("edit", ["masked_reset", "set_tick", "set_step_count", "set_pos"])
def test_kept_noisy_observation_survives_calls_that_do_not_step_the_env(edit):
    """Reference-default (noisy) Hover: the history half of the next row is the observation that was RETURNED by the previous
    step (envs/base.py:303-319).  The single-step kernels regenerate it from the previous tick's Philox blocks and the env's
    state, so every call that moves a tile's clock, the step counter or the state without stepping the env -- a masked
    pds_reset (its kernel advances the clock of every tile), pds_set_tick, pds_set_state -- has to put it into memory first
    (pds_api.hip materialize_kept_obs; ADVICE round 4, high + medium)."""
    import phoenix_drone_simulation_amd as pds
    n = 1000
    env = pds.make(ENV_ID["hover"], num_envs=n, seed=4, auto_reset=False)
    assert env.obs_dim == 34
    env.reset()
    for k in range(3):
        prev, *_ = env.step(_actions(n, env.device, k))
    prev = prev.clone()
    keep = torch.ones(n, dtype=torch.bool, device=env.device)
    if edit == "masked_reset":
        mask = (torch.arange(n, device=env.device) % 3 == 0)
        env.reset(mask=mask)
        keep = ~mask
    elif edit == "set_tick":
        rc = env.lib.pds_set_tick(env._handle, int(env.tick) + 1000)
        assert rc == 0
    elif edit == "set_step_count":
        env.set_state("step_count", torch.zeros(n, 1, dtype=torch.int32))  # "a reset happened in the previous tick" it did not
    else:
        env.set_state("pos", env.get_state("pos") + 0.25)
    nxt, *_ = env.step(_actions(n, env.device, 9))
    # o(k): position, quaternion, velocity, filtered gyro, u(k-1) == the second half of the previous row, bit for bit
    assert torch.equal(nxt[keep, :17], prev[keep, 17:]), float((nxt[keep, :17] - prev[keep, 17:]).abs().max())
    env.close()


@pytest.mark.parametrize("task,kw", [
    ("hover", {}),
    ("hover", dict(control_mode="AttitudeRate", aggregate_phy_steps=2)),
    ("circle", dict(use_motor_dynamics=True, domain_randomization=0.10, motor_thrust_noise=0.05)),
    ("takeoff", dict(use_ground_effect=True)),
])
def test_half_and_full_observation_tile_agree_bitwise(task, kw, monkeypatch):
    """The step kernel exists with a 64-row and a 32-row LDS observation tile (csrc/pds_types.h
    kHalfTileRows; pds_step picks by grid size).  Same arithmetic, different staging: every output,
    including final_obs of the envs that finish and the ragged last tile, must be identical."""
    import phoenix_drone_simulation_amd as pds
    n = 70001  # not a multiple of 64 nor of 32
    base = dict(DET, seed=5, max_episode_steps=23)
    base.update(kw)
    envs = []
    for mode in ("half", "full"):
        monkeypatch.setenv("PDS_FORCE_TILE", mode)
        envs.append(pds.make(ENV_ID[task], num_envs=n, **base))
    monkeypatch.delenv("PDS_FORCE_TILE")
    oa, _ = envs[0].reset(); ob, _ = envs[1].reset()
    assert torch.equal(oa, ob)
    finished = 0
    for k in range(60):
        a = _actions(n, oa.device, 300 + k, shift=0.2 if task == "takeoff" else 0.0)
        ra = envs[0].step(a); rb = envs[1].step(a)
        for x, y in zip(ra[:4], rb[:4]):
            assert torch.equal(x, y)
        assert torch.equal(ra[4]["cost"], rb[4]["cost"])
        fin = (ra[2] | ra[3])
        finished += int(fin.sum())
        assert torch.equal(ra[4]["final_obs"][fin], rb[4]["final_obs"][fin])
    assert finished > n  # every env hit the 23-step limit at least once
    for e in envs:
        e.close()


@pytest.mark.parametrize("kw", [{}, dict(use_motor_dynamics=True, domain_randomization=0.1)])
def test_more_than_2_pow_23_envs_per_handle(kw):
    """Env indices beyond 2^23 (the handle is sized for HBM, not for a packed queue word): the head
    and the tail of an 8.4 M-env batch reproduce small handles with the same global env ids bitwise,
    through resets (max_episode_steps = 3) on both the merged and the deferred auto-reset path."""
    import phoenix_drone_simulation_amd as pds
    n, m = (1 << 23) + 100, 1000
    base = dict(DET, seed=17, max_episode_steps=3)
    base.update(kw)
    big = pds.make(ENV_ID["hover"], num_envs=n, **base)
    head = pds.make(ENV_ID["hover"], num_envs=m, env_id_base=0, **base)
    tail = pds.make(ENV_ID["hover"], num_envs=m, env_id_base=n - m, **base)
    ob, _ = big.reset(); oh, _ = head.reset(); ot, _ = tail.reset()
    assert torch.equal(ob[:m], oh) and torch.equal(ob[n - m:], ot)
    g = torch.Generator(device=ob.device); g.manual_seed(4)
    for k in range(7):
        a = (-1.0 + 2.0 / 2.25) + 0.1 * torch.randn(n, 4, generator=g, device=ob.device)
        rb = big.step(a); rh = head.step(a[:m].contiguous()); rt = tail.step(a[n - m:].contiguous())
        for x, y, z in zip(rb[:4], rh[:4], rt[:4]):
            assert torch.equal(x[:m], y) and torch.equal(x[n - m:], z)
        fin = rh[2] | rh[3]
        assert torch.equal(rb[4]["final_obs"][:m][fin], rh[4]["final_obs"][fin])
        if k == 2:  # envs that have not fallen yet hit the 3-step limit together
            assert float(rb[3].float().mean()) > 0.5
    for e in (big, head, tail):
        e.close()


@pytest.mark.parametrize("task,kw", [
    ("hover", dict(observation_noise=1, domain_randomization=0.1, motor_thrust_noise=0.05)),
    ("circle", dict(DET, use_motor_dynamics=True, domain_randomization=0.1, control_mode="Attitude", aggregate_phy_steps=2)),
    ("takeoff", dict(DET)),
])
def test_checkpoint_resume_is_bit_exact(task, kw):
    """state_dict() -> a fresh env with the same kwargs -> load_state_dict(): the continuation (noise,
    domain randomisation, PID state, auto-resets and their Philox draws included) is identical."""
    import phoenix_drone_simulation_amd as pds
    n = 5000
    base = dict(seed=21, max_episode_steps=17)
    base.update(kw)
    env = pds.make(ENV_ID[task], num_envs=n, **base)
    obs, _ = env.reset()
    for k in range(9):
        obs, *_ = env.step(_actions(n, obs.device, 500 + k))
    sd = env.state_dict()
    first = [env.step(_actions(n, obs.device, 600 + k))[0].clone() for k in range(25)]  # the original run
    env2 = pds.make(ENV_ID[task], num_envs=n, **base)
    env2.reset()
    env2.load_state_dict(sd)
    env.load_state_dict(sd)  # rewind the original too: both replay the same 25 steps
    for k in range(25):
        a = _actions(n, obs.device, 600 + k)
        r1, r2 = env.step(a), env2.step(a)
        for x, y in zip(r1[:4], r2[:4]):
            assert torch.equal(x, y), (k,)
        assert torch.equal(r1[0], first[k])  # ... and reproduce the original continuation
        assert torch.equal(r1[4]["cost"], r2[4]["cost"])
        fin = r1[2] | r1[3]
        assert torch.equal(r1[4]["final_obs"][fin], r2[4]["final_obs"][fin])
    env.close(); env2.close()


def test_count_nonfinite_flags_the_takeoff_overflow():
    """pds_count_nonfinite: 0 on a healthy Hover batch; > 0 once TakeOff + domain randomisation has run
    into the explicit-Euler overflow the reference's integrator has there."""
    import phoenix_drone_simulation_amd as pds
    n = 2048
    env = pds.make(ENV_ID["hover"], num_envs=n, seed=1)
    obs, _ = env.reset()
    for k in range(20):
        env.step(_actions(n, obs.device, 900 + k))
    assert env.count_nonfinite() == 0
    env.close()
    env = pds.make(ENV_ID["takeoff"], num_envs=n, seed=1)
    obs, _ = env.reset()
    g = torch.Generator(device=obs.device); g.manual_seed(0)
    for k in range(400):
        env.step(0.5 * torch.randn(n, 4, generator=g, device=obs.device))
    assert 0 < env.count_nonfinite() <= n
    env.close()


def test_device_philox_known_answers():
    """The device function behind every in-kernel draw == Random123's Philox4x32-10 / -7 known answers
    (and == the oracle's restatement on random counters)."""
    import ctypes as C
    from test_oracle_golden import PHILOX_KAT
    from oracle import oracle as po
    import phoenix_drone_simulation_amd as pds
    lib = pds.native.load()
    dev = torch.device("cuda", 0)
    for rounds in (10, 7):
        kat = [k for k in PHILOX_KAT if k[0] == rounds]
        rs = np.random.RandomState(rounds)
        extra_c = rs.randint(0, 2 ** 32, size=(500, 4), dtype=np.uint64).astype(np.uint32)
        extra_k = rs.randint(0, 2 ** 32, size=(500, 2), dtype=np.uint64).astype(np.uint32)
        ctr = np.concatenate([np.array([k[1] for k in kat], np.uint32), extra_c])
        key = np.concatenate([np.array([k[2] for k in kat], np.uint32), extra_k])
        d_ctr = torch.from_numpy(ctr.view(np.int32)).to(dev)
        d_key = torch.from_numpy(key.view(np.int32)).to(dev)
        d_out = torch.zeros(ctr.shape[0], 4, dtype=torch.int32, device=dev)
        rc = lib.pds_philox4x32(d_ctr.data_ptr(), d_key.data_ptr(), rounds, ctr.shape[0], d_out.data_ptr(),
                                torch.cuda.current_stream(dev).cuda_stream)
        assert rc == 0
        out = d_out.cpu().numpy().view(np.uint32)
        for j, k in enumerate(kat):
            assert [int(v) for v in out[j]] == k[3], (rounds, j)
        L = po.lib()
        for j in range(len(kat), ctr.shape[0], 37):
            o = (C.c_uint32 * 4)()
            L.po_philox4x32((C.c_uint32 * 4)(*[int(v) for v in ctr[j]]), (C.c_uint32 * 2)(*[int(v) for v in key[j]]), rounds, o)
            assert [int(v) for v in o] == [int(v) for v in out[j]]


def test_fresh_outputs_are_not_overwritten():
    """fresh_outputs=True: step() hands out newly allocated tensors like the reference's fresh arrays
    (envs/base.py:311); the default alternates two env-owned buffer sets (an observation survives ONE further
    step).  Same numbers either way, and a masked reset leaves the other envs' last observation in place."""
    import phoenix_drone_simulation_amd as pds
    N = 1000
    kw = dict(num_envs=N, seed=3, observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0.0)
    a = pds.make("DroneHoverSimpleEnv-v0", **kw)
    b = pds.make("DroneHoverSimpleEnv-v0", fresh_outputs=True, **kw)
    oa, _ = a.reset()
    ob, _ = b.reset()
    assert torch.equal(oa, ob)
    g = torch.Generator(device="cuda").manual_seed(0)
    kept_a, kept_b, kept_ref = [], [], []
    for t in range(5):
        act = (-0.1 + 0.2 * torch.randn(N, 4, device="cuda", generator=g))
        ra, rb = a.step(act), b.step(act)
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(ra[2], rb[2])
        done = ra[2] | ra[3]  # final_obs rows are defined where an env finished in this step
        assert torch.equal(ra[4]["final_obs"][done], rb[4]["final_obs"][done])
        kept_a.append(ra[0]); kept_b.append(rb[0]); kept_ref.append(ra[0].clone())
    assert all(torch.equal(x, y) for x, y in zip(kept_b, kept_ref))          # fresh: every step's tensor is intact
    assert kept_a[0].data_ptr() == kept_a[2].data_ptr() == kept_a[4].data_ptr()  # default: two alternating sets
    assert len({x.data_ptr() for x in kept_b}) == 5
    mask = torch.zeros(N, dtype=torch.uint8, device="cuda"); mask[::7] = 1
    last = kept_b[-1].clone()
    om, _ = b.reset(mask=mask)
    keep = ~mask.bool()
    assert torch.equal(om[keep], last[keep]) and not torch.equal(om[mask.bool()], last[mask.bool()])
    assert torch.equal(kept_b[-1], last)  # ... and the tensor handed out earlier was not written to
    a.close(); b.close()


@pytest.mark.parametrize("task,H,auto_reset,noise", [("hover", 1, True, False), ("hover", 4, True, False), ("circle", 3, True, False),
                                                      ("takeoff", 8, True, False), ("hover", 6, False, False), ("hover", 4, True, True)])
def test_history_kernel_equals_the_torch_composition(task, H, auto_reset, noise):
    """observation_history_size != 2: pds_history_advance (one launch per step) against the same update written
    with torch.cat / torch.where, bit for bit, over steps with auto-resets (ragged N, final histories on the
    rows of the envs that finished)."""
    import phoenix_drone_simulation_amd as pds
    N = 1000
    kw = dict(DET) if not noise else dict(observation_noise=1, domain_randomization=0.1, motor_thrust_noise=0.05)
    env = pds.make(ENV_ID[task], num_envs=N, seed=11, observation_history_size=H, auto_reset=auto_reset,
                   max_episode_steps=23, **kw)
    obs, _ = env.reset()
    half = env._half
    assert obs.shape == (N, H * half)
    hist = env._hist.clone()
    finished = 0
    for t in range(40):
        act = _actions(N, env.device, 100 + t) + 0.3 * torch.randn(N, 4, device=env.device)
        o, r, te, tr, info = env.step(act)
        raw = env._bufs[env._flip]["_ret"]  # the kernel's own [o(k-1), u(k-2), o(k), u(k-1)] row of this step
        want, want_final = env._advance_history_torch(hist, raw)
        assert torch.equal(o, want.reshape(N, -1)), t
        done = te | tr
        if auto_reset:
            assert torch.equal(info["final_obs"][done], want_final.reshape(N, -1)[done]), t
            finished += int(done.sum())
        else:
            assert info["final_obs"].data_ptr() == o.data_ptr()
        hist = want
    assert (not auto_reset) or finished > N // 2
    env.close()


@pytest.mark.parametrize("freq", [100, 50, 77])
def test_circle_reference_point_evaluated_in_kernel_equals_the_float64_table(freq):
    """envs/circle.py:45-56 builds the reference trajectory as a float64 table; the kernel EVALUATES point t
    (csrc/pds_types.h target_at: angle from a two-float 2 pi / P, the step's sincos).  Every index t < P: the
    target behind the observation's `target - xyz` columns of both history halves == the table to 1.5e-7 m
    (bound 9e-8 of the evaluation + the float32 rounding of the subtraction and of this test's re-addition)."""
    import phoenix_drone_simulation_amd as pds
    P = 3 * freq  # circle_time x observation_frequency (circle.py:47-49)
    env = pds.make(ENV_ID["circle"], num_envs=P, seed=1, auto_reset=False, enable_reset_distribution=False,
                   observation_frequency=freq, **{k: v for k, v in DET.items()})
    env.reset()
    env.set_state("ref_offset", np.arange(P, dtype=np.int32))
    p0 = env.get_state("pos").cpu().numpy().astype(np.float64)
    obs, *_ = env.step(torch.zeros(P, 4))
    p1 = env.get_state("pos").cpu().numpy().astype(np.float64)
    o = obs.cpu().numpy().astype(np.float64)
    half = o.shape[1] // 2
    th = 2 * np.pi * np.arange(2 * P) / P
    table = np.stack([0.25 * (1 - np.cos(th)), 0.25 * np.sin(th), np.ones(2 * P)], 1)  # circle.py:50-55 (periodic)
    t = np.arange(P)
    first = o[:, 13:16] + p0           # o(k): target of index (0 + ref_offset) % P
    second = o[:, half + 13:half + 16] + p1  # o(k+1): index (1 + ref_offset) % P
    assert np.abs(first - table[t]).max() < 1.5e-7, np.abs(first - table[t]).max()
    assert np.abs(second - table[t + 1]).max() < 1.5e-7, np.abs(second - table[t + 1]).max()
    env.close()


@pytest.mark.parametrize("task,kw", [
    ("hover", {}),                                                                          # reference default: regenerating kernel
    ("hover", dict(domain_randomization=-1, motor_thrust_noise=0.0)),                       # observation noise only
    ("hover", dict(aggregate_phy_steps=2)),
    ("hover", dict(use_motor_dynamics=True, aggregate_phy_steps=4)),
    ("circle", dict(use_motor_dynamics=True)),                                              # ref_offset, PT1 + DR
    ("circle", dict(enable_reset_distribution=False)),                                      # ref_offset survives the reset
    ("hover", dict(DET, use_latency=True, latency=0.035)),                                  # latency ring without noise
    ("hover", dict(use_latency=True, latency=0.02, use_motor_dynamics=True)),
    ("hover", dict(observation_frequency=50)),                                              # Kalman hold: the observation stays stored
    ("hover", dict(control_mode="AttitudeRate", aggregate_phy_steps=2)),                    # PID state zeroed
    ("hover", dict(control_mode="Attitude", use_latency=True, latency=0.02, observation_frequency=50)),
])
def test_split_reset_equals_the_in_place_reset_bitwise(task, kw, monkeypatch):
    """Round 6: where the single-step kernel used to reset finished envs in place (observation noise / latency ring), pds_step now
    launches it without that reset (SplitReset<V>, csrc/pds_types.h) and post_reset_kernel (csrc/pds_reset.h) behind it resets the
    finished envs densely, one env per lane.  PDS_SPLIT_RESET=0 keeps the in-place form: every output of every step -- the fresh
    rows, final_obs, flags -- and the whole state afterwards (noise states, randomised parameters, latency ring, PID state, the
    kept observation as pds_get_state returns it) must be identical, over steps in which many envs finish (short TimeLimit,
    wide actions), with a ragged last tile and a ragged last 1024-env block."""
    import phoenix_drone_simulation_amd as pds
    n = 3 * 1024 + 700 + 37
    base = dict(seed=11, max_episode_steps=11)
    base.update(kw)
    envs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("PDS_SPLIT_RESET", mode)
        envs.append(pds.make(ENV_ID[task], num_envs=n, **base))
    monkeypatch.delenv("PDS_SPLIT_RESET")
    oa, _ = envs[0].reset(); ob, _ = envs[1].reset()
    assert torch.equal(oa, ob)
    g = torch.Generator(device=oa.device); g.manual_seed(3)
    finished = 0
    for k in range(45):
        a = (-0.1 + 0.25 * torch.randn(n, 4, device=oa.device, generator=g)).contiguous()
        ra = envs[0].step(a); rb = envs[1].step(a)
        for x, y, what in zip(ra[:4], rb[:4], ("obs", "reward", "terminated", "truncated")):
            assert torch.equal(x, y), (k, what, (x != y).nonzero()[:3])
        assert torch.equal(ra[4]["cost"], rb[4]["cost"])
        fin = (ra[2] | ra[3])
        finished += int(fin.sum())
        assert torch.equal(ra[4]["final_obs"][fin], rb[4]["final_obs"][fin]), k
    assert finished >= 4 * n  # (every env at least by the 11-step limit; most configurations also by termination)
    sa, sb = envs[0].state_dict(), envs[1].state_dict()
    assert sa.keys() == sb.keys()
    for key in sa:
        x, y = sa[key], sb[key]
        assert (torch.equal(x, y) if isinstance(x, torch.Tensor) else x == y), key
    for e in envs:
        e.close()
