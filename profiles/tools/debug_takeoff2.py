import sys, torch
sys.path.insert(0, '.')
import phoenix_drone_simulation_amd as pds
for kw in (dict(), dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0)):
    env = pds.make("DroneTakeOffSimpleEnv-v0", num_envs=4096, seed=0, **kw)
    obs, _ = env.reset()
    g = torch.Generator(device=obs.device); g.manual_seed(0)
    for t in range(1100):
        a = 0.5 * torch.randn(4096, 4, generator=g, device=obs.device)
        obs, r, te, tr, info = env.step(a)
        bad = ~torch.isfinite(obs).all(1)
        if bad.any() or not torch.isfinite(r).all():
            i = int(bad.nonzero()[0]) if bad.any() else int((~torch.isfinite(r)).nonzero()[0])
            print(kw, "step", t, "bad envs", int(bad.sum()), "env", i, "reward", float(r[i]))
            print(" obs", [round(float(x), 3) for x in obs[i]])
            for f in ("pos", "rpy", "vel", "omega"):
                print(" ", f, env.get_state(f)[i].tolist())
            break
        if t % 250 == 0:
            print(kw, "step", t, "obs absmax %.4g" % float(obs.abs().max()), "omega max %.4g" % float(env.get_state("omega").abs().max()), "rpy max %.4g" % float(env.get_state("rpy").abs().max()))
    env.close()
