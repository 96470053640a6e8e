"""Which outputs differ between the 32-row and the 64-row tile kernels? (debug helper)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phoenix_drone_simulation_amd as pds
task = sys.argv[1] if len(sys.argv) > 1 else "DroneHoverSimpleEnv-v0"
n = 70001
kw = dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0, seed=5, max_episode_steps=23)
envs = []
for mode in ("half", "full"):
    os.environ["PDS_FORCE_TILE"] = mode
    envs.append(pds.make(task, num_envs=n, **kw))
oa, _ = envs[0].reset(); ob, _ = envs[1].reset()
print("reset equal", torch.equal(oa, ob))
g = torch.Generator(device=oa.device); g.manual_seed(1)
for k in range(30):
    a = -0.111 + 0.1 * torch.randn(n, 4, generator=g, device=oa.device)
    ra = envs[0].step(a); rb = envs[1].step(a)
    d = (ra[0] != rb[0])
    if d.any() or not torch.equal(ra[1], rb[1]):
        cols = d.any(0).nonzero().flatten().tolist()
        rows = d.any(1).nonzero().flatten()
        print("step", k, "obs cols differing", cols, "rows", rows.numel(), "first rows", rows[:5].tolist(),
              "reward diff", int((ra[1] != rb[1]).sum()))
        r = rows[0].item()
        print(" half", ra[0][r].tolist()); print(" full", rb[0][r].tolist())
        print(" term/trunc", ra[2][r].item(), ra[3][r].item(), rb[2][r].item(), rb[3][r].item())
        for f in ("pos", "rpy", "vel", "omega"):
            try:
                sa = envs[0].get_state(f); sb = envs[1].get_state(f)
                print(" state", f, int((sa != sb).any(1).sum()))
            except Exception as e:
                print(" state", f, "err", e)
        break
else:
    print("no difference in 30 steps")
