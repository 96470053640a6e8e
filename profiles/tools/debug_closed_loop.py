"""Error growth of the closed-loop policy rollout, HIP f32 vs oracle f64 (debug helper)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import phoenix_drone_simulation_amd as pds
from oracle import oracle as po
from phoenix_drone_simulation_amd.policy_io import load_network_json
fix = os.path.join(ROOT, "tests", "golden", "policy_PWM_seed_00000_model.json")
N, seed = 256, 9
base = dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0)
env = pds.make("DroneCircleSimpleEnv-v0", num_envs=N, seed=seed, **base)
orc = po.OracleBatch("circle", N, precision=sys.argv[1] if len(sys.argv) > 1 else "f64", **base)
pol = load_network_json(fix).to(env.device).double()
obs, _ = env.reset(); oobs = orc.reset(seed, 0)
alive = np.ones(N, bool); nterm = 0
for t in range(500):
    a_g = pol(obs.double()).float().contiguous()
    a_o = pol(torch.tensor(oobs, dtype=torch.float64, device=env.device)).float().cpu().numpy()
    tick = env.tick
    obs, r, term, trunc, info = env.step(a_g)
    oobs, orr, oterm, otrunc, _ = orc.step(a_o, seed=seed, tick=tick, auto_reset=True)
    tg, to = term.cpu().numpy().astype(bool), oterm.astype(bool)
    alive &= (tg == to); nterm += int(to.sum())
    if t in (0, 1, 5, 10, 20, 50, 100, 200, 300, 499):
        e = np.abs(obs.cpu().numpy()[:, 20:23] - oobs[:, 20:23]).max(1)[alive]
        print(t, "alive", int(alive.sum()), "terminated so far", nterm, "pos err median %.2e p90 %.2e p99 %.2e max %.2e" % (np.median(e), np.percentile(e, 90), np.percentile(e, 99), e.max()))
