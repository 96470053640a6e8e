import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import golden_util as gu
from test_gpu_parity import _make, _inject
from test_gpu_noise import _inject_noise_state
name = sys.argv[1] if len(sys.argv) > 1 else "circle_defaults"
g = gu.Golden(name)
pre = {k: [] for k in ("xyz", "rpy", "quat", "xyz_dot", "rpy_dot", "x", "act_hist", "iteration", "ref_offset",
                       "dt", "m", "J", "ftf1", "A", "K", "ou", "gyro_bias", "lpf", "obs_hist")}
acts, variates, exp, idx = [], [], [], []
for ep in range(g.E):
    for t in range(g.n_valid(ep)):
        for k in pre:
            pre[k].append(g["reset_" + k][ep] if t == 0 else g["step_" + k][ep, t - 1])
        acts.append(g["actions"][ep, t]); variates.append(gu.step_noise_variates(g, ep, t)); exp.append(g["obs"][ep, t]); idx.append((ep, t))
pre = {k: np.array(v) for k, v in pre.items()}
env = _make(g, len(acts), auto_reset=False)
env.reset(); _inject(env, pre, 1)
_inject_noise_state(env, pre["ou"], pre["gyro_bias"], pre["lpf"], pre["obs_hist"][:, 1, :10])
obs, *_ = env.step(torch.tensor(np.array(acts), dtype=torch.float32), noise_variates=np.array(variates, dtype=np.float32))
err = np.abs(obs.cpu().numpy() - np.array(exp))
np.set_printoptions(precision=3, linewidth=200, suppress=True)
print("max err per column:", err.max(0))
bad = np.where(err.max(1) > 1e-4)[0]
print("bad rows", len(bad), "of", len(acts), [idx[b] for b in bad[:20]])
r = bad[0] if len(bad) else 0
print("got ", obs.cpu().numpy()[r]); print("want", np.array(exp)[r])
