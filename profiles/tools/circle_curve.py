import json, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import phoenix_drone_simulation_amd as pds
from phoenix_drone_simulation_amd.ppo import PPOTrainer
ref = json.load(open('tests/golden/learning_curve_circle.json'))
L = np.array([ref['curves'][str(s)]['EpLen/Mean'] for s in ref['seeds']]); R = np.array([ref['curves'][str(s)]['EpRet/Mean'] for s in ref['seeds']])
LV = np.array([ref['curves'][str(s)]['Loss/Value'] for s in ref['seeds']])
def run(seed, n, **kw):
    env = pds.make('DroneCircleSimpleEnv-v0', num_envs=n, seed=seed)
    tr = PPOTrainer(env, rollout_len=32000 // n, epochs=40, seed=seed, reset_each_rollout=True, **kw)
    tr.learn(); torch.cuda.synchronize()
    out = (np.array([r['ep_len'] for r in tr.log]), np.array([r['ep_ret'] for r in tr.log]), np.array([r['loss_v'] for r in tr.log]))
    env.close(); return out
res = {}
for name, n, kw, ns in (('n8', 8, {}, 16), ('n8_torch', 8, dict(fused=False), 4), ('n1', 1, {}, 6)):
    runs = [run(300 + s, n, **kw) for s in range(ns)]
    res[name] = [np.array([r[i] for r in runs]) for i in range(3)]
    print(name, 'done', flush=True)
print('epoch | ref EpLen (SE) EpRet LossV | ' + ' | '.join(f'{k} EpLen (SE) EpRet LossV' for k in res))
se = lambda x: x.std(0, ddof=1) / np.sqrt(x.shape[0])
for e in range(40):
    print(f'{e+1:3d} | {L[:,e].mean():6.1f} ({se(L)[e]:4.1f}) {R[:,e].mean():7.1f} {LV[:,e].mean():8.3f} | ' + ' | '.join(f'{v[0][:,e].mean():6.1f} ({se(v[0])[e]:4.1f}) {v[1][:,e].mean():7.1f} {v[2][:,e].mean():8.3f}' for v in res.values()))
