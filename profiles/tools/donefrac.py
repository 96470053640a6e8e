import torch, sys
sys.path.insert(0, '.')
import phoenix_drone_simulation_amd as pds
n = 1 << 20
env = pds.make("DroneHoverSimpleEnv-v0", num_envs=n, seed=0, observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0)
g = torch.Generator(device='cuda'); g.manual_seed(0)
hover = -1.0 + 2.0/2.25
env.reset()
for s in range(300):
    a = hover + 0.1*torch.randn(n, 4, generator=g, device='cuda')
    o, r, t, tr, info = env.step(a)
    if s % 25 == 0 or s > 295:
        d = (t | tr)
        waves = d.view(-1, 64).any(1).float().mean().item()
        print(s, 'done frac %.5f' % d.float().mean().item(), 'trunc %.5f' % tr.float().mean().item(), 'waves with reset %.3f' % waves)
