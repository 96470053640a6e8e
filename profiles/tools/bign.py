import sys, time, torch
sys.path.insert(0, '.')
import phoenix_drone_simulation_amd as pds
n = int(sys.argv[1]); T = int(sys.argv[2])
env = pds.make("DroneHoverSimpleEnv-v0", num_envs=n, seed=0, observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0)
ring = -0.11 + 0.1 * torch.randn(T, n, 4, device='cuda')
env.reset()
for s in range(20): env.step(ring[s % T])
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for s in range(200): env.step(ring[s % T])
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 200
print(n, T, round(ms * 1000, 1), "us", round(n * 346 / ms / 1e6 / 8000, 4))
