"""Learning curve + throughput of the on-device PPO caller (profiles/r01_ppo_curve.txt)."""
import sys, time, torch
sys.path.insert(0, '.')
import phoenix_drone_simulation_amd as pds
from phoenix_drone_simulation_amd.ppo import PPOTrainer
n, T, E = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
env = pds.make("DroneHoverSimpleEnv-v0", num_envs=n, seed=0)   # reference default config (noise + DR)
tr = PPOTrainer(env, rollout_len=T, epochs=E, seed=0)
t0 = time.time()
for e in range(E):
    i = tr.learn_one_epoch()
    if e % max(1, E // 12) == 0 or e == E - 1:
        print(f"epoch {i['epoch']:3d} ep_ret {i['ep_ret']:9.2f} ep_len {i['ep_len']:6.1f} episodes {int(i['episodes']):7d} "
              f"loss_v {i['loss_v']:9.3f} noise {i['noise_std']:.3f} fps {i['fps']:.3e}")
torch.cuda.synchronize()
print(f"total {E * n * T} env-steps in {time.time() - t0:.1f} s incl. updates")
