"""Longer PPO runs on the three tasks (reference default env config) -- does the policy learn the task?"""
import sys, time, torch
sys.path.insert(0, '.')
import phoenix_drone_simulation_amd as pds
from phoenix_drone_simulation_amd.ppo import PPOTrainer
task, n, T, E = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
H = int(sys.argv[5]) if len(sys.argv) > 5 else 2          # observation_history_size (round 6: one launch per rollout for H != 2 too)
hid = int(sys.argv[6]) if len(sys.argv) > 6 else 50       # policy hidden width (experiments/04_*: 32 / 48 / 64)
env = pds.make(task, num_envs=n, seed=0, **({"observation_history_size": H} if H != 2 else {}))
tr = PPOTrainer(env, rollout_len=T, epochs=E, seed=0,
                ac_kwargs={"pi": {"hidden_sizes": (hid, hid), "activation": "relu"}, "val": {"hidden_sizes": (64, 64), "activation": "tanh"}})
task = f"{task} H={H} pi={hid}-{hid}"
t0 = time.time()
for e in range(E):
    i = tr.learn_one_epoch()
    if e % max(1, E // 15) == 0 or e == E - 1:
        print(f"{task} epoch {i['epoch']:4d} ep_ret {i['ep_ret']:9.2f} ep_len {i['ep_len']:6.1f} episodes {int(i['episodes']):7d} "
              f"loss_v {i['loss_v']:9.4f} noise {i['noise_std']:.3f} fps {i['fps']:.3e}", flush=True)
torch.cuda.synchronize()
print(f"{task}: {E * n * T} env-steps in {time.time() - t0:.1f} s incl. updates")
