"""Where a PPO epoch goes: rollout vs update (profiles/tools/ppo_breakdown.py N T [task]); the rollout both ways:
ONE launch (pds_rollout, csrc/pds_rollout.h) and the per-step kernels replayed from a hipGraph (round 2)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phoenix_drone_simulation_amd as pds
from phoenix_drone_simulation_amd.ppo import PPOTrainer
N, T = int(sys.argv[1]), int(sys.argv[2])
task = sys.argv[3] if len(sys.argv) > 3 else "DroneHoverSimpleEnv-v0"
def sync(): torch.cuda.synchronize()
for label, fr in (("one launch per rollout", None), ("7 launches per step, hipGraph", False)):
    env = pds.make(task, num_envs=N, seed=0)
    tr = PPOTrainer(env, rollout_len=T, epochs=10, fused_rollout=fr)
    for ep in range(3):
        tr.ac.update(frac=ep / 10)
        sync(); t0 = time.time(); tr.roll_out(); sync(); t1 = time.time(); tr.update(); sync(); t2 = time.time()
        if ep:
            print(f"N {N} T {T} [{label}; fused_rollout={tr.fused_rollout}] epoch {ep}: rollout {1e3*(t1-t0):8.2f} ms ({1e6*(t1-t0)/T:7.1f} us/step)"
                  f"  update {1e3*(t2-t1):8.1f} ms  -> {N*T/(t2-t0):.3e} env-steps/s", flush=True)
    env.close()
