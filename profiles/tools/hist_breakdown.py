"""Where a PPO epoch goes with an observation history other than 2 (profiles/tools/hist_breakdown.py N T task H): the rollout as
ONE launch (pds_rollout_history, csrc/pds_rollout_hist.h; round 6) against the per-step kernels (8 launches per step), and the
update on the K-tiled fused kernels (csrc/pds_mlp_wide.hip; round 6) against PyTorch ops (what rounds 1-5 ran beyond 64 inputs)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phoenix_drone_simulation_amd as pds
from phoenix_drone_simulation_amd.ppo import PPOTrainer
N, T = int(sys.argv[1]), int(sys.argv[2])
task = sys.argv[3] if len(sys.argv) > 3 else "DroneHoverSimpleEnv-v0"
H = int(sys.argv[4]) if len(sys.argv) > 4 else 4
def sync(): torch.cuda.synchronize()
for label, kw in (("one launch per rollout + fused update", dict()),
                  ("per-step kernels + fused update", dict(fused_rollout=False)),
                  ("per-step env, PyTorch-op networks (rounds 1-5 beyond 64 inputs)", dict(fused=False))):
    env = pds.make(task, num_envs=N, seed=0, observation_history_size=H)
    tr = PPOTrainer(env, rollout_len=T, epochs=10, **kw)
    for ep in range(3):
        tr.ac.update(frac=ep / 10)
        sync(); t0 = time.time(); tr.roll_out(); sync(); t1 = time.time(); tr.update(); sync(); t2 = time.time()
        if ep:
            print(f"{task} H {H} ({env.obs_dim} inputs) N {N} T {T} [{label}; fused {tr.fused}, fused_rollout {tr.fused_rollout}] epoch {ep}: "
                  f"rollout {1e3*(t1-t0):8.2f} ms ({1e6*(t1-t0)/T:7.1f} us/step)  update {1e3*(t2-t1):8.1f} ms  -> {N*T/(t2-t0):.3e} env-steps/s", flush=True)
    env.close()
