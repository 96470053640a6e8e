"""Where a closed-loop step of pds_rollout goes, per wave role (diagnostic build of the library:
    python -c "import phoenix_drone_simulation_amd as p; p.build_library(extra_flags=('-DPDS_ROLLOUT_TIMING', '-DPDS_STAMPS'), out='/tmp/libpds_timing.so')"
    PDS_LIB=/tmp/libpds_timing.so python profiles/tools/rollout_timing.py
s_memtime differences accumulated over the rollout by block 0: the env wave's wait for the action / step / hand-over and
the three phases of its step (csrc/pds_step.h stamps 3-5), the four network waves' noise draw / wait for o(t) / row gather /
actor pass / hand-over / shadow work (critic passes, V(final_obs), buffer writes).  The kernel writes them behind the
three statistics words, so `stats` gets room for 64 floats here."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phoenix_drone_simulation_amd as pds
from phoenix_drone_simulation_amd.ppo import PPOTrainer
keep = []
_zeros = torch.zeros
def zeros(*a, **k):
    if a == (3,):
        big = _zeros(128, **k); keep.append(big); return big[:3]
    return _zeros(*a, **k)
torch.zeros = zeros
for kw in (dict(), dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0)):
    env = pds.make("DroneHoverSimpleEnv-v0", num_envs=8192, seed=0, **kw)
    tr = PPOTrainer(env, rollout_len=64, epochs=10)
    for ep in range(3):
        tr.ac.update(frac=ep / 10)
        torch.cuda.synchronize(); t0 = time.time(); st = tr.roll_out(); torch.cuda.synchronize(); dt = time.time() - t0
    raw = keep[-1].cpu().tolist()
    print(kw, "rollout %.3f ms" % (dt * 1e3))
    print(" E wave cycles per step: wait action %.0f  step %.0f  post %.0f | step = row half 1 + physics %.0f, reward + row half 2 + stores %.0f, final_obs + reset + flush %.0f (final_obs copies %.0f, reset %.0f)" % tuple(raw[8:16]))
    if raw[48] or raw[49]: print("   reset = cooperative Philox + conversion %.0f, evaluation + rows %.0f (-DPDS_STAMPS_RESET)" % (raw[48], raw[49]))
    for w in range(4):
        o = raw[16 + 8 * w: 16 + 8 * w + 6]
        print(" M%d cycles per step: pre (noise draw) %.0f  wait obs %.0f  gather %.0f  actor+sample %.0f  post %.0f  critic etc %.0f" % (w, *o))
    env.close()
