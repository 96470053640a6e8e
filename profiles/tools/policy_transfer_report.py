import json, os, sys, numpy as np
sys.path.insert(0, '.')
from scipy import stats
import phoenix_drone_simulation_amd as pds
from phoenix_drone_simulation_amd.evaluation import evaluate
from phoenix_drone_simulation_amd.ppo import ActorCritic
ref = json.load(open('tests/golden/policy_eval_stats.json'))
print("# HIP-trained policies (tests/golden/hip_policy_{early,late}.npz) played deterministically: the REFERENCE's DroneHoverSimpleEnv-v0 (CPU, numpy randomness;")
print("# oracle/refgen/gen_golden_policy_stats.py) vs the HIP envs (Philox, 8 192 episodes); env defaults (sensor + thrust noise, 10 % DR, reset distribution)")
for name in ("early", "late", "circle_attrate", "hover_latency_motor", "hover_hold", "circle_default", "hover_history4"):
    sd = np.load('tests/golden/hip_policy_circle_attrate_late.npz' if name == "circle_attrate" else f'tests/golden/hip_policy_{name}.npz')
    env = pds.make(ref[name].get("env_id", "DroneHoverSimpleEnv-v0"), num_envs=8192, seed=5, **ref[name].get("env_kwargs", {}))
    ac = ActorCritic.from_reference_state_dict({k: sd[k] for k in sd.files}).to(env.device)
    ret, length, _ = evaluate(env, ac)
    ret, length = ret.numpy().astype(float), length.numpy().astype(float)
    rl, rr = np.array(ref[name]["ep_len"], float), np.array(ref[name]["ep_ret"], float)
    for mine, theirs, what in ((length, rl, "episode length"), (ret, rr, "episode return")):
        t, p = stats.ttest_ind(mine, theirs, equal_var=False)
        print(f"{name:14s} {what:15s}: reference {theirs.mean():9.2f} +- {theirs.std()/np.sqrt(len(theirs)):5.2f} (SD {theirs.std():6.2f}, {len(theirs)} episodes) | HIP {mine.mean():9.2f} +- {mine.std()/np.sqrt(len(mine)):5.2f} (SD {mine.std():6.2f}) | Welch t {t:+.2f} p {p:.3f}")
    print(f"{name:5s} terminated share: reference {np.mean(ref[name]['terminated']):.4f} | HIP {(length < 500).mean():.4f}")
    if name == "late":
        print("# a HIP-trained policy on exp-07's AttitudeRate configuration (PID rate loop under the policy, 4 sub-steps per step), Circle task:")
    env.close()
from phoenix_drone_simulation_amd.policy_io import load_network_json
r = ref["circle_reference_policy"]
env = pds.make(r["env_id"], num_envs=4096, seed=6, **r["env_kwargs"])
pol = load_network_json('tests/golden/policy_PWM_seed_00000_model.json').to(env.device)
ret, length, _ = evaluate(env, pol)
ret, length = ret.numpy().astype(float), length.numpy().astype(float)
print("# a policy trained BY THE REFERENCE (exp-07 PWM, bundled JSON) in DroneCircleSimpleEnv-v0 at that experiment's settings (2 sub-steps, noise, 10 % DR):")
for mine, theirs, what in ((length, np.array(r["ep_len"], float), "episode length"), (ret, np.array(r["ep_ret"], float), "episode return")):
    t, p = stats.ttest_ind(mine, theirs, equal_var=False)
    print(f"circle {what:15s}: reference {theirs.mean():9.2f} +- {theirs.std()/np.sqrt(len(theirs)):5.2f} (SD {theirs.std():6.2f}, {len(theirs)} episodes) | HIP {mine.mean():9.2f} +- {mine.std()/np.sqrt(len(mine)):5.2f} (SD {mine.std():6.2f}) | Welch t {t:+.2f} p {p:.3f}")
print(f"circle terminated share: reference {np.mean(r['terminated']):.4f} | HIP {(length < 500).mean():.4f}")
import torch
r = ref["takeoff_const"]
env = pds.make(r["env_id"], num_envs=8192, seed=8, **r["env_kwargs"])
a = torch.full((env.num_envs, 4), -1.0 + 2.0 / 2.25 + r["action_offset"], device=env.device)
ret, length, _ = evaluate(env, lambda obs: a)
ret, theirs = ret.numpy().astype(float), np.array(r["ep_ret"], float)
t, p = stats.ttest_ind(ret, theirs, equal_var=False)
print("# DroneTakeOffSimpleEnv-v0 at its defaults under the open-loop command HOVER_ACTION + 0.04 (500 steps, no termination):")
print(f"takeoff 500-step return: reference {theirs.mean():9.1f} +- {theirs.std()/np.sqrt(len(theirs)):5.1f} (SD {theirs.std():7.1f}, {len(theirs)} episodes) | HIP {ret.mean():9.1f} +- {ret.std()/np.sqrt(len(ret)):5.1f} (SD {ret.std():7.1f}) | Welch t {t:+.2f} p {p:.3f}")
print("takeoff return quantiles 10/25/50/75/90 %: reference", np.round(np.quantile(theirs, [.1,.25,.5,.75,.9]), 0), "| HIP", np.round(np.quantile(ret, [.1,.25,.5,.75,.9]), 0))
