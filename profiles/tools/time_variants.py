"""Time pds_step for a list of env configurations on one GPU (us/step, % of the 8 TB/s HBM peak by the
algorithmic bytes): python profiles/tools/time_variants.py [N] [steps].  PDS_LIB selects another build
of the library for same-box A/B runs."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phoenix_drone_simulation_amd as pds

IDS = {"hover": "DroneHoverSimpleEnv-v0", "circle": "DroneCircleSimpleEnv-v0", "takeoff": "DroneTakeOffSimpleEnv-v0"}
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8  # also time pds_step_k with K steps per launch (0: skip)
OFF = dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0.0)
CASES = [
    ("hover   lean", "hover", OFF),
    ("hover   on", "hover", dict(OFF, observation_noise=1)),
    ("hover   on dr tn (reference default)", "hover", dict()),
    ("hover   pt1 dr", "hover", dict(OFF, use_motor_dynamics=True, domain_randomization=0.1)),
    ("hover   pt1 dr on tn", "hover", dict(use_motor_dynamics=True)),
    ("circle  on dr tn (reference default)", "circle", dict()),
    ("circle  pt1 dr on tn", "circle", dict(use_motor_dynamics=True)),
    ("takeoff on dr tn (reference default)", "takeoff", dict()),
    ("takeoff ge on dr tn", "takeoff", dict(use_ground_effect=True)),
    ("hover   latency 0.02 on dr tn", "hover", dict(use_latency=True, latency=0.02)),
    ("hover   obs 50 Hz on dr tn", "hover", dict(observation_frequency=50)),
    ("hover   AttitudeRate PID lean", "hover", dict(OFF, control_mode="AttitudeRate")),
    ("hover   Attitude PID lean", "hover", dict(OFF, control_mode="Attitude")),
    ("hover   Attitude PID on dr tn", "hover", dict(control_mode="Attitude")),
    ("circle  AttitudeRate PID pt1 dr on tn", "circle", dict(control_mode="AttitudeRate", use_motor_dynamics=True)),
    ("hover   latency 0.02 lean", "hover", dict(OFF, use_latency=True, latency=0.02)),
    ("circle  latency 0.03 pt1 dr on tn", "circle", dict(use_latency=True, latency=0.03, use_motor_dynamics=True)),
    ("hover   agg 2 lean", "hover", dict(OFF, aggregate_phy_steps=2)),
    ("hover   agg 2 on dr tn", "hover", dict(aggregate_phy_steps=2)),
]
g = torch.Generator(device="cuda").manual_seed(0)
acts = [(-0.1 + 0.25 * torch.randn(N, 4, device="cuda", generator=g)).contiguous() for _ in range(8)]
for name, task, kw in CASES:
    env = pds.make(IDS[task], num_envs=N, seed=0, **kw)
    env.reset()
    for s in range(60):
        env.step(acts[s % 8])
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for s in range(STEPS):
            env.step(acts[s % 8])
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / STEPS)
    b = env.bytes_per_env_step
    line = f"{name:40s} N={N:8d}  {best * 1e6:8.2f} us/step   {b:4d} B/env-step  {100 * b * N / best / 8e12:5.1f} % of 8 TB/s"
    if K > 0:
        ak = torch.stack(acts[:K]) if K <= 8 else torch.stack([acts[j % 8] for j in range(K)])
        try:
            for _ in range(5):
                env.step_k(ak)
            torch.cuda.synchronize()
            bk = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                for _ in range(max(STEPS // K, 4)):
                    env.step_k(ak)
                torch.cuda.synchronize()
                bk = min(bk, (time.perf_counter() - t0) / (max(STEPS // K, 4) * K))
            kb = env.bytes_per_env_step_k(K)
            line += f"   | step_k K={K}: {bk * 1e6:7.2f} us/env-step  {kb:4d} B  {100 * kb * N / bk / 8e12:5.1f} %"
        except NotImplementedError as e:
            line += f"   | step_k: {e}"
    print(line, flush=True)
    env.close()
