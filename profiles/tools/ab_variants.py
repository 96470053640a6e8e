"""Same-box A/B of pds_step timings between library builds / env-var settings for a few env configurations:
   python profiles/tools/ab_variants.py  (GPU box; PDS_LIB / PDS_STORED_OH_FROM_AGG are set per child process)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import os, sys, time, json, torch
sys.path.insert(0, %r)
import phoenix_drone_simulation_amd as pds
IDS = {"hover": "DroneHoverSimpleEnv-v0", "circle": "DroneCircleSimpleEnv-v0", "takeoff": "DroneTakeOffSimpleEnv-v0"}
task, kw, N, steps = json.loads(sys.argv[1])
g = torch.Generator(device="cuda").manual_seed(0)
acts = [(-0.1 + 0.25 * torch.randn(N, 4, device="cuda", generator=g)).contiguous() for _ in range(8)]
env = pds.make(IDS[task], num_envs=N, seed=0, auto_reset=os.environ.get("AB_AUTO_RESET", "1") != "0", **kw)
env.reset()
for s in range(60):
    env.step(acts[s %% 8])
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for s in range(steps):
        env.step(acts[s %% 8])
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / steps)
print(json.dumps(dict(us=best * 1e6, bytes=env.bytes_per_env_step)))
""" % ROOT

CASES = [
    ("hover default (config 6)", "hover", {}),
    ("hover noise only", "hover", dict(domain_randomization=-1, motor_thrust_noise=0.0)),
    ("circle pt1 default", "circle", dict(use_motor_dynamics=True)),
    ("hover agg 2 default", "hover", dict(aggregate_phy_steps=2)),
    ("hover agg 4 default", "hover", dict(aggregate_phy_steps=4)),
    ("hover lean (headline)", "hover", dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0.0)),
    ("circle default", "circle", dict()),
    ("hover latency 0.02 default", "hover", dict(use_latency=True, latency=0.02)),
    ("hover obs 50 Hz default", "hover", dict(observation_frequency=50)),
    ("hover Attitude PID default", "hover", dict(control_mode="Attitude")),
    ("hover latency 0.02 lean", "hover", dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0.0, use_latency=True, latency=0.02)),
]
if os.environ.get("AB_SIGMA"):  # the action recipe: a = -0.1 + sigma * N(0, 1); 0.25 (default) ends ~5 % of the episodes per step, bench.py's 0.1 ~2 %
    CHILD = CHILD.replace("0.25 * torch.randn", os.environ["AB_SIGMA"] + " * torch.randn")


def main():
    libs = sys.argv[1:] or ["libpds_hip.so"]
    settings = []
    for lib in libs:
        settings.append((lib, {}))
    if os.environ.get("AB_SPLIT"):  # round 6: in-place reset (PDS_SPLIT_RESET=0) against SplitReset step kernel + post_reset_kernel
        settings = [(libs[0] + " in-place-reset", {"PDS_SPLIT_RESET": "0"})] + [(l + " split-reset", {"PDS_SPLIT_RESET": "1"}) for l in libs]
    if os.environ.get("AB_STORED"):  # (builds with PDS_STORED_OH_FROM_AGG > 0 only)
        settings.append((libs[-1] + " regen@agg>=2", {"PDS_STORED_OH_FROM_AGG": "0"}))
    N, steps = 1 << 20, 300
    for name, task, kw in CASES:
        for rep in range(2):
            row = []
            for label, env in settings:
                e = dict(os.environ, PDS_LIB=os.path.join(ROOT, "phoenix-drone-simulation_amd", label.split()[0]), **env)
                out = subprocess.run([sys.executable, "-c", CHILD, json.dumps([task, kw, N, steps])], env=e, capture_output=True, text=True)
                try:
                    d = json.loads(out.stdout.strip().splitlines()[-1])
                    row.append(f"{label}: {d['us']:7.2f} us ({d['bytes']} B)")
                except Exception:
                    row.append(f"{label}: failed {out.stderr[-200:]}")
            print(f"{name:28s} | " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
