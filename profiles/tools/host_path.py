"""Where a small-N step goes: host enqueue time vs GPU time (python profiles/tools/host_path.py [N]).
Times (a) env.step(actions) as bench.py calls it, (b) the bare ctypes call with prebuilt arguments,
each as host-only enqueue time (no sync) and as wall time including the final sync."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phoenix_drone_simulation_amd as pds

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = 4096
env = pds.make("DroneHoverSimpleEnv-v0", num_envs=N, seed=0, observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0.0)
env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
ring = [(-0.1 + 0.25 * torch.randn(N, 4, device="cuda", generator=g)).contiguous() for _ in range(8)]
for s in range(256):
    env.step(ring[s % 8])
torch.cuda.synchronize()

def timed(fn, label):
    best_h, best_w = 1e9, 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(S):
            fn(s)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best_h, best_w = min(best_h, (t1 - t0) / S), min(best_w, (t2 - t0) / S)
    print(f"N={N:7d} {label:44s} host enqueue {best_h * 1e6:6.2f} us/step   wall {best_w * 1e6:6.2f} us/step", flush=True)

timed(lambda s: env.step(ring[s & 7]), "env.step(actions)")
lib, h = env.lib, env._handle
stream = torch.cuda.current_stream(env.device).cuda_stream
ptrs = [r.data_ptr() for r in ring]
bufs = [env._bufs[0]["_args"], env._bufs[1]["_args"]]
f = lib.pds_step_with_variates
timed(lambda s: f(h, ptrs[s & 7], None, *bufs[s & 1], stream), "bare ctypes pds_step_with_variates")
