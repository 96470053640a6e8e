#!/usr/bin/env python3
"""Per-wave timeline of pds::step_kernel from s_memtime stamps (diagnostic build -DPDS_STAMPS,
libpds_hip_stamps.so; select it with PDS_LIB).  Splits a step into: kernel entry -> loads issued ->
loads arrived -> physics done -> outputs issued -> observation tile flushed -> reset drain done ->
all stores acknowledged; plus the spread of the waves' start times (dispatch ramp) from s_memrealtime.

  PDS_LIB=phoenix-drone-simulation_amd/libpds_hip_stamps.so python profiles/tools/stamps.py --config 2
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--envs", type=int, default=None)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--no-auto-reset", action="store_true")
    args = ap.parse_args()
    import torch
    import phoenix_drone_simulation_amd as pds
    kw = dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0)
    task, n = "hover", 1 << 20
    if args.config == 2:
        n = 65536
    elif args.config == 3:
        task, n = "circle", 262144
        kw.update(use_motor_dynamics=True, domain_randomization=0.10)
    elif args.config == 4:
        task, n = "takeoff", 1 << 20
        kw.update(use_ground_effect=True)
    elif args.config == 6:  # the reference's default env config
        kw = dict(observation_noise=1, domain_randomization=0.10, motor_thrust_noise=0.05)
    if args.envs:
        n = args.envs
    env_id = {"hover": "DroneHoverSimpleEnv-v0", "circle": "DroneCircleSimpleEnv-v0", "takeoff": "DroneTakeOffSimpleEnv-v0"}[task]
    env = pds.make(env_id, num_envs=n, seed=0, auto_reset=not args.no_auto_reset, **kw)
    lib = env.lib
    lib.pds_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    assert lib.pds_debug_stamps(env._handle, None, 0) == 0, "not a -DPDS_STAMPS build (set PDS_LIB)"
    dev = env.device
    g = torch.Generator(device=dev); g.manual_seed(0)
    ring = (-1.0 + 2.0 / 2.25) + 0.1 * torch.randn(64, n, 4, generator=g, device=dev)
    env.reset()
    ntiles = (n + 63) // 64
    S = 10
    acc = []
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for s in range(args.steps):
        if s == args.steps // 2:
            ev0.record()
        env.step(ring[s % 64])
        if s >= args.steps - 8:  # the stamps of the last launches, one readback each (synchronises)
            buf = np.zeros(ntiles * S, np.uint64)
            got = lib.pds_debug_stamps(env._handle, buf.ctypes.data_as(C.c_void_p), buf.size)
            assert got == ntiles
            acc.append(buf.reshape(ntiles, S).astype(np.int64))
    ev1.record(); torch.cuda.synchronize()
    st = np.stack(acc)  # [launch, tile, slot]
    t = st[:, :, :8]
    names = ["entry -> loads issued (kernarg loads, address math)", "loads issued -> all arrived (vmcnt 0)",
             "o(k) row half + physics sub-steps", "task, o(k+1) row half, state + reward stores issued",
             "LDS tile: final_obs copy, row rewrite, flush issued", "deferred reset drain (0 when merged)",
             "all stores acknowledged (vmcnt 0)"]
    print(f"config {args.config}: {env_id} {n} envs, {ntiles} waves; s_memtime cycles per phase (median / p10 / p90 over waves and 8 launches)")
    tot = (t[:, :, 7] - t[:, :, 0])
    for j, nm in enumerate(names):
        d = (t[:, :, j + 1] - t[:, :, j]).reshape(-1)
        print(f"  {nm:58s} {np.median(d):8.0f} {np.percentile(d, 10):8.0f} {np.percentile(d, 90):8.0f}   {100 * np.median(d) / np.median(tot):5.1f} %")
    print(f"  {'wave lifetime (entry -> stores acknowledged)':58s} {np.median(tot):8.0f} {np.percentile(tot, 10):8.0f} {np.percentile(tot, 90):8.0f}")
    rt0, rt1 = st[:, :, 8], st[:, :, 9]  # 100 MHz
    start = (rt0 - rt0.min(axis=1, keepdims=True)) * 10.0  # ns after the first wave of the launch
    end = (rt1 - rt0.min(axis=1, keepdims=True)) * 10.0
    cyc_per_ns = np.median(tot / np.maximum((rt1 - rt0) * 10.0, 1.0))
    print(f"  shader clock during the kernel: {cyc_per_ns:.2f} GHz (s_memtime / s_memrealtime)")
    print(f"  wave start after the launch's first wave: median {np.median(start):.0f} ns, p90 {np.percentile(start, 90):.0f} ns, last {start.max(axis=1).mean():.0f} ns")
    print(f"  last wave end after the first wave's start: {end.max(axis=1).mean():.0f} ns  (kernel span seen from inside)")
    print(f"  launch-to-launch on the stream (hipEvents, incl. this tool's readbacks: upper bound): {ev0.elapsed_time(ev1) / (args.steps - args.steps // 2) * 1e3:.2f} us")
    env.close()


if __name__ == "__main__":
    main()
