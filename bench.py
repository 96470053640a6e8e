#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the fused HIP step on DroneHoverSimpleEnv-v0, 2^20 envs per GPU.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the
driver launches it under torch.distributed.run (one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

A "step" is one lockstep pds_step over all envs of the rank (auto-reset ON, so TimeLimit truncations,
terminations and in-kernel Philox resets are inside the timed region).  Inputs (the action ring)
are resident in HBM before the timed region.  Envs shard across ranks with no data-path collective
(weak scaling: 2^20 envs per GPU); `--allgather-obs` adds the optional RCCL all-gather of the
observations for the single-policy layout.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def physical_cores():
    """Physical cores of the host as `lscpu` counts them (unique core/socket pairs); None if lscpu is absent."""
    import subprocess
    try:
        out = subprocess.run(["lscpu", "-p=CORE,SOCKET"], capture_output=True, text=True, timeout=10).stdout
        return len({l for l in out.splitlines() if l and not l.startswith("#")}) or None
    except Exception:
        return None


def cpu_baseline(task, kw, n_cpu, target_seconds=12.0):
    """Time the CPU oracle (C restatement, float32, OpenMP over envs) on a bounded sample of the
    same workload: the SAME number of envs as the GPU run, same action recipe, auto-reset on,
    as many steps as fit in ~12 s of wall clock.  Reported baseline only."""
    import numpy as np
    from oracle import oracle as po
    threads = po.lib().po_max_threads()
    orc = po.OracleBatch(task, n_cpu, precision="f32", nthreads=threads, **kw)
    orc.reset(0, 0)
    rs = np.random.RandomState(0)
    hover = -1.0 + 2.0 / 2.25
    acts = (hover + 0.1 * rs.standard_normal((8, n_cpu, 4))).astype(np.float32)
    orc.step(acts[0], seed=0, tick=1)  # first touch
    # timed by the clock, not by a pilot: the first steps after the first touch run several times faster than the
    # steady state on a 128-thread host (a 4-step pilot once sized an "8 s" sample that took 37 s)
    steps = 0
    t0 = time.perf_counter()
    while steps < 4 or (time.perf_counter() - t0 < target_seconds and steps < 8000):
        orc.step(acts[steps % 8], seed=0, tick=2 + steps)
        steps += 1
    dt = time.perf_counter() - t0
    # the 1-thread figure SURVEY 8(d) asks for beside the all-core one (about 3 s)
    n1 = 4096
    one_t = po.OracleBatch(task, n1, precision="f32", nthreads=1, **kw)
    one_t.reset(0, 0)
    a1 = np.ascontiguousarray(acts[:, :n1])
    one_t.step(a1[0], seed=0, tick=1)
    t0 = time.perf_counter()
    k1 = 0
    while time.perf_counter() - t0 < 3.0:
        one_t.step(a1[k1 % 8], seed=0, tick=2 + k1)
        k1 += 1
    dt1 = time.perf_counter() - t0
    phys = physical_cores()
    out = {"value": n_cpu * steps / dt, "unit": "env-steps/s", "cores": int(threads), "kind": "port",
           "physical_cores": phys,
           "sample": f"oracle/phoenix_oracle.c (C restatement of the reference, float32) + OpenMP on {threads} host threads "
                     f"(`cores`; lscpu: {phys} physical cores), env structs first-touched by the thread that steps them, "
                     f"{n_cpu} envs (= the GPU run's N) x {steps} steps ({dt:.1f} s), same config and action recipe, "
                     f"auto-reset on; value_1_thread: {n1} envs on 1 thread for 3 s",
           "value_1_thread": n1 * k1 / dt1}
    # the reference's own Python loop cannot travel to the GPU box; its rate was measured in the build
    # container (1 process, Bullet calls stubbed => upper bound) by oracle/refgen/gen_golden.py --rate
    ref = os.path.join(ROOT, "tests", "golden", "reference_cpu_rate.json")
    if os.path.exists(ref):
        r = json.load(open(ref))
        key = f"{task}_det" if kw.get("observation_noise", 1) <= 0 else f"{task}_defaults"
        if key in r:
            out["reference_python_env_steps_per_s"] = r[key]
            out["reference_python_note"] = "the reference's Python step loop, 1 process, build container (tests/golden/reference_cpu_rate.json), not this box"
    return out


def measure_traffic(argv_tail, timeout=90):
    """HBM bytes per launch of pds::step_kernel, measured NOW: two child runs of this very command under
    `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE need separate passes: MI355X_MICROARCH.md, rocprofv3 PMC slots),
    corrected as that guide's HBM section prescribes (FETCH_SIZE x2 on gfx950, both counters in KiB).
    Returns (bytes or None, description)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process already runs under a profiler"
    total = 0.0
    env = dict(os.environ, TMPDIR="/tmp")
    for counter, scale in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        out = tempfile.mkdtemp(prefix="pds_pmc_", dir="/tmp")
        cmd = [rp, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
               "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-traffic"] + argv_tail
        try:
            # own session: a timeout must take down rocprofv3 AND the python grandchild that holds the GPU
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                    start_new_session=True)
            try:
                _, err = proc.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                proc.wait()
                return None, f"rocprofv3 --pmc {counter}: timed out after {timeout} s (process group killed)"
            if proc.returncode != 0:
                tail = (err or b"").decode("utf-8", "replace").strip().splitlines()[-3:]
                return None, f"rocprofv3 --pmc {counter} exited with {proc.returncode}: {' | '.join(tail)[-300:]}"
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            vals = [float(r["Counter_Value"]) for f in files for r in csv.DictReader(open(f))
                    if r.get("Counter_Name") == counter and "step_kernel" in r.get("Kernel_Name", "")]
            if not vals:
                return None, f"no {counter} rows for pds::step_kernel"
            total += sum(vals) / len(vals) * 1024.0 * scale
        except Exception as e:  # profiler unavailable / refused on this box: the caller falls back to the recorded value
            return None, f"rocprofv3 --pmc {counter} failed: {type(e).__name__}: {e}"
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return total, ("measured in this run: child passes `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` of the same command "
                   "(30 steps), FETCH_SIZE x2 (gfx950), KiB -> bytes, mean over the step_kernel dispatches")


def self_launch(n):
    import socket
    import subprocess
    with socket.socket() as sk:  # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL / HIP IPC across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=None,
                    help="override the env count of the chosen config (default: the config's own, 2^20 for 0)")
    ap.add_argument("--total-envs", type=int, default=None,
                    help="STRONG scaling (SURVEY 8e: 2^23 total): the env count of the whole job, split into contiguous "
                         "global-id blocks of total/N envs per GPU; the line then says \"scaling\": \"strong\"")
    ap.add_argument("--task", default="hover", choices=["hover", "circle", "takeoff"])
    ap.add_argument("--config", type=int, default=0,
                    help="0: north-star headline (Hover 2^20/GPU); 2/3/4: BASELINE.json configs[1..3]; "
                         "6: Hover 2^20 with the reference's default noise + DR")
    ap.add_argument("--mode", default="eager", choices=["eager", "graph", "stepk"],
                    help="eager: one pds_step launch per step (the headline); graph: the same launches captured "
                         "in ONE hipGraph per ring pass and replayed; stepk: open-loop pds_step_k, --k steps per launch")
    ap.add_argument("--k", type=int, default=8, help="steps per launch for --mode stepk")
    ap.add_argument("--allgather-obs", nargs="?", const="rccl", default=None, choices=["rccl", "p2p"],
                    help="single-policy layout: gather every rank's observations after each step; rccl = "
                         "all_gather_into_tensor, p2p = direct stores into the peers' buffers (sharding.P2PObsGather)")
    ap.add_argument("--also-envs", type=int, nargs="*", default=None,
                    help="after the main measurement, time the same configuration at these env counts too (one GPU, eager "
                         "mode) and report them under roofline.other_sizes -- e.g. --config 6 --also-envs 2097152: 2^21 envs per "
                         "handle is the noisy kernels' sweet spot (DESIGN section 9)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not measure roofline.traffic with child rocprofv3 --pmc passes (the children pass this)")
    ap.add_argument("--same-device", action="store_true",
                    help="TEST ONLY: every rank uses cuda:0 and the control collectives run over gloo, so that the "
                         "multi-rank code path (incl. --allgather-obs p2p) can be exercised on a 1-GPU box; "
                         "the numbers of such a run mean nothing")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the collectives of the multi-GPU path even with ONE rank "
                         "(started under torch.distributed.run --nproc-per-node 1): proves on a 1-GPU box that librccl loads, "
                         "that the device_id binding works and that barrier / all_gather / all_gather_into_tensor / all_reduce "
                         "run on this ROCm; says nothing about xGMI")
    ap.add_argument("--no-auto-reset", action="store_true", help="diagnostic only: INVALID as a benchmark number")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # No launcher around us: start the N ranks ourselves (the role of the reference's mpi_fork,
        # utils/mpi_tools.py:47-99) -- as a CHILD torch.distributed.run, before this process has imported
        # anything that touches the GPU; never exec from here.  The child ranks print the JSON line; this
        # process relays stdout and the exit code.
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    import phoenix_drone_simulation_amd as pds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev  # device of the small control tensors (timings)
    if args.force_dist and "WORLD_SIZE" not in os.environ:
        raise SystemExit("--force-dist needs a launcher: python -m torch.distributed.run --nproc-per-node 1 bench.py --force-dist ...")
    multi = world > 1 or args.force_dist  # the collectives of the multi-rank path run
    if multi:
        if args.same_device:
            dist.init_process_group("gloo")
            cdev = torch.device("cpu")
            if args.allgather_obs == "rccl":
                raise SystemExit("--same-device has no RCCL: use --allgather-obs p2p")
        else:
            dist.init_process_group("nccl", device_id=dev)

    task = args.task
    n = 1 << 20
    kw = dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0)
    act_center_shift = 0.0
    if args.config == 2:
        task, n = "hover", 65536
    elif args.config == 3:
        task, n = "circle", 262144
        kw.update(use_motor_dynamics=True, domain_randomization=0.10)
    elif args.config == 4:
        task, n = "takeoff", 1 << 20
        kw.update(use_ground_effect=True)
        act_center_shift = 0.2
    elif args.config == 6:  # the reference's DEFAULT env config: sensor + thrust noise, 10 % DR
        task, n = "hover", 1 << 20
        kw = dict(observation_noise=1, domain_randomization=0.10, motor_thrust_noise=0.05)
    if args.envs_per_gpu is not None:
        n = args.envs_per_gpu
    scaling = "weak"
    if args.total_envs is not None:  # fixed total work: rank r owns global ids [r * total / N, (r + 1) * total / N)
        if args.envs_per_gpu is not None:
            raise SystemExit("--total-envs and --envs-per-gpu exclude each other")
        if args.total_envs % world:
            raise SystemExit(f"--total-envs {args.total_envs} is not a multiple of the {world} ranks")
        n = args.total_envs // world
        scaling = "strong"
    env_id = {"hover": "DroneHoverSimpleEnv-v0", "circle": "DroneCircleSimpleEnv-v0",
              "takeoff": "DroneTakeOffSimpleEnv-v0"}[task]
    env = pds.make(env_id, num_envs=n, device=dev, seed=0, env_id_base=rank * n,
                   auto_reset=not args.no_auto_reset, **kw)

    # action ring generated once ([T, N, 4], a = HOVER_ACTION + 0.1 N(0,1)), reused cyclically
    T = 64
    g = torch.Generator(device=dev)
    g.manual_seed(rank)
    hover = -1.0 + 2.0 / 2.25
    ring = (hover + act_center_shift) + 0.1 * torch.randn(T, n, 4, generator=g, device=dev, dtype=torch.float32)
    gather_mode = args.allgather_obs if multi else None
    gathered = torch.empty(world * n, env.obs_dim, device=dev) if gather_mode == "rccl" else None
    p2p = None
    if gather_mode == "p2p":
        ctl = dist.new_group(backend="gloo")  # host-side hand-shake of the P2P gather (IPC handles, per-step barrier)
        p2p = pds.P2PObsGather(n, env.obs_dim, dev, sync_group=ctl)

    def do_gather(obs):
        """-> the [world * n, D] observations of all ranks (valid until the next call)"""
        if gather_mode == "rccl":
            dist.all_gather_into_tensor(gathered, obs)
            return gathered
        if gather_mode == "p2p":
            return p2p.gather(obs)
        return None

    def one_step(s):
        out = env.step(ring[s % T])
        do_gather(out[0])
        return out

    env.reset()
    if args.mode != "eager":
        # diagnostic modes (never the headline `value` of the driver's default run): same env-steps, fewer launches
        if gather_mode is not None:
            raise SystemExit("--mode graph/stepk: no --allgather-obs")
        K = args.k if args.mode == "stepk" else T
        if args.steps % K or args.warmup % K:
            raise SystemExit(f"--steps and --warmup must be multiples of {K} in --mode {args.mode}")
        if args.mode == "graph":
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for s in range(T):
                    env.step(ring[s])
            launch = lambda j: graph.replay()
        else:
            assert T % K == 0
            launch = lambda j: env.step_k(ring[(j * K) % T:(j * K) % T + K])
        for j in range(args.warmup // K):
            launch(j)
    else:
        K = 1
        launch = one_step
        for s in range(args.warmup):
            one_step(s)

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # torch creates the HIP event at its first record(): do that outside the timed region
    ev0.record(); ev1.record()
    sync()
    t0 = time.perf_counter()
    ev0.record()  # same stream pds_step launches on (torch's current stream)
    for j in range(args.steps // K):
        launch(args.warmup // K + j)
    ev1.record()
    sync()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / (args.steps // K)  # average launch-to-launch duration on the stream
    per_rank_ms, gather_ms = None, None
    if multi:
        mine = torch.tensor([wall], device=cdev, dtype=torch.float64)
        allw = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allw, mine)
        per_rank_ms = [float(x.item()) / args.steps * 1e3 for x in allw]  # every rank's own wall time per step
        wall = max(float(x.item()) for x in allw)                          # the slowest rank defines the step
        if gather_mode is not None:  # the exchange alone, same buffers, timed separately (not part of `value`)
            reps = max(4, min(50, args.steps // 4))
            last = env.step(ring[0])[0]
            # the exchange is checked once, outside the timed region: this rank's rows and a peer's rows of
            # the gathered buffer against what the owners computed (the peer's block over the control group)
            full = do_gather(last)
            peer = torch.empty_like(last)
            blocks = [torch.empty_like(last, device=cdev) for _ in range(world)]
            dist.all_gather(blocks, last.to(cdev))
            nxt = (rank + 1) % world
            peer.copy_(blocks[nxt])
            if not (torch.equal(full[rank * n:(rank + 1) * n], last) and torch.equal(full[nxt * n:(nxt + 1) * n], peer)):
                raise SystemExit(f"rank {rank}: gathered observations differ from the owners' blocks")
            sync()
            t1 = time.perf_counter()
            for _ in range(reps):
                do_gather(last)
            sync()
            g = torch.tensor([(time.perf_counter() - t1) / reps * 1e3], device=cdev, dtype=torch.float64)
            dist.all_reduce(g, op=dist.ReduceOp.MAX)
            gather_ms = float(g.item())

    total_envs = n * world
    value = total_envs * args.steps / wall
    bytes_per = env.bytes_per_env_step if args.mode != "stepk" else env.bytes_per_env_step_k(K)
    launch_bytes = n * bytes_per * K  # algorithmic bytes of one launch (graph: one replay = K = 64 step launches)
    achieved = launch_bytes / (kernel_ms * 1e-3) / 1e9
    # HBM bytes per launch measured with the PMC counters (separate rocprofv3 passes,
    # profiles/run_profile.sh) for exactly this workload; null for workloads that were not profiled
    traffic, traffic_src = None, None
    if rank == 0 and world == 1 and args.mode == "eager" and not args.no_traffic and not args.no_auto_reset:
        tail = ["--config", str(args.config), "--task", args.task]
        if args.envs_per_gpu is not None:
            tail += ["--envs-per-gpu", str(args.envs_per_gpu)]
        traffic, traffic_src = measure_traffic(tail)
    if traffic is None and args.config == 0 and task == "hover" and n == (1 << 20) and not args.no_auto_reset and args.mode == "eager":
        why = traffic_src
        for name in ("r02_traffic_headline.json", "r01_traffic_headline.json"):
            tf = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tf):
                traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
                traffic_src = (f"profiles/{name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                               "(profiles/run_profile.sh), recorded, NOT measured in this run"
                               + (f" ({why})" if why else ""))
                break
    other_sizes = None
    if args.also_envs and world == 1 and args.mode == "eager":
        other_sizes = []
        for n2 in args.also_envs:
            env2 = pds.make(env_id, num_envs=n2, device=dev, seed=0, auto_reset=not args.no_auto_reset, **kw)
            ring2 = (hover + act_center_shift) + 0.1 * torch.randn(8, n2, 4, generator=g, device=dev, dtype=torch.float32)
            env2.reset()
            for s_ in range(args.warmup):
                env2.step(ring2[s_ % 8])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record(); torch.cuda.synchronize()
            e0.record()
            for s_ in range(args.steps):
                env2.step(ring2[s_ % 8])
            e1.record(); torch.cuda.synchronize()
            ms2 = e0.elapsed_time(e1) / args.steps
            other_sizes.append({"envs_per_gpu": n2, "avg_launch_ms": ms2,
                                "frac": n2 * env2.bytes_per_env_step / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS})
            env2.close()
            del ring2
    if rank == 0:
        line = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{env_id}, {n} envs per GPU lockstep, fp32, observation noise "
                                   f"{'on' if kw['observation_noise'] > 0 else 'off'}, thrust noise {kw['motor_thrust_noise']}, "
                                   f"domain randomisation {kw['domain_randomization']}, auto-reset on, "
                                   f"action ring [64,N,4] = hover{act_center_shift:+.1f} + 0.1*N(0,1)"
                                   + (f", all-gather of obs ({gather_mode})" if gather_mode else ""),
                       "envs_per_gpu": n, "obs_dim": env.obs_dim, "bytes_per_env_step": bytes_per,
                       "parallelism": f"env-shard x{world}, no data-path collective" if gather_mode is None
                       else f"env-shard x{world} + all-gather(obs, {gather_mode})"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         # the same fraction from the wall clock of the timed region (host launch gaps included)
                         "frac_wall": launch_bytes / K / (wall / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "kernel": "pds::step_kernel" if args.mode != "stepk" else "pds::step_k_kernel",
                         "avg_launch_ms": kernel_ms, "steps_per_launch": K, "mode": args.mode,
                         "algorithmic_bytes_per_launch": launch_bytes},
        }
        # what the job really ran on (for reading a scaling curve: ranks, devices this process saw, the collective backend)
        if other_sizes is not None:
            line["roofline"]["other_sizes"] = other_sizes
        line["ranks"] = world
        line["visible_devices"] = torch.cuda.device_count()
        line["device_name"] = torch.cuda.get_device_name(dev)
        line["collective_backend"] = (dist.get_backend() if multi else None)
        line["total_envs"] = total_envs
        if per_rank_ms is not None:
            line["per_rank_ms_per_step"] = per_rank_ms
        if gather_ms is not None:
            line["allgather_ms"] = gather_ms  # the exchange alone (max over ranks), for reading the scaling curve
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(task, {k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()}, n)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port",
                                        "sample": f"failed: {e}"}
        print(json.dumps(line))
    if p2p is not None:
        p2p.release()
    env.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
