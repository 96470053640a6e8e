"""Build libpds_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The translation units (the host API, one per (task, variant family) -- each instantiating its step /
K-step / reset kernel variants -- and the trainer kernels) are compiled in parallel and linked into
one shared library."""
import concurrent.futures
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_UNITS = ["pds_task_hover.hip", "pds_task_circle.hip", "pds_task_takeoff.hip", "pds_task_hover_lat.hip",
          "pds_task_circle_lat.hip", "pds_task_hover_pid.hip", "pds_task_circle_pid.hip", "pds_task_takeoff_lat.hip",
          "pds_mlp.hip", "pds_task_hover_hold.hip", "pds_task_circle_hold.hip", "pds_task_takeoff_hold.hip",
          "pds_rollout_hover_pwm.hip", "pds_rollout_circle_pwm.hip", "pds_rollout_hover_lat.hip", "pds_rollout_circle_lat.hip",
          "pds_api.hip", "pds_rollout_hover.hip", "pds_rollout_circle.hip", "pds_rollout_takeoff.hip",
          "pds_gae.hip", "pds_train.hip", "pds_history.hip"]  # longest first
_HEADERS = ["pds_device.h", "pds_types.h", "pds_reset.h", "pds_step.h", "pds_mlp_fwd.h", "pds_rollout.h"]
_DEPS = [os.path.join(_CSRC, f) for f in _UNITS + _HEADERS] + [os.path.join(_HERE, "..", "include", "pds.h")]
_LIB = os.path.join(_HERE, "libpds_hip.so")
_OBJ = os.path.join(_HERE, "build")
# -ffp-contract=on: fuse a*b+c only where one source expression says so (the frontend decides), not
# wherever the optimiser finds a multiply next to an add (hipcc's default "fast" makes that depend
# on inlining and use counts, i.e. differ between template instantiations of the same code): the
# 32-row and 64-row tile kernels, and hence any sharding of a batch, must agree bitwise.
_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-ffp-contract=on"]


def library_path():
    # PDS_LIB: A/B-test another build of the same sources (profiling only)
    return os.environ.get("PDS_LIB", _LIB)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, extra_flags=(), out=None):
    """hipcc --offload-arch=gfx950 -c csrc/*.hip (in parallel) -> link libpds_hip.so"""
    out = out or _LIB
    if not force and not extra_flags and not _stale(out, _DEPS):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libpds_hip.so (there is no CPU fallback)")
    os.makedirs(_OBJ, exist_ok=True)
    tag = "".join(c if c.isalnum() else "_" for c in "".join(extra_flags))
    objs = [os.path.join(_OBJ, os.path.splitext(u)[0] + tag + ".o") for u in _UNITS]

    def compile_unit(pair):
        unit, obj = pair
        if not force and not _stale(obj, _DEPS):
            return
        cmd = [hipcc] + _FLAGS + list(extra_flags) + ["-c", os.path.join(_CSRC, unit), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        list(ex.map(compile_unit, zip(_UNITS, objs)))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out
