"""Build libpds_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The translation units (the host API, one per (task, variant family) -- each instantiating its step /
K-step / reset kernel variants -- and the trainer kernels) are compiled in parallel and linked into
one shared library."""
import concurrent.futures
import hashlib
import os
import re
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_UNITS = ["pds_task_hover.hip", "pds_task_circle.hip", "pds_task_takeoff.hip", "pds_task_hover_lat.hip",
          "pds_task_circle_lat.hip", "pds_task_hover_pid.hip", "pds_task_circle_pid.hip", "pds_task_takeoff_lat.hip",
          "pds_mlp.hip", "pds_mlp_wide.hip", "pds_task_hover_hold.hip", "pds_task_circle_hold.hip", "pds_task_takeoff_hold.hip",
          "pds_task_hover_pid_ge.hip", "pds_task_circle_pid_ge.hip", "pds_rollout_hover_pwm.hip", "pds_rollout_circle_pwm.hip", "pds_rollout_hover_lat.hip", "pds_rollout_circle_lat.hip",
          "pds_api.hip", "pds_rollout_hover.hip", "pds_rollout_circle.hip", "pds_rollout_takeoff.hip",
          "pds_rollout_hist_hover.hip", "pds_rollout_hist_circle.hip", "pds_rollout_hist_takeoff.hip",
          "pds_gae.hip", "pds_train.hip", "pds_history.hip"]  # longest first
_HEADERS = ["pds_device.h", "pds_types.h", "pds_reset.h", "pds_step.h", "pds_mlp_fwd.h", "pds_rollout.h", "pds_rollout_hist.h", "pds_mlp_common.h"]
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


_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _closure(path, seen=None):
    """`path` and every file it includes with #include "..." (recursively), in a stable order."""
    seen = seen if seen is not None else []
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.append(path)
    with open(path) as f:
        text = f.read()
    for inc in _INC.findall(text):
        _closure(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def _signature(unit, flags):
    """sha256 over the flags and the CONTENTS of a translation unit and of the headers it pulls in: an object is rebuilt when
    what it was compiled from changed, not when a file was touched or a header it never sees was edited."""
    h = hashlib.sha256(" ".join(flags).encode())
    for path in _closure(os.path.join(_CSRC, unit)):
        h.update(os.path.relpath(path, _HERE).encode())  # (relative: the signature travels with the checkout)
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build_library(force=False, verbose=False, extra_flags=(), out=None):
    """hipcc --offload-arch=gfx950 -c csrc/*.hip (in parallel) -> link libpds_hip.so"""
    out = out or _LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(_OBJ, exist_ok=True)
    tag = "".join(c if c.isalnum() else "_" for c in "".join(extra_flags))
    flags = _FLAGS + list(extra_flags)
    objs = [os.path.join(_OBJ, os.path.splitext(u)[0] + tag + ".o") for u in _UNITS]
    sigs = {u: _signature(u, flags) for u in _UNITS}

    def current(unit, obj):
        sig = obj + ".sig"
        return os.path.exists(obj) and os.path.exists(sig) and open(sig).read().strip() == sigs[unit]

    # the library carries the signature of everything it was built from (libpds_hip.so.sig travels with it to the GPU box,
    # the objects under build/ do not): up to date == same sources, whatever the time stamps say
    link_sig = hashlib.sha256("".join(sigs[u] for u in _UNITS).encode()).hexdigest()
    lsig = out + ".sig"
    if not force and os.path.exists(out):
        if os.path.exists(lsig):
            if open(lsig).read().strip() == link_sig:
                return out
        elif not extra_flags and not _stale(out, _DEPS):
            return out  # (a library without its signature file: trust the time stamps)
    todo = [(u, o) for u, o in zip(_UNITS, objs) if force or not current(u, o)]
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libpds_hip.so (there is no CPU fallback)")

    def compile_unit(pair):
        unit, obj = pair
        cmd = [hipcc] + flags + ["-c", os.path.join(_CSRC, unit), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(obj + ".sig", "w") as f:
            f.write(sigs[unit])

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        list(ex.map(compile_unit, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    with open(lsig, "w") as f:
        f.write(link_sig)
    return out
