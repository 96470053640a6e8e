"""Build libpds_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "pds_kernels.hip")
_DEPS = [_SRC, os.path.join(_HERE, "csrc", "pds_device.h"),
         os.path.join(_HERE, "..", "include", "pds.h")]
_LIB = os.path.join(_HERE, "libpds_hip.so")


def library_path():
    # PDS_LIB: A/B-test another build of the same sources (profiling only)
    return os.environ.get("PDS_LIB", _LIB)


def _stale():
    if not os.path.exists(_LIB):
        return True
    t = os.path.getmtime(_LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in _DEPS)


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared -fPIC csrc/pds_kernels.hip -> libpds_hip.so"""
    if not force and not _stale():
        return _LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libpds_hip.so (there is no CPU fallback)")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
           "-o", _LIB + ".tmp", _SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(_LIB + ".tmp", _LIB)
    return _LIB
