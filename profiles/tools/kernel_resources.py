"""Register / scratch / LDS table of every kernel in the built library, read from the code objects'
AMDGPU metadata (no recompilation):  python profiles/tools/kernel_resources.py [substring] [lib.so]
Variant<TASK, MOTOR, DR, GE, TN, ON, CTRL, LAT, HOLD>; step_kernel<V, TR> (TR = tile rows).
`scratch` that the VGPR spills do not explain means a local object was forced into private memory (a
dynamic index or a pointer select): on a hot path that is a bug (r02: the latency variants ran at half
speed for it).  tests/test_host_cpu.py checks the shipped library for it."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _bundle_rows(data, base):
    rows = []
    (nent,) = struct.unpack_from("<Q", data, base + len(MAGIC))
    p = base + len(MAGIC) + 8
    for _ in range(nent):
        off, size, tlen = struct.unpack_from("<QQQ", data, p)
        triple = data[p + 24:p + 24 + tlen].decode()
        p += 24 + tlen
        if "gfx950" not in triple or size == 0:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(data[base + off:base + off + size])
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for k in notes.split("  - .agpr_count:")[1:]:
            g = lambda key: re.search(r"\." + key + r":\s+(\S+)", k).group(1)
            rows.append((g("name"), int(g("vgpr_count")), int(g("vgpr_spill_count")), int(g("sgpr_count")),
                         int(g("sgpr_spill_count")), int(g("private_segment_fixed_size")), int(g("group_segment_fixed_size"))))
    return rows


def kernel_table(lib=None):
    """[(demangled name, vgprs, vgpr spills, sgprs, sgpr spills, scratch bytes per lane, LDS bytes per block)]"""
    lib = lib or os.path.join(ROOT, "phoenix-drone-simulation_amd", "libpds_hip.so")
    data = open(lib, "rb").read()
    rows = []
    for m in re.finditer(MAGIC, data):
        rows += _bundle_rows(data, m.start())
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    return sorted((n,) + r[1:] for n, r in zip(names, rows))


def forced_scratch(row):
    """private memory that the register spills do not explain: a local object that lives in scratch"""
    return row[5] > 4 * row[2] + 96


if __name__ == "__main__":
    want = sys.argv[1] if len(sys.argv) > 1 else "step_kernel"
    for r in kernel_table(sys.argv[2] if len(sys.argv) > 2 else None):
        if want not in r[0]:
            continue
        short = re.sub(r"pds::|\(pds::StepArgs\)|void |\(bool\)|\(int\)", "", r[0])
        flag = "  <-- scratch without spills" if forced_scratch(r) else ""
        print(f"{short:78s} VGPR {r[1]:3d} spill {r[2]:3d}  SGPR {r[3]:3d} spill {r[4]:3d}  scratch {r[5]:4d} B/lane  LDS {r[6]:6d}{flag}")
