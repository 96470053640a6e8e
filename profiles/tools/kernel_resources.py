"""Per-kernel register / LDS / occupancy table of one translation unit:
  hipcc <build flags> -Rpass-analysis=kernel-resource-usage -c csrc/pds_task_hover.hip -o /tmp/x.o 2> res.txt
  python profiles/tools/kernel_resources.py res.txt [substring]
Variant<TASK, MOTOR, DR, GE, TN, ON, CTRL, LAT, HOLD>; step_kernel<V, TR> (TR = tile rows)."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else "step_kernel"
for b in txt.split("Function Name: ")[1:]:
    name = b.split()[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if want not in dem:
        continue
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    short = re.sub(r"pds::|\(pds::StepArgs\)|void ", "", dem)
    short = re.sub(r"\(bool\)|\(int\)", "", short)
    scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{short:74s} VGPR {g('VGPRs'):3d} spill {g('VGPRs Spill'):3d}  SGPR {g('TotalSGPRs'):3d} spill {g('SGPRs Spill'):3d}  "
          f"scratch {scratch:4d}  occ {occ}  LDS {lds}")
