#!/bin/bash
# SQ counter pass for the step kernel (dynamic VALU instructions per wave, wait share, LDS conflicts).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_sq_${1:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-traffic ${BENCH_ARGS:-} > /dev/null 2>&1
cd $REPO && python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "${KERNEL_PATTERN:-step_kernel}" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v)/len(v) for k, v in acc.items()}
for k, v in sorted(m.items()): print(f"{k:24s} {v:14.1f}")
w = m["SQ_WAVES"]
print(f"per 64-env tile (wave): VALU {m['SQ_INSTS_VALU']/w:.0f}   SALU {m['SQ_INSTS_SALU']/w:.0f}   LDS instr {m['SQ_INSTS_LDS']/w:.1f}   "
      f"LDS bank-conflict cycles {m['SQ_LDS_BANK_CONFLICT']/w:.0f}   wave cycles (quad) {m['SQ_WAVE_CYCLES']/w:.0f}   wait-any {m['SQ_WAIT_ANY']/w:.0f}")
PY
