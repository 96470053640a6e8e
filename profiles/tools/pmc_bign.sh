#!/bin/bash
# L2 <-> fabric request counters of the step kernel at 2^20 vs 2^22 envs (why is the large batch slower per env?)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for n in 1048576 4194304; do
  for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pmc_bn; rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_bn -- python3 $REPO/bench.py --envs-per-gpu $n --steps 20 --warmup 5 --no-cpu-baseline > /tmp/pmc_bn.log 2>&1 || tail -2 /tmp/pmc_bn.log
    python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_bn/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print(f"n=$n {k:40s} {sum(v)/len(v):16.1f}  per env {sum(v)/len(v)/$n:10.4f}")
PY
  done
done
