#!/bin/bash
# SQ counters of the fused MLP kernel (profiles/tools/pmc_mlp.sh): two passes, one size (B = 2^21)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/mlp_one.py <<PY
import math, sys, torch
sys.path.insert(0, "$REPO")
from phoenix_drone_simulation_amd.fused import FusedMLP
from phoenix_drone_simulation_amd.ppo import _mlp
B, D, H, A = 2097152, 34, 50, 4
net = _mlp([D, H, H, A], "relu").cuda(); fm = FusedMLP(net, "relu")
x = torch.randn(B, D, device="cuda"); act = torch.randn(B, A, device="cuda"); adv = torch.randn(B, device="cuda")
lp = torch.randn(B, device="cuda") - 4; ls = torch.full((A,), math.log(0.3), device="cuda")
for _ in range(5): fm.ppo_grad(x, act, adv, lp, ls, 0.2)
torch.cuda.synchronize()
PY
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pmc_mlp; rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_mlp -- python3 /tmp/mlp_one.py > /tmp/pmc_mlp.log 2>&1 || tail -3 /tmp/pmc_mlp.log
  python3 - <<PY
import csv, glob, collections
fs = glob.glob("/tmp/pmc_mlp/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for f in fs:
    for r in csv.DictReader(open(f)):
        if "mlp_kernel" in r["Kernel_Name"] or "ppo_split_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print(f"{k:28s} {sum(v)/len(v):16.1f}   (n={len(v)})")
PY
done
