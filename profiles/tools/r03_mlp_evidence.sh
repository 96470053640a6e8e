#!/bin/bash
# Round-3 evidence for the trainer's PPO-gradient kernel (profiles/tools/r03_mlp_evidence.sh), run on the GPU box.
# A/B libraries are built beforehand (CPU container) into profiles/tools/scratch/ from the same sources:
#   libpds_r2mlp.so   -DPDS_MLP_SPLIT=0 -DPDS_MLP_EDGE=0  (the round-2 kernel)
#   libpds_nosplit.so -DPDS_MLP_SPLIT=0                   (round 2 + features 48/49 on the vector ALU)
#   libpds_split3.so  -DPDS_MLP_SPLIT=2                   (three waves per SIMD)
#   libpds_dbg1.so / libpds_dbg2.so -DPDS_SPLIT_DEBUG=1/2 (role F alone / role G alone: timing only, results invalid)
#   libpds_dbg3.so    -DPDS_SPLIT_DEBUG=3                  (role F without its MFMAs next to G: timing only)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r03_mlp; mkdir -p $O
S=$REPO/profiles/tools/scratch
cd $REPO
python profiles/tools/mlp_bench.py 2>&1 | grep -v amdgpu > $O/mlp_bench.txt
( for v in r2mlp nosplit; do echo "== $v"; PDS_LIB=$S/libpds_$v.so python profiles/tools/mlp_sizes.py 2>&1 | grep "^D"; done
  echo "== wave roles (shipped)"; python profiles/tools/mlp_sizes.py 2>&1 | grep "^D"
  echo "== three waves per SIMD (-DPDS_MLP_SPLIT=2)"; PDS_LIB=$S/libpds_split3.so python profiles/tools/mlp_sizes.py 2>&1 | grep "^D 42"
  echo "== role F without its MFMAs, next to G (PDS_SPLIT_DEBUG=3)"; PDS_LIB=$S/libpds_dbg3.so python profiles/tools/mlp_sizes.py 2>&1 | grep "^D 42" | grep -E "B +(16|524288|1048576):"
  for d in 1 2; do echo "== role $( [ $d = 1 ] && echo F || echo G ) alone (PDS_SPLIT_DEBUG=$d)"; PDS_LIB=$S/libpds_dbg$d.so python profiles/tools/mlp_sizes.py 2>&1 | grep "^D 42" | grep -E "B +(16|524288|1048576):"; done ) > $O/mlp_ab.txt 2>&1
bash profiles/tools/pmc_mlp.sh > $O/pmc_mlp_split.txt 2>&1
( $S/wave_simd; $S/mfma_issue ) > $O/microbench.txt 2>&1
( python profiles/tools/ppo_breakdown.py 8192 64; python profiles/tools/ppo_breakdown.py 65536 32; python profiles/tools/ppo_breakdown.py 1048576 8 ) 2>&1 | grep '^N' > $O/ppo_breakdown.txt
python profiles/tools/mlp_small.py 2>&1 | grep '^B' > $O/mlp_small.txt
cd /tmp && export TMPDIR=/tmp
cat > /tmp/mlp_prof.py <<PY
import math, sys, torch
sys.path.insert(0, "$REPO")
from phoenix_drone_simulation_amd.fused import FusedMLP
from phoenix_drone_simulation_amd.ppo import _mlp
B, D, H, A = 1048576, 42, 50, 4
net = _mlp([D, H, H, A], "relu").cuda(); fm = FusedMLP(net, "relu")
x = torch.randn(B, D, device="cuda"); act = torch.randn(B, A, device="cuda"); adv = torch.randn(B, device="cuda")
lp = torch.randn(B, device="cuda") - 4; ls = torch.full((A,), math.log(0.3), device="cuda")
for _ in range(55): fm.ppo_grad(x, act, adv, lp, ls, 0.2)
torch.cuda.synchronize()
PY
rm -rf /tmp/mlp_prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mlp_prof -- python3 /tmp/mlp_prof.py > /tmp/mlp_prof.log 2>&1
f=$(find /tmp/mlp_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" > $O/kernel_stats_ppo_grad_1M.csv
