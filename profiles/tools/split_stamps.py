"""Phase boundaries (s_memtime, shader clocks) of one tile of the role-split PPO gradient kernel.
Needs a profiling build of pds_mlp.hip (-DPDS_SPLIT_STAMPS=1) given as PDS_LIB (profiles/tools/split_stamps.py)."""
import ctypes, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phoenix_drone_simulation_amd.fused import FusedMLP
from phoenix_drone_simulation_amd.ppo import _mlp
from phoenix_drone_simulation_amd import build
B, D, H, A = 1048576, 42, 50, 4
net = _mlp([D, H, H, A], "relu").cuda(); fm = FusedMLP(net, "relu")
x = torch.randn(B, D, device="cuda"); act = torch.randn(B, A, device="cuda"); adv = torch.randn(B, device="cuda")
lp = torch.randn(B, device="cuda") - 4; ls = torch.full((A,), math.log(0.3), device="cuda")
for _ in range(3): fm.ppo_grad(x, act, adv, lp, ls, 0.2)
torch.cuda.synchronize()
lib = ctypes.CDLL(build.library_path())
out = (ctypes.c_ulonglong * 32)()
assert lib.pds_debug_split_stamps(out) == 0
names = {0: ["top", "set free", "L1 MFMAs+x prefetch issued", "H1 epilogue done", "L2 MFMAs issued", "H2 epilogue done", "L3 issued", "loss+dY+sync", "dZ2,dW3 issued", "dZ2 stored", "signalled"],
         1: ["top", "tile arrived", "dZ1 issued", "dW2 issued", "dz1 stored+sync", "dW1 issued", "signalled"]}
for role in (0, 1):
    st = [out[role * 16 + i] for i in range(len(names[role]))]
    print("role", "FG"[role], "tile 10 of pair 0, block 0: total", st[-1] - st[0], "clocks")
    for i in range(1, len(st)): print(f"   {names[role][i]:32s} +{st[i] - st[i - 1]:6d}")
