"""development aid: how many "other" contacts a bar pair has on the headline workload (library built with -DTGR_OTHERS_HIST) -- sizes the slots
a pair needs besides the r-cylinders' side-to-side contact.  usage: TGR_LIB=... gpu_others_hist.py [n] [env id] [steps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensegrity_rl_amd import TensegrityVecEnv, _native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = sys.argv[2] if len(sys.argv) > 2 else "tensegrity_env-v0"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
env = TensegrityVecEnv(n, kind, auto_reset=True)
env.reset_tensors()
g = torch.Generator(device="cuda"); g.manual_seed(0)
acts = [torch.rand((n, 6), device="cuda", generator=g) * 0.6 - 0.45 for _ in range(16)]
for i in range(steps):
    env.step_tensors(acts[i % 16])
torch.cuda.synchronize()
lib = _native.load()
out = (ctypes.c_ulonglong * 64)()
lib.tgr_debug_others_hist(out)
h = np.array(out[:], dtype=np.int64)
print(f"{n} envs (+ donor lanes) x {steps} steps x 20 sub-steps, {kind}: (env, pair, sub-step) with k other contacts:")
for k in range(1, 32):
    if h[k]: print(f"  k = {k:2d}: {h[k]}")
print("environments (sub-steps) in which 1 / 2 / 3 pairs want more than their fixed slots:", h[33], h[34], h[35])
print("stats", env.stats())
