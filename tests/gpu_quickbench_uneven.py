"""throughput of the uneven-ground model (BASELINE config #4: 8192 envs), random actions, auto-reset"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensegrity_rl_amd import TensegrityVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
kind = sys.argv[2] if len(sys.argv) > 2 else "tr_env-v0"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
img = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "terrain_100.npy"))
env = TensegrityVecEnv(n, kind, model="uneven", terrain=img)
env.reset_tensors()
g = torch.Generator(device="cuda"); g.manual_seed(0)
acts = [torch.rand((n, 6), device="cuda", generator=g) * 0.6 - 0.45 for _ in range(16)]
for i in range(20): env.step_tensors(acts[i % 16])
torch.cuda.synchronize(); t0 = time.time(); dones = 0
for i in range(steps):
    o, r, te, tr, _, _ = env.step_tensors(acts[i % 16])
    dones += int((te | tr).sum().item()) if i % 10 == 0 else 0
torch.cuda.synchronize(); dt = time.time() - t0
print(f"uneven n={n} {kind}: {dt/steps*1e3:.3f} ms/step, {n*steps/dt/1e6:.3f} M env-steps/s; dones sampled {dones}; stats {env.stats()}", flush=True)
