"""development aid: wall time per chunk of steps from reset() on (the donor pool's start-up transient), one line per chunk.
usage: TGR_LIB=... gpu_step_timeline.py [n] [env id] [steps] [chunk]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensegrity_rl_amd import TensegrityVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = sys.argv[2] if len(sys.argv) > 2 else "tensegrity_env-v0"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 600
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 20
env = TensegrityVecEnv(n, kind, auto_reset=True, n_donors=5 * n // 4)
env.reset_tensors()
g = torch.Generator(device="cuda"); g.manual_seed(0)
acts = [torch.rand((n, 6), device="cuda", generator=g) * 0.6 - 0.45 for _ in range(16)]
torch.cuda.synchronize()
out = []
s_prev = env.stats()
for c in range(steps // chunk):
    t0 = time.perf_counter()
    for i in range(chunk):
        env.step_tensors(acts[(c * chunk + i) % 16])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s = env.stats()
    out.append(f"steps {c*chunk:4d}..{(c+1)*chunk-1:4d}: {dt/chunk*1e3:.3f} ms/step  donor resets {s['donor_resets']-s_prev['donor_resets']:5d}  settle-kernel resets {s['settle_kernel_resets']-s_prev['settle_kernel_resets']:5d}")
    s_prev = s
print(os.environ.get("TGR_LIB", "in-tree"), "\n" + "\n".join(out), flush=True)
