import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensegrity_rl_amd import TensegrityVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
def rollout(steps=12):
    env = TensegrityVecEnv(n, "tensegrity_env-v0", seed=42, auto_reset=False)
    env.reset_tensors()
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    outs = []
    for i in range(steps):
        a = torch.rand((n, 6), device="cuda", generator=g) * 0.6 - 0.45
        o, r, te, tr, _, _ = env.step_tensors(a)
        outs.append((o.clone(), r.clone()))
    env.close()
    return outs
runs = [rollout() for _ in range(5)]
for i in range(5):
    for j in range(i + 1, 5):
        first = None
        for t, ((o1, r1), (o2, r2)) in enumerate(zip(runs[i], runs[j])):
            if not torch.equal(o1, o2):
                bad = (o1 != o2).any(dim=1).nonzero().flatten()
                e = int(bad[0]); d = (o1[e] - o2[e]).abs()
                first = (t, bad[:6].tolist(), int(bad.numel()), float(d.max()), int(d.argmax())); break
        print("runs", i, j, "first mismatch (step, envs, count, maxdiff, argmax):", first, flush=True)
