"""diagnostic: distribution of the fp32 HIP path's one-step error against the oracle on identical (state, action)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from oracle import oracle as O
from gpu_common import oracle_to_gpu_state, qfix, rel_env
from tensegrity_rl_amd import TensegrityVecEnv
n = 512
kind = sys.argv[1] if len(sys.argv) > 1 else "tensegrity_env"
variants = [dict(), dict(solver_tolerance=1e-5), dict(solver_tolerance=1e-6), dict(solver_iterations=6)]
ob = O.OracleBatch(O.make_cfg(kind, "straight"), n, seed=1234); ob.reset()
envs = [TensegrityVecEnv(n, kind + "-v0", precision="f32", auto_reset=False, **kw) for kw in variants]
rng = np.random.default_rng(0)
rows = [[] for _ in envs]
for t in range(6):
    a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
    for env in envs: oracle_to_gpu_state(ob, env)
    ob.step(a, auto_reset=False)
    q2, v2, _, _, _ = ob.get_state()
    for i, env in enumerate(envs):
        env.step_tensors(a); st = env.get_state()
        rows[i].append((rel_env(qfix(st["qpos"], q2) - q2, q2, 1e-3), rel_env(st["qvel"] - v2, v2, 1e-2)))
for i, kw in enumerate(variants):
    eq = np.concatenate([r[0] for r in rows[i]]); ev = np.concatenate([r[1] for r in rows[i]])
    print(kw, "qpos rel err quantiles 50/90/99/max: %.1e %.1e %.1e %.1e  frac<1e-4 %.3f" % (*np.quantile(eq, [.5, .9, .99]), eq.max(), (eq < 1e-4).mean()))
    print(" " * 10, "qvel rel err quantiles 50/90/99/max: %.1e %.1e %.1e %.1e  frac<1e-4 %.3f" % (*np.quantile(ev, [.5, .9, .99]), ev.max(), (ev < 1e-4).mean()), flush=True)
