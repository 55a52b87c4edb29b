"""diagnostic: what solver tolerance / arithmetic does the 1e-4 per-env-step bound need?  One-step error of HIP variants
(precision x solver_tolerance) against the oracle on identical (state, action); also which qvel component carries the error.
usage: python tests/gpu_tol_study.py [tensegrity_env|tr_env] [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from oracle import oracle as O
from gpu_common import oracle_to_gpu_state, qfix, rel_env
from tensegrity_rl_amd import TensegrityVecEnv
kind = sys.argv[1] if len(sys.argv) > 1 else "tensegrity_env"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
variants = [dict(precision="f32"), dict(precision="f32", solver_tolerance=1e-6), dict(precision="f32", solver_tolerance=3e-7, solver_iterations=12),
            dict(precision="f64", solver_tolerance=3e-6), dict(precision="f64", solver_tolerance=1e-6), dict(precision="f64", solver_tolerance=1e-7),
            dict(precision="f64", solver_tolerance=1e-8), dict(precision="f64", solver_tolerance=1e-9)]
ob = O.OracleBatch(O.make_cfg(kind, "straight"), n, seed=1234); ob.reset()
envs = [TensegrityVecEnv(n, kind + "-v0", auto_reset=False, **kw) for kw in variants]
rng = np.random.default_rng(0)
rows = [[] for _ in envs]
comp = [np.zeros(18) for _ in envs]
for t in range(6):
    a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
    for env in envs: oracle_to_gpu_state(ob, env)
    ob.step(a, auto_reset=False)
    q2, v2, _, _, _ = ob.get_state()
    for i, env in enumerate(envs):
        env.step_tensors(a); st = env.get_state()
        ev = rel_env(st["qvel"] - v2, v2, 1e-2)
        rows[i].append((rel_env(qfix(st["qpos"], q2) - q2, q2, 1e-3), ev))
        comp[i] += np.median(np.abs(st["qvel"] - v2), axis=0)
for i, kw in enumerate(variants):
    eq = np.concatenate([r[0] for r in rows[i]]); ev = np.concatenate([r[1] for r in rows[i]])
    print(kind, kw, envs[i].stats())
    print(" " * 6, "qpos rel err 50/90/99/max: %.1e %.1e %.1e %.1e  frac<1e-4 %.4f" % (*np.quantile(eq, [.5, .9, .99]), eq.max(), (eq < 1e-4).mean()))
    print(" " * 6, "qvel rel err 50/90/99/max: %.1e %.1e %.1e %.1e  frac<1e-4 %.4f" % (*np.quantile(ev, [.5, .9, .99]), ev.max(), (ev < 1e-4).mean()))
    print(" " * 6, "median |dqvel| by component (bar0 v, w_body; ...):", np.array2string(comp[i] / 6, precision=1, max_line_width=200), flush=True)
