import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from oracle import oracle as O
from tensegrity_rl_amd import TensegrityVecEnv
from gpu_debug import qfix

kind, action, n = "tr_env", "straight", 64
cfg = O.make_cfg(kind, action)
ob = O.OracleBatch(cfg, n, seed=1234); ob.reset()
rng = np.random.default_rng(0)
states = []
for t in range(6):
    qpos, qvel, ctrl, warm, aux = ob.get_state()
    a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
    ob.step(a, auto_reset=False)
    q2, v2, _, w2, _ = ob.get_state()
    ncon = np.array([ob.env(i).data.ncon for i in range(n)])
    states.append((qpos, qvel, ctrl, warm, aux, a, q2, v2, ncon))
for iters, tol in ((1, 0), (2, 0), (3, 0), (4, 0), (6, 0), (12, 0), (24, 0), (12, 1e-4), (12, 1e-5), (50, 1e-7)):
    env = TensegrityVecEnv(n, kind + "-v0", desired_action=action, precision="f32", auto_reset=False, solver_iterations=iters, solver_tolerance=tol)
    out = []
    for (qpos, qvel, ctrl, warm, aux, a, q2, v2, ncon) in states:
        A = np.zeros((n, 16)); A[:, 0] = aux["xy_lag_x"]; A[:, 1] = aux["xy_lag_y"]; A[:, 3] = aux["reset_psi"]
        env.set_state(qpos, qvel, ctrl, warm, A)
        env.step_tensors(a)
        st = env.get_state()
        ev = np.abs(st["qvel"] - v2).max(axis=1)
        out.append((ev.max(), int(ev.argmax()), int(ncon[ev.argmax()]), np.median(ev)))
    print("iters", iters, "tol", tol, " ".join("%.2e@%d(nc%d,med%.1e)" % o for o in out), flush=True)
    env.close()
