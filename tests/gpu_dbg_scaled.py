import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from tensegrity_rl_amd import TensegrityVecEnv
from gpu_common import oracle_to_gpu_state, qfix, rel_env
n=64
for its in [0]*12:
    ob = O.OracleBatch(O.make_cfg("tensegrity_env", "straight"), n, seed=1234); ob.reset()
    kw = dict(solver_warm_start=its)
    env = TensegrityVecEnv(n, "tensegrity_env-v0", precision="f64", auto_reset=False, **kw)
    rng = np.random.default_rng(0)
    for t in range(6):
        oracle_to_gpu_state(ob, env)
        a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
        if t % 3 == 1: a[: n // 4] *= 1.4
        ob.step(a, auto_reset=False); env.step_tensors(a)
        st = env.get_state(); q2, v2 = ob.get_state()[:2]
        e = rel_env(qfix(st["qpos"], q2) - q2, q2, 1e-3)
        flags = [ob.env(i).data.flags for i in range(n)]
        if e.max() > 1e-7: print("BAD its", its, "t", t, "max %.2e" % e.max(), "bad envs", np.nonzero(e > 1e-7)[0].tolist(), flush=True)
    print('run done', flush=True); env.close(); ob.close()
