"""ad-hoc GPU debugging driver (not a test)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import oracle as O
from tensegrity_rl_amd import TensegrityVecEnv

def qfix(a, b):
    a = a.reshape(-1, 3, 7).copy(); b = b.reshape(-1, 3, 7)
    s = np.sign(np.sum(a[:, :, 3:] * b[:, :, 3:], axis=2, keepdims=True)); s[s == 0] = 1
    a[:, :, 3:] *= s
    return a.reshape(-1, 21)

def run(kind, action, prec, n=64, steps=6):
    cfg = O.make_cfg(kind, action)
    ob = O.OracleBatch(cfg, n, seed=1234)
    ob.reset()
    env = TensegrityVecEnv(n, kind + "-v0", desired_action=action, precision=prec, auto_reset=False)
    rng = np.random.default_rng(0)
    worst = {}
    for t in range(steps):
        qpos, qvel, ctrl, warm, aux = ob.get_state()
        A = np.zeros((n, 17)); A[:, 0] = aux["xy_lag_x"]; A[:, 1] = aux["xy_lag_y"]; A[:, 3] = aux["reset_psi"]
        A[:, 5] = aux["step_num"]; A[:, 6] = aux["elapsed"]
        env.set_state(qpos, qvel, ctrl, warm, A)
        a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
        o_obs, o_rew, o_term, o_trunc, _, o_info = ob.step(a, auto_reset=False)
        g_obs, g_rew, g_term, g_trunc, _, g_info = [x.cpu().numpy() for x in env.step_tensors(a)]
        st = env.get_state()
        q2, v2, _, w2, _ = ob.get_state()
        def rel(x, y): return np.abs(x - y).max() / (np.abs(y).max() + 1e-12)
        r = dict(qpos=rel(qfix(st["qpos"], q2), q2), qvel=rel(st["qvel"], v2), obs=np.abs(g_obs - o_obs).max(),
                 rew=np.abs(g_rew - o_rew).max(), term=int((g_term.astype(bool) != o_term).sum()), warm=rel(st["qacc_warmstart"], w2))
        print(kind, action, prec, "step", t, {k: (float("%.3g" % v) if not isinstance(v, int) else v) for k, v in r.items()}, "terms", int(o_term.sum()), flush=True)
        for k, v in r.items(): worst[k] = max(worst.get(k, 0), v)
    print("WORST", kind, action, prec, worst, env.stats(), flush=True)
    env.close()

if __name__ == "__main__":
    t0 = time.time()
    for prec in ("f64", "f32"):
        for kind, action in (("tensegrity_env", "straight"), ("tr_env", "straight")):
            run(kind, action, prec)
    print("elapsed", time.time() - t0)
