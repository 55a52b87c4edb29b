import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from oracle import oracle as O
from tensegrity_rl_amd import TensegrityVecEnv
from tensegrity_rl_amd import _native as N0
N0._build.LIB = N0._build.LIB.replace('libtensegrity.so','libtensegrity_dbg.so'); N0._build.lib_path=lambda: N0._build.LIB; N0._build.needs_build=lambda: False

kind, action, n = "tr_env", "straight", 64
cfg = O.make_cfg(kind, action)
ob = O.OracleBatch(cfg, n, seed=1234); ob.reset()
rng = np.random.default_rng(0)
L = O.lib()
for t in range(4):
    qpos, qvel, ctrl, warm, aux = ob.get_state()
    a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
    if t == 3:
        break
    ob.step(a, auto_reset=False)
# state before step 3; worst env was 12
ctrl_f = ctrl + (a.astype(np.float64) - ctrl) * 0.02
for fs, prec, tol, its in [(k,'f32',0,0) for k in (1,2,3,4,5,6,8,10,12,16,20)]+[(k,'f64',1e-6,12) for k in (1,4,12,20)]:
    env = TensegrityVecEnv(n, kind + "-v0", desired_action=action, precision="f32", auto_reset=False)
    env.cfg.frame_skip = fs
    env.close()
    import tensegrity_rl_amd.vec_env as V
    env = TensegrityVecEnv.__new__(TensegrityVecEnv)
    # build with custom frame_skip via make_config then patch
    import torch
    from tensegrity_rl_amd import _native as N
    c = V.make_config(kind + "-v0", desired_action=action, precision=prec, auto_reset=False, solver_tolerance=tol, solver_iterations=its)
    c.frame_skip = fs
    h = C.c_void_p(); lib = N.load()
    assert lib.tgr_create(C.byref(c), n, 0, C.byref(h)) == 0
    A = np.zeros((n, 16)); A[:, 0] = aux["xy_lag_x"]; A[:, 1] = aux["xy_lag_y"]; A[:, 3] = aux["reset_psi"]
    arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (qpos, qvel, ctrl, warm, A)]
    assert lib.tgr_set_state(h, *[x.ctypes.data for x in arrs]) == 0
    dev = torch.device("cuda:0")
    dt_ = torch.float64 if prec=="f64" else torch.float32
    a_eff = (ctrl + (ctrl_f - ctrl) / (0.001 * fs)).astype(np.float32)
    at = torch.as_tensor(a_eff, device=dev); obs = torch.zeros((n, 45), device=dev, dtype=dt_); rew = torch.zeros(n, device=dev, dtype=dt_)
    te = torch.zeros(n, dtype=torch.uint8, device=dev); tr = torch.zeros(n, dtype=torch.uint8, device=dev)
    assert (lib.tgr_step_f64 if prec=='f64' else lib.tgr_step)(h, at.data_ptr(), obs.data_ptr(), rew.data_ptr(), te.data_ptr(), tr.data_ptr(), None, None, None) == 0
    gq = np.zeros((n, 21)); gv = np.zeros((n, 18)); gw = np.zeros((n, 18))
    assert lib.tgr_get_state(h, gq.ctypes.data, gv.ctypes.data, None, gw.ctypes.data, None) == 0
    lib.tgr_destroy(h)
    # oracle: fs substeps from same state for env 12
    for e in (12,):
        m = O.default_model(); d = O.new_data(m)
        O.arr(d.qpos)[:] = qpos[e]; O.arr(d.qvel)[:] = qvel[e]; O.arr(d.ctrl)[:] = ctrl_f[e]; O.arr(d.qacc_warmstart)[:] = warm[e]
        L.tgo_step(C.byref(m), C.byref(d), fs)
        ev = gv[e] - O.arr(d.qvel)
        cons = [(d.con[k].geom1, d.con[k].geom2, "%.3e" % d.con[k].dist, d.con[k].zone, "%.1f" % d.con[k].force[0]) for k in range(d.ncon)]
        np.set_printoptions(precision=5, suppress=True, linewidth=200)
        
        print("fs", fs, "env", e, "max|dv| %.2e" % np.abs(ev).max(), "argmax", int(np.abs(ev).argmax()), "ncon", d.ncon, cons, "iters", d.solver_iter, flush=True)
