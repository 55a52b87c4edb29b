import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import oracle as O
from tensegrity_rl_amd.evaluate import load_sb3_actor
actor = load_sb3_actor("/root/reference/models_tracking/SAC_16525000.zip")
n = 24
cfg = O.make_cfg("tr_env", "tracking", is_test=1)
ob = O.OracleBatch(cfg, n, seed=1234)
obs = ob.reset()
import ctypes as C
L = O.lib()
poses = []
for i in range(n):
    pi = C.c_int(); th = C.c_double(); ten = (C.c_double*6)(); uw = (C.c_double*2)()
    L.tgo_rng_draws(C.c_uint64(1234), C.c_uint64(i), C.c_uint64(0), 1, C.c_double(cfg.tendon_reset_mean), C.c_double(cfg.tendon_reset_stdev), C.c_double(cfg.tendon_min_length), C.c_double(cfg.tendon_max_length), C.c_double(cfg.min_reset_heading), C.c_double(cfg.max_reset_heading), C.byref(pi), C.byref(th), ten, uw)
    poses.append((pi.value, round(th.value, 2), np.round(list(ten), 2)))
d0 = None
for t in range(400):
    a = actor.act(torch.from_numpy(obs), deterministic=True).numpy().astype(np.float32)
    obs, rew, term, trunc, tobs, info = ob.step(a, auto_reset=False)
    d = np.linalg.norm(info[:, 10:12] - info[:, 3:5], axis=1)
    if d0 is None: d0 = d.copy()
for i in range(n):
    print(i, "pose", poses[i][0], "theta", poses[i][1], "tend", poses[i][2], "dist %.2f -> %.2f" % (d0[i], d[i]))
