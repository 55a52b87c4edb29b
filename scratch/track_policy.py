import sys, io, zipfile, json, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import oracle as O
from tensegrity_rl_amd.evaluate import load_sb3_actor
path = sys.argv[1]; action = sys.argv[2]; direction = int(sys.argv[3]) if len(sys.argv) > 3 else 1
actor = load_sb3_actor(path)
print(actor.kind, actor.obs_dim, actor.low, actor.high)
n = 8
cfg = O.make_cfg("tr_env", action, desired_direction=direction, is_test=1)
ob = O.OracleBatch(cfg, n, seed=1234)
obs = ob.reset()
print("obs dim", obs.shape)
x0 = np.array([O.arr(ob.env(i).data.xpos)[:, :2].mean(axis=0) for i in range(n)])
alive = np.ones(n, bool); tlen = np.zeros(n, int)
for t in range(600):
    a = actor.act(torch.from_numpy(obs), deterministic=True).numpy().astype(np.float32)
    obs, rew, term, trunc, tobs, info = ob.step(a, auto_reset=False)
    newly = alive & term
    for i in np.nonzero(newly)[0]:
        print("env", i, "terminated at", t, "xy", info[i, 3:5].round(3), "waypt", info[i, 10:12].round(3), "cfrc?", np.abs(np.array(ob.env(i).data.cfrc_ext)).max().round(0), "bar", info[i,25].round(0))
    alive &= ~term
    tlen[alive] = t + 1
    if t % 100 == 99:
        d = np.linalg.norm(info[:, 10:12] - info[:, 3:5], axis=1)
        print(t, "dist to waypt", d.round(2), "alive", alive.astype(int))
