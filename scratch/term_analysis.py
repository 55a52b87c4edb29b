import sys, numpy as np
sys.path.insert(0, "/root/repo")
from oracle import oracle as O
cfg = O.make_cfg("tensegrity_env", "straight")
rng = np.random.default_rng(0)
names = {0: "floor"}
for b in range(3):
    for k, n in enumerate(["r", "s+", "s-", "b+", "b-"]):
        names[1 + 5 * b + k] = f"{n}{b}"
from collections import Counter
cnt = Counter(); lens = []
for ep in range(24):
    e = O.OracleEnv(cfg)
    e.reset_seeded(1, ep, 0)
    for t in range(2000):
        a = rng.uniform(-0.45, 0.15, 6).astype(np.float32)
        obs, r, term, trunc, info = e.step(a)
        if term:
            dd = e.data
            cf = np.array(dd.cfrc_ext)
            big = np.abs(cf) > 1500
            cons = [(names[dd.con[i].geom1], names[dd.con[i].geom2], round(dd.con[i].dist, 5), round(dd.con[i].force[0], 1)) for i in range(dd.ncon)]
            which = "torque" if big[:, :3].any() and not big[:, 3:].any() else ("force" if big[:, 3:].any() else "other")
            # biggest contact
            bc = max(cons, key=lambda c: abs(c[3])) if cons else None
            kind = "none" if bc is None else ("floor" if bc[0] == "floor" else bc[0][:-1] + "-" + bc[1][:-1])
            cnt[(which, kind)] += 1
            print(ep, t, which, "max|cf|=%.0f" % np.abs(cf).max(), cons)
            break
    lens.append(t + 1)
print(cnt); print("mean len", np.mean(lens), lens)
