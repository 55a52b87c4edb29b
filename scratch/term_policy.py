import sys, numpy as np
sys.path.insert(0, "/root/repo")
from oracle import oracle as O
names = {0: "floor"}
for b in range(3):
    for k, n in enumerate(["r", "s+", "s-", "b+", "b-"]):
        names[1 + 5 * b + k] = f"{n}{b}"
for name, direction in (("forward", 1), ("backward", -1)):
    g = np.load(f"/root/repo/tests/golden/policy_{name}.npz")
    W = [g[k].astype(np.float64) for k in ("latent_pi_0_weight", "latent_pi_2_weight", "mu_weight")]
    B = [g[k].astype(np.float64) for k in ("latent_pi_0_bias", "latent_pi_2_bias", "mu_bias")]
    lo, hi = float(g["low"]), float(g["high"])
    cfg = O.make_cfg("tensegrity_env", "straight", desired_direction=direction)
    for ep in range(8):
        e = O.OracleEnv(cfg)
        obs = e.reset_seeded(1234, ep, 0)
        mx = 0
        hist = []
        for t in range(400):
            h = obs
            for i in range(2): h = np.maximum(W[i] @ h + B[i], 0)
            a = (lo + 0.5 * (np.tanh(W[2] @ h + B[2]) + 1) * (hi - lo)).astype(np.float32)
            obs, r, term, trunc, info = e.step(a)
            dd = e.data
            cf = np.abs(np.array(dd.cfrc_ext)).max()
            hist.append(cf)
            if term:
                cons = [(names[dd.con[i].geom1], names[dd.con[i].geom2], round(dd.con[i].dist, 5), round(dd.con[i].force[0], 1)) for i in range(dd.ncon)]
                print(name, ep, t, "max|cf|=%.0f" % cf, cons, "prev cf max", np.round(sorted(hist)[-5:]))
                break
        else:
            print(name, ep, "survived", np.round(sorted(hist)[-5:]))
