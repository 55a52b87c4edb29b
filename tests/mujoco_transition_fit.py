"""Development aid (CPU, oracle): what the ONE MuJoCo transition each SB3 checkpoint of the reference holds says about the dynamics.

`_last_original_obs` and `_last_obs` in a checkpoint's `data` member are the observations of step t and t + 1 of one rollout (SB3's
`_store_transition`; integrating the recorded velocities over 20 ms carries the first pose into the second to 1e-5): real MuJoCo, one
control period apart.  Not recorded: the action, the filter activations, the height above the floor, (tr_env) the bars' spin rates about their
own axes -- and the constants of the model version the checkpoint was trained on.  This script fits those unknowns (13 for tensegrity_env,
16 for tr_env) with the oracle stepping TODAY's uneven-XML robot on a flat floor, starting from the static force balance, and reports what is
left of the 36 / 33 recorded numbers of step t + 1.  A residual far below the recorded change would pin the dynamics against MuJoCo; what
it finds (profiles/round4_mujoco_transition_fit.log) is agreement to a few per cent of the velocity change on the slow records -- the right
ballpark, not a pin: the model version of the time is not the one of today (its action box, and for six checkpoints its site offset, differ).

    python tests/mujoco_transition_fit.py [substring of the checkpoint path ...] [constant=value ...] > profiles/roundN_mujoco_transition_fit.log"""
import ctypes as C
import json
import os
import sys

import numpy as np
from scipy.optimize import least_squares

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
import smooth_numpy as SN  # noqa: E402
import test_oracle_anchors as A  # noqa: E402


def advance(q, v, h):
    X = q.reshape(3, 7).copy()
    for b in range(3):
        X[b, :3] += h * v[6 * b:6 * b + 3]
        w = v[6 * b + 3:6 * b + 6]; n = np.linalg.norm(w)
        if n > 0:
            qq = SN.qmul(X[b, 3:7], np.concatenate([[np.cos(h * n / 2)], np.sin(h * n / 2) * w / n])); X[b, 3:7] = qq / np.linalg.norm(qq)
    return X.reshape(21)


def statics_start(m, c, q, cap_points):
    """actuator forces that balance passive tendons + gravity together with forces at the three lowest caps (least squares)"""
    Lq, J = SN.tendons(c, q)
    g = np.zeros(18)
    for b in range(3):
        g[6 * b + 2] = -c.mass * c.g
    known = J.T @ (-c.k * (Lq - c.L0)) + g
    cols = [J[t] for t in range(6)]
    X = q.reshape(3, 7)
    for b, r in cap_points:
        R = SN.rot(X[b, 3:7])
        for e in np.eye(3):
            col = np.zeros(18); col[6 * b:6 * b + 3] = e; col[6 * b + 3:6 * b + 6] = R.T @ np.cross(r, e); cols.append(col)
    x = np.linalg.lstsq(np.array(cols).T, -known, rcond=None)[0]
    return x[:6] / m.gain, x[[8, 11, 14]]


OVERRIDES = {}   # model constants to try instead of today's: name=value on the command line (solref0, solref1, act_tau, friction0, ten_damping, ten_stiffness, ...)


def model(flat05=False):
    m = A._with_site_offset(O.default_model(), 0.05) if flat05 else O.uneven_model(None)
    if flat05:
        m.cap_radius = 0.05     # (sites sit on the cap's surface in every version of the XML)
    for k, v in OVERRIDES.items():
        if k in ("ten_damping", "ten_stiffness"):
            for t in range(9):
                getattr(m, k)[t] = v
        elif k in ("ten_damping_act", "ten_stiffness_act", "ten_damping_pas", "ten_stiffness_pas"):   # the six actuated / the three passive tendons
            for t in (range(6) if k.endswith("_act") else range(6, 9)):
                getattr(m, k[:-4])[t] = v
        elif k[-1].isdigit() and hasattr(m, k[:-1]):
            getattr(m, k[:-1])[int(k[-1])] = v
        else:
            setattr(m, k, v)
    return m


def fit(rec, flat05=False):
    L = O.lib()
    m = model(flat05); c = SN.Consts(m); h = m.timestep; H = m.bar_half
    o0 = np.array(rec["_last_original_obs"]); o1 = np.array(rec["_last_obs"]); dim = rec["obs_dim"]
    d = O.new_data(m)
    if dim == 39:
        def pose(o):
            quats = [A._body_quat_from_obs(o[4 * b:4 * b + 4]) for b in range(3)]
            f = A._fit_relative_positions(O, m, quats, o[30:39], starts=6)
            q = np.zeros(21)
            for b in range(3):
                q[7 * b + 3:7 * b + 7] = quats[b]
            q[7:10] = f.x[:3]; q[14:17] = f.x[3:]
            return q
        ql0, v0 = pose(o0), o0[12:30]; ql1, v1 = pose(o1), o1[12:30]
        nspin = 0
        build_v = lambda s: v0
    else:
        if dim < 45:
            return None
        caps0, vel0, t0 = o0[:18].reshape(6, 3), o0[18:36].reshape(6, 3), o0[36:45]
        caps1, vel1, t1 = o1[:18].reshape(6, 3), o1[18:36].reshape(6, 3), o1[36:45]
        a, ql0 = A._fit_spins(O, m, caps0, t0, want_qpos=True)
        if a > 1e-12:
            return None if flat05 else fit(rec, flat05=True)     # (six models_aiming checkpoints: the flat XML's layout, sites at 0.05 m)
        ql0 = ql0.copy(); ql0[2::7] -= 5.0
        axes = [(caps0[2 * b] - caps0[2 * b + 1]) / (2 * H) for b in range(3)]
        vlin = [(vel0[2 * b] + vel0[2 * b + 1]) / 2 for b in range(3)]
        wperp = [np.cross(axes[b], vel0[2 * b] - vel0[2 * b + 1]) / (2 * H) for b in range(3)]   # (tr_env.py:586-604: body-frame w used as a world vector)
        nspin = 3

        def build_v(s):
            v = np.zeros(18)
            for b in range(3):
                v[6 * b:6 * b + 3] = vlin[b]; v[6 * b + 3:6 * b + 6] = wperp[b] + s[b] * axes[b]
            return v
    q0 = advance(ql0, build_v(np.zeros(3)), h)
    capz, pts = [], []
    for b in range(3):
        ez = SN.rot(q0[7 * b + 3:7 * b + 7])[:, 2]
        for s in (1, -1):
            capz.append(q0[7 * b + 2] + s * H * ez[2]); pts.append((b, s * H * ez - np.array([0, 0, m.cap_radius])))
    low = np.argsort(capz)[:3]
    act0, normal = statics_start(m, c, q0, [pts[k] for k in low])
    if flat05:      # affine actuators: F = gain ctrl + b0 + b1 L (+ b2 Ldot): the statics force names ctrl; `act` is not a state of this model
        Lq0 = SN.tendons(c, q0)[0][:6]
        act0 = np.clip((act0 * m.gain - m.bias0 - m.bias1 * Lq0) / m.gain, -0.45, 0.15)
    zlow = min(capz)

    def predict(p):
        v = build_v(p[13:16]) if nspin else v0
        q = advance(ql0, v, h); q[2::7] += p[0]
        O.arr(d.qpos)[:] = q; O.arr(d.qvel)[:] = v; O.arr(d.ctrl)[:] = p[1:7]; O.arr(d.act)[:] = p[7:13]; O.arr(d.qacc_warmstart)[:] = 0
        L.tgo_step(C.byref(m), C.byref(d), 20)          # (pose-derived quantities of d lag the velocity by one sub-step, as in mujoco)
        return d.ncon

    def residual(p):
        predict(p)
        vv = O.arr(d.qvel).copy()
        if dim == 39:
            xm = np.array([[d.xmat[b][k] for k in range(9)] for b in range(3)]).reshape(3, 3, 3)
            xp = np.array([[d.xpos[b][k] for k in range(3)] for b in range(3)])
            r = []
            for b in range(3):
                want = SN.rot(ql1[7 * b + 3:7 * b + 7])
                r += list((xm[b] - want).ravel() * 0.5)
            r += list((xp[1:] - xp[0]).ravel() - (ql1.reshape(3, 7)[1:, :3] - ql1.reshape(3, 7)[0, :3]).ravel())
            return np.array(r), vv - v1
        cp = O.arr(d.cap_xpos).reshape(6, 3).copy(); xp = np.array([[d.xpos[b][k] for k in range(3)] for b in range(3)])
        cv = np.array([vv[6 * (k // 2):6 * (k // 2) + 3] + np.cross(vv[6 * (k // 2) + 3:6 * (k // 2) + 6], cp[k] - xp[k // 2]) for k in range(6)])
        return np.concatenate([(cp - cp.mean(0) - caps1).ravel(), O.arr(d.ten_length) - t1]), (cv - vel1).ravel()

    res = lambda p: np.concatenate([residual(p)[0], 0.1 * residual(p)[1]])
    best = None
    for pen in (1e-4, 3e-4, 6e-4, 1e-3, 1.5e-3):
        z0 = m.cap_radius - zlow - pen
        x0 = np.concatenate([[z0], act0, act0, np.zeros(nspin)])
        lo = np.concatenate([[z0 - 3e-4], act0 - 0.4, act0 - 0.05, np.full(nspin, -20.0)])
        hi = np.concatenate([[z0 + 3e-4], act0 + 0.4, act0 + 0.05, np.full(nspin, 20.0)])
        xs = np.concatenate([[1e-4], np.full(6, 1e-2), np.full(6, 1e-4), np.full(nspin, 1.0)])
        r = least_squares(res, x0, bounds=(lo, hi), xtol=1e-15, ftol=1e-15, gtol=1e-15, diff_step=1e-7, x_scale=xs)
        if best is None or r.cost < best.cost:
            best = r
    rp, rv = residual(best.x)
    dv = np.abs((v1 - v0) if dim == 39 else (vel1 - vel0)).max()
    dp = np.abs(ql1 - ql0).max() if dim == 39 else np.abs(caps1 - caps0).max()
    speed = np.abs(v0).max() if dim == 39 else np.abs(vel0).max()
    return dict(file=rec["file"], dim=dim, speed=speed, pose_change=dp, vel_change=dv, pose_err=np.abs(rp).max(), vel_err=np.abs(rv).max(),
                ncon=predict(best.x), penetration=m.cap_radius - zlow - best.x[0], statics_normal=normal)


def main():
    O.build()
    recs = json.load(open(os.path.join(ROOT, "tests", "golden", "mujoco_last_obs.json")))["records"]
    want = [a for a in sys.argv[1:] if "=" not in a]
    OVERRIDES.update({a.split("=")[0]: float(a.split("=")[1]) for a in sys.argv[1:] if "=" in a})
    if OVERRIDES:
        print("# model constants overridden:", OVERRIDES)
    print("# fit of the unknowns of the one MuJoCo transition per checkpoint with the oracle (today's uneven-XML robot, flat floor at z = 0)")
    print("# file | obs dim | speed at t | recorded change over the step: pose, velocity | left after the fit: pose, velocity (fraction of the change) | contacts | lowest cap's penetration")
    for r in recs:
        if want and not any(w in r["file"] for w in want):
            continue
        out = fit(r)
        if out is None:
            continue
        print("%-62s %2d  speed %.3f  change %.1e / %.3f  left %.1e / %.4f (%.0f %% of the velocity change)  ncon %d  pen %.2f mm  statics normal forces %s"
              % (out["file"], out["dim"], out["speed"], out["pose_change"], out["vel_change"], out["pose_err"], out["vel_err"],
                 100 * out["vel_err"] / max(out["vel_change"], 1e-12), out["ncon"], 1e3 * out["penetration"], np.round(out["statics_normal"], 1)), flush=True)


if __name__ == "__main__":
    main()
