"""Pin the physics restatement against the real MuJoCo engine -- WHEN `import mujoco` works (TEST / VALIDATION TOOLING).

mujoco is not installed in the build container nor on the GPU box of this round (no network, no wheel), so nothing here has
run yet: physics parity is recorded as "unpinned" (DESIGN.md section 2).  The moment a machine has `mujoco` (the reference
pins 2.3.7), `python -m oracle.mujoco_check` or `pytest tests/test_mujoco_optional.py` closes that gap without touching the
reference checkout: the model comes from oracle/mjcf.py (generated from the constants table, checked number by number
against the reference XML by tests/test_model_constants.py).

What is compared, on identical (qpos, qvel, ctrl, qacc_warmstart):
  * derived constants: body mass / inertia, tendon lengths of the initial pose;
  * one control period = mj_step(nstep=20) against tgo_step(20): relative error of qpos and qvel, reported separately for
    periods with and without bar-bar contacts (MuJoCo collides cylinders with MPR, the restatement analytically -- see
    DESIGN.md "deliberate deviations").
"""
import ctypes as C

import numpy as np


def available():
    try:
        import mujoco  # noqa: F401
        return True
    except Exception:
        return False


def compare(n_states=32, periods=5, seed=0):
    import mujoco
    from . import mjcf
    from . import oracle as O
    L = O.lib()
    m = O.default_model()
    model = mujoco.MjModel.from_xml_string(mjcf.emit_mjcf(m))
    data = mujoco.MjData(model)
    out = {"mujoco_version": mujoco.__version__}
    out["mass_err"] = float(np.abs(model.body_mass[1:] - m.mass).max())
    out["inertia_err"] = float(np.abs(model.body_inertia[1:] - np.array(m.inertia)).max())
    mujoco.mj_forward(model, data)
    d = O.new_data(m)
    L.tgo_forward(C.byref(m), C.byref(d))
    out["ten_length_err_qpos0"] = float(np.abs(np.array(data.ten_length) - O.arr(d.ten_length)).max())
    rng = np.random.default_rng(seed)
    worst = {"free": [0.0, 0.0], "barbar": [0.0, 0.0]}
    counts = {"free": 0, "barbar": 0}
    cfg = O.make_cfg("tensegrity_env", "straight")
    for s in range(n_states):
        e = O.OracleEnv(cfg)
        e.reset_seeded(1234 + seed, s, 0)           # the reference's reset procedure: settled, in contact with the floor
        dd = e.data
        for _ in range(periods):
            ctrl = rng.uniform(m.ctrl_lo, m.ctrl_hi, 6)
            data.qpos[:] = O.arr(dd.qpos); data.qvel[:] = O.arr(dd.qvel); data.ctrl[:] = ctrl
            data.qacc_warmstart[:] = O.arr(dd.qacc_warmstart); data.time = 0.0
            O.arr(dd.ctrl)[:] = ctrl
            mujoco.mj_step(model, data, nstep=20)
            L.tgo_step(C.byref(e.e.model), C.byref(dd), 20)
            q_m, v_m = np.array(data.qpos), np.array(data.qvel)
            q_o, v_o = O.arr(dd.qpos).copy(), O.arr(dd.qvel).copy()
            for b in range(3):                                     # q and -q are the same rotation
                if np.dot(q_m[7 * b + 3:7 * b + 7], q_o[7 * b + 3:7 * b + 7]) < 0:
                    q_o[7 * b + 3:7 * b + 7] *= -1
            eq = float(np.abs(q_m - q_o).max() / max(np.abs(q_m).max(), 1e-3))
            ev = float(np.abs(v_m - v_o).max() / max(np.abs(v_m).max(), 1e-2))
            kind = "barbar" if any(dd.con[i].body1 >= 0 and dd.con[i].body2 >= 0 for i in range(dd.ncon)) else "free"
            counts[kind] += 1
            worst[kind][0] = max(worst[kind][0], eq); worst[kind][1] = max(worst[kind][1], ev)
            # continue from MuJoCo's state so that both always start a period from the identical (state, action)
            O.arr(dd.qpos)[:] = q_m; O.arr(dd.qvel)[:] = v_m; O.arr(dd.qacc_warmstart)[:] = np.array(data.qacc_warmstart)
    out["periods"] = counts
    out["rel_err_qpos_qvel_no_barbar"] = worst["free"]
    out["rel_err_qpos_qvel_barbar"] = worst["barbar"]
    return out


def mujoco_rate(seconds=10.0, workers=1):
    """env-steps/s of the real engine on this host: `workers` processes, each stepping its own MjData with random actions
    (one env-step = mj_step(nstep=20) + the contact-wrench readout the reference's step() performs)."""
    import multiprocessing as mp
    import time

    def work(q, wid):
        import mujoco
        from . import mjcf
        model = mujoco.MjModel.from_xml_string(mjcf.emit_mjcf())
        data = mujoco.MjData(model)
        rng = np.random.default_rng(wid)
        t0 = time.time(); n = 0
        while time.time() - t0 < seconds:
            data.ctrl[:] = rng.uniform(-0.45, 0.15, 6)
            mujoco.mj_step(model, data, nstep=20)
            mujoco.mj_rnePostConstraint(model, data)
            if not np.isfinite(data.qpos).all() or np.abs(data.cfrc_ext).max() > 1500:
                mujoco.mj_resetData(model, data)
            n += 1
        q.put((n, time.time() - t0))

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    ps = [ctx.Process(target=work, args=(q, w)) for w in range(workers)]
    for p in ps:
        p.start()
    res = [q.get() for _ in ps]
    for p in ps:
        p.join()
    return sum(n / dt for n, dt in res)


if __name__ == "__main__":
    import json
    if not available():
        raise SystemExit("mujoco is not importable here: physics parity stays unpinned (see DESIGN.md)")
    print(json.dumps(compare(), indent=1))
