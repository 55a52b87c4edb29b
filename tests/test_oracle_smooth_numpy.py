"""The oracle's contact-free sub-step against an independent numpy restatement of the smooth dynamics (tests/smooth_numpy.py): tendon
geometry, spring / damper / actuator forces incl. the force clamp and the filter activations, gyroscopic bias, implicitfast, quaternion
integration -- both models, 20 sub-steps, robots in the air.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import smooth_numpy as SN


@pytest.mark.parametrize("model", ["flat", "uneven"])
def test_oracle_airborne_substeps_match_the_numpy_restatement(oracle, model):
    O = oracle
    L = O.lib()
    m = O.default_model() if model == "flat" else O.uneven_model(None)
    c = SN.Consts(m)
    assert c.filter == (model == "uneven")
    rng = np.random.default_rng(5)
    clamped = 0
    for trial in range(6):
        d = O.new_data(m)
        q = np.array(m.qpos0[:]) if trial % 2 == 0 else O.rolling_qpos()[trial % 6].copy()
        q = q.reshape(3, 7); q[:, 2] += 5.0                                   # in the air: neither floor nor terrain
        q[:, 3:7] += rng.normal(0, 0.02, (3, 4)); q[:, 3:7] /= np.linalg.norm(q[:, 3:7], axis=1, keepdims=True)
        q = q.reshape(21)
        v = np.concatenate([np.concatenate([rng.uniform(-0.5, 0.5, 3), rng.uniform(-2, 2, 3)]) for _ in range(3)])
        ctrl = rng.uniform(-0.6, 0.3, 6)                                      # beyond ctrlrange on purpose: the clip is part of the path
        act = rng.uniform(-0.3, 0.1, 6) if c.filter else np.zeros(6)
        O.arr(d.qpos)[:] = q; O.arr(d.qvel)[:] = v; O.arr(d.ctrl)[:] = ctrl; O.arr(d.act)[:] = act
        for s in range(20):
            q, v, act, ft = SN.substep(c, q, v, ctrl, act)
            L.tgo_step(C.byref(m), C.byref(d), 1)
            assert d.ncon == 0
            if not c.filter:
                clamped += int(np.sum((np.array(d.actuator_force[:]) <= c.frange[0]) | (np.array(d.actuator_force[:]) >= c.frange[1])))
        q2 = O.arr(d.qpos).copy(); v2 = O.arr(d.qvel).copy()
        for b in range(3):                                                    # q and -q are the same rotation
            if np.dot(q2[7 * b + 3:7 * b + 7], q[7 * b + 3:7 * b + 7]) < 0:
                q2[7 * b + 3:7 * b + 7] *= -1
        assert np.abs(q2 - q).max() < 1e-11, np.abs(q2 - q).max()
        assert np.abs(v2 - v).max() < 1e-9 * max(1.0, np.abs(v).max()), np.abs(v2 - v).max()
        if c.filter:
            assert np.abs(O.arr(d.act) - act).max() < 1e-13
    if not c.filter:
        assert clamped > 0        # the force clamp [-267, 0] was exercised
