"""Physics restatement (oracle/tgr_physics.c) against the only MuJoCo-generated material the reference ships
 plus self-consistency properties.  Physics vs MuJoCo itself stays "parity unpinned"."""
import ctypes as C
import os

import numpy as np
import pytest


def test_fk_tendon_table_qpos0(oracle):
    """xml:71,90,108 initial pose -> tendon lengths / cap heights recorded in SURVEY.md App. D."""
    O = oracle
    m = O.default_model(); d = O.new_data(m)
    O.lib().tgo_forward(C.byref(m), C.byref(d))
    np.testing.assert_allclose(O.arr(d.ten_length), [0.57384, 0.57174, 0.57109, 0.57362, 0.57200, 0.57000, 1.05355, 1.05407, 1.05429], atol=6e-6)
    np.testing.assert_allclose(O.arr(d.cap_xpos)[:, 2], [0.0499, 0.4467, 0.6513, 0.0497, 0.0498, 0.7409], atol=6e-5)
    assert abs(m.mass - 4.0) < 1e-12 and abs(m.inertia[0] - 1.0964112358) < 1e-9 and abs(m.inertia[2] - 0.003773305) < 1e-9


def test_fk_rolling_poses(oracle):
    """tr_env.py:723-728: settled MuJoCo poses; FK values of SURVEY.md App. D; rows 3,4,5 are bar relabelings."""
    O = oracle
    m = O.default_model(); rq = O.rolling_qpos()
    d = O.new_data(m); O.arr(d.qpos)[:] = rq[0]
    O.lib().tgo_forward(C.byref(m), C.byref(d))
    np.testing.assert_allclose(O.arr(d.ten_length), [0.47567, 0.47418, 0.47287, 0.47659, 0.47329, 0.47162, 1.11797, 1.11944, 1.13383], atol=6e-6)
    np.testing.assert_allclose(rq[4][:7], rq[3][7:14], atol=0)
    np.testing.assert_allclose(rq[5][:7], rq[3][14:21], atol=0)


@pytest.mark.parametrize("pose", range(6))
def test_rolling_poses_are_near_equilibria(oracle, pose):
    """The six poses were produced by MuJoCo's own settle.  With the best-fitting constant ctrl, this restatement must
    leave them (nearly) static: residual generalised forces below 10 N (the captures still carry ~cm/s of motion,
    i.e. a few N of tendon damping), total floor reaction within 3 % of the weight.  A wrong contact stiffness scaling
    (e.g. no 1/dmax^2) or tendon table would miss this by 10 % or more."""
    from scipy.optimize import least_squares
    O = oracle
    L = O.lib(); m = O.default_model(); rq = O.rolling_qpos()
    d = O.new_data(m); O.arr(d.qpos)[:] = rq[pose]
    L.tgo_forward(C.byref(m), C.byref(d))
    tl = O.arr(d.ten_length).copy()

    def res(c):
        O.arr(d.ctrl)[:] = c; O.arr(d.qacc_warmstart)[:] = 0
        L.tgo_forward(C.byref(m), C.byref(d))
        return O.arr(d.qacc).copy()

    c0 = (-100 - 3290 + 6666 * tl[:6]) / 6667
    r = least_squares(res, c0, bounds=(-0.45, 0.15), x_scale=0.01)
    q = res(r.x)
    Md = np.tile([4, 4, 4, m.inertia[0], m.inertia[1], m.inertia[2]], 3)
    assert np.abs(Md * q).max() < 10.0  # newton; tendon and contact forces here are 40-70 N each
    fz = sum(d.con[k].force[0] for k in range(d.ncon))
    assert d.ncon == 3 and abs(fz - 12 * 9.81) < 0.05 * 12 * 9.81
    caps = sorted(O.arr(d.cap_xpos)[:, 2])[:3]
    assert all(0.0670 < z < 0.0675 for z in caps)  # 0.1-0.5 mm penetration of r = 0.0675 caps


def test_settle_reaches_rest_on_three_caps(oracle):
    O = oracle
    L = O.lib(); m = O.default_model(); d = O.new_data(m)
    O.arr(d.ctrl)[:] = -0.15
    L.tgo_step(C.byref(m), C.byref(d), 1500)
    assert d.ncon == 3 and d.flags == 0
    assert np.abs(O.arr(d.qvel)).max() < 0.2
    assert sorted(O.arr(d.cap_xpos)[:, 2])[2] < 0.0675


def test_yaw_invariance_of_dynamics(oracle):
    """Flat floor: rotating the whole state about z commutes with stepping."""
    O = oracle
    L = O.lib(); m = O.default_model()
    outs = []
    for theta in (0.0, 1.234):
        d = O.new_data(m)
        q = np.zeros(21)
        L.tgo_yaw_pose(O.dptr(np.array(m.qpos0)), theta, O.dptr(q))
        O.arr(d.qpos)[:] = q; O.arr(d.ctrl)[:] = [-0.2, -0.1, -0.3, -0.15, -0.25, -0.2]
        L.tgo_step(C.byref(m), C.byref(d), 300)
        outs.append((O.arr(d.qpos).copy(), O.arr(d.ten_length).copy()))
    np.testing.assert_allclose(outs[0][1], outs[1][1], atol=1e-7)
    np.testing.assert_allclose(outs[0][0][[2, 9, 16]], outs[1][0][[2, 9, 16]], atol=1e-7)


def test_bar_relabeling_invariance(oracle):
    """rolling_qpos[3..5] are the same physical pose with the bars renamed cyclically; tendon lengths permute accordingly."""
    O = oracle
    L = O.lib(); m = O.default_model(); rq = O.rolling_qpos()
    ls = []
    for i in (3, 4, 5):
        d = O.new_data(m); O.arr(d.qpos)[:] = rq[i]
        L.tgo_forward(C.byref(m), C.byref(d))
        ls.append(O.arr(d.ten_length).copy())
    assert np.allclose(sorted(ls[0][:6]), sorted(ls[1][:6]), atol=1e-12) and np.allclose(sorted(ls[0][6:]), sorted(ls[2][6:]), atol=1e-12)


def test_philox_known_answer(oracle):
    """Random123 known-answer test for philox4x32-10 (counter = key = 0 and the all-ones vector)."""
    O = oracle
    out = (C.c_uint32 * 4)()
    O.lib().tgo_philox4x32((C.c_uint32 * 4)(0, 0, 0, 0), (C.c_uint32 * 2)(0, 0), out)
    assert [hex(x) for x in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    O.lib().tgo_philox4x32((C.c_uint32 * 4)(*[0xFFFFFFFF] * 4), (C.c_uint32 * 2)(0xFFFFFFFF, 0xFFFFFFFF), out)
    assert [hex(x) for x in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]


def test_reset_draw_distributions(oracle):
    O = oracle
    th, tend = [], []
    for i in range(2000):
        pose = C.c_int(); theta = C.c_double(); t = (C.c_double * 6)(); u = (C.c_double * 2)()
        O.lib().tgo_rng_draws(1234, i, 0, 1, 0.15, 0.2, -0.45, 0.15, 0.0, 2 * np.pi, C.byref(pose), C.byref(theta), t, u)
        th.append(theta.value); tend.append(list(t)); assert 0 <= pose.value < 6
    th = np.array(th); tend = np.array(tend)
    assert 0 <= th.min() and th.max() < 2 * np.pi and abs(th.mean() - np.pi) < 0.15
    assert tend.max() <= 0.15 and tend.min() >= -0.45 and abs(np.mean(tend >= 0.15) - 0.5) < 0.03


@pytest.mark.parametrize("name,direction,need", [("forward", 1, 5), ("backward", -1, 6)])
def test_mujoco_trained_policies_move_the_right_way(oracle, golden_dir, name, direction, need):
    """Behavioural anchor: the SAC actors shipped in best_models_pretrained/ were trained against real
    MuJoCo (weights extracted by oracle/extract_policies.py).  Rolled out deterministically in this restatement
    (termination ignored, 250 control periods = 5 s, 8 reset seeds) most 'forward' rollouts must travel along their reset
    heading and most 'backward' rollouts against it.  This is a plausibility check, not parity: the gaits are chaotic and
    in this simulator the policies hit the 1500 N wrench termination after ~40 steps, far earlier than the ~1600-step
    episodes of their training logs."""
    import os
    O = oracle
    g = np.load(os.path.join(golden_dir, f"policy_{name}.npz"))
    W = [g["latent_pi_0_weight"].astype(np.float64), g["latent_pi_2_weight"].astype(np.float64), g["mu_weight"].astype(np.float64)]
    B = [g["latent_pi_0_bias"].astype(np.float64), g["latent_pi_2_bias"].astype(np.float64), g["mu_bias"].astype(np.float64)]
    lo, hi = float(g["low"]), float(g["high"])
    n = 8
    ob = O.OracleBatch(O.make_cfg("tensegrity_env", "straight", desired_direction=direction), n, seed=1234)
    obs = ob.reset()
    psi0 = np.array([ob.env(i).reset_psi for i in range(n)])
    x0 = np.array([O.arr(ob.env(i).data.xpos)[:, :2].mean(axis=0) for i in range(n)])
    for t in range(250):
        h = obs
        for i in range(2):
            h = np.maximum(h @ W[i].T + B[i], 0)
        a = (lo + 0.5 * (np.tanh(h @ W[2].T + B[2]) + 1) * (hi - lo)).astype(np.float32)
        obs, rew, term, trunc, tobs, info = ob.step(a, auto_reset=False)
    d = info[:, 3:5] - x0
    along = d[:, 0] * np.cos(psi0) + d[:, 1] * np.sin(psi0)
    assert int(np.sum(direction * along > 0.05)) >= need, along
    ob.close()


def test_mujoco_trained_tracking_policy_closes_in_on_its_waypoint(oracle, golden_dir):
    """Behavioural anchor on the *current* XML: models_tracking/SAC_16525000.zip (tr_env "tracking", 48-dim obs, action box
    [-0.45, 0.15], 16.5 M MuJoCo steps) rolled deterministically in this restatement.  is_test puts the waypoint 3 m
    straight ahead of the reset heading (tr_env.py:833-838).  In this simulator the policy never trips the 1500 N rule
    (tr_env.py:480) in 300 control periods and walks at least three of eight robots more than a metre towards the waypoint,
    one of them onto it.  Not parity (a gait is chaotic and pose dependent) but a simulator with wrong tendon, actuator
    or contact scaling does not get a MuJoCo-trained gait to walk at all."""
    import os
    O = oracle
    g = np.load(os.path.join(golden_dir, "policy_tracking.npz"))
    W = [g[k].astype(np.float64) for k in ("latent_pi_0_weight", "latent_pi_2_weight", "mu_weight")]
    B = [g[k].astype(np.float64) for k in ("latent_pi_0_bias", "latent_pi_2_bias", "mu_bias")]
    lo, hi = float(g["low"]), float(g["high"])
    n = 8
    ob = O.OracleBatch(O.make_cfg("tr_env", "tracking", is_test=1), n, seed=1234)
    obs = ob.reset()
    d0 = dmin = None
    tend = []
    wrench_trip = np.zeros(n, bool)
    for t in range(300):
        h = obs
        for i in range(2):
            h = np.maximum(h @ W[i].T + B[i], 0)
        a = (lo + 0.5 * (np.tanh(h @ W[2].T + B[2]) + 1) * (hi - lo)).astype(np.float32)
        obs, rew, term, trunc, tobs, info = ob.step(a, auto_reset=False)
        d = np.linalg.norm(info[:, 10:12] - info[:, 3:5], axis=1)
        if t >= 100:
            tend.append(info[:, 16:25].copy())
        if d0 is None:
            d0, dmin = d.copy(), d.copy()
        dmin = np.minimum(dmin, d)
        for i in range(n):
            wrench_trip[i] |= bool(np.abs(O.arr(ob.env(i).data.cfrc_ext)).max() > 1500)
    assert np.all(np.abs(d0 - 3.0) < 0.1)
    assert not wrench_trip.any()
    assert int(np.sum(d0 - dmin > 1.0)) >= 3 and dmin.min() < 0.5, (d0 - dmin)
    # the one real MuJoCo state of this gait the reference ships: the checkpoint's `_last_obs` -- tendon
    # lengths of the robot mid-gait.  They must be ordinary values of the same gait in this simulator.
    mj_act = np.array([.322809, .490437, .448440, .385049, .364695, .544164]); mj_pas = np.array([1.207020, 1.200822, 1.198159])
    T = np.concatenate(tend)
    lo_a, hi_a = np.quantile(T[:, :6], [0.05, 0.95]); lo_p, hi_p = np.quantile(T[:, 6:], [0.05, 0.95])
    assert lo_a < mj_act.min() and mj_act.max() < hi_a, (lo_a, hi_a)
    assert lo_p < mj_pas.min() and mj_pas.max() < hi_p and abs(T[:, 6:].mean() - mj_pas.mean()) < 0.03, (lo_p, hi_p, T[:, 6:].mean())
    ob.close()


def test_uneven_model_settles_on_the_heightfield_and_carries_its_weight(oracle, golden_dir):
    """uneven-ground model: reset_model() drops the robot ~0.8 m onto terrain_image_100.png's heightfield; after
    the 50 settle periods plus 2 s at constant tendon targets it stands on the terrain (caps within a cap radius of the surface
    between z = -1 and 0, never on a z = 0 plane), nearly at rest, and the contact forces balance gravity."""
    import os
    import ctypes as C
    from tensegrity_rl_amd.vec_env import heights_from_image
    O = oracle
    h = heights_from_image(np.load(os.path.join(golden_dir, "terrain_100.npy")))
    assert h.shape == (100, 100) and h.min() == 0.0 and h.max() == 1.0
    cfg = O.make_cfg("tensegrity_env", "straight", heights=h, model_kind=1, terminate_when_unhealthy=0)
    for seed in (0, 1):
        e = O.OracleEnv(cfg)
        e.reset_seeded(7, seed, 0)
        d = e.data
        a = O.arr(d.ctrl).astype(np.float32).copy()
        for _ in range(100):
            e.step(a)
        z = O.arr(d.qpos).reshape(3, 7)[:, 2]
        assert z.max() < 0.2 and z.min() > -1.1
        assert np.abs(O.arr(d.qvel).reshape(3, 6)[:, :3]).max() < 0.05 and np.abs(O.arr(d.qvel)).max() < 0.3  # the activations are still creeping
        fzs = []
        for _ in range(25):   # soft contacts chatter from sub-step to sub-step: average the snapshots of 25 control periods
            e.step(a)
            fz = 0.0
            for i in range(d.ncon):
                c = d.con[i]
                if c.body1 == -1:  # terrain contact: world force on the bar = sum_j force_j * frame_row_j
                    fr = np.array(list(c.frame)).reshape(3, 3)
                    fz += c.force[0] * fr[0, 2] + c.force[1] * fr[1, 2] + c.force[2] * fr[2, 2]
            fzs.append(fz)
        assert abs(np.mean(fzs) - 12 * 9.81) < 0.05 * 12 * 9.81, (np.mean(fzs), fzs[:5])
        assert np.abs(O.arr(d.act) - np.clip(a, -0.45, 0.15)).max() < 0.1   # activations have followed the targets (tau = 1 s, 3 s)


# ------------------------------------------------------------------ exact inverse-kinematics checks on MuJoCo-made numbers
# `_last_original_obs` of best_models_pretrained/forward/SAC_5500000.zip -- a 39-dim tensegrity_env observation
# (tensegrity_env.py:412-430) written by real MuJoCo mid-gait: three geom-frame quaternions (x, y, z, w) and nine tendon lengths, six
# decimals each.  Three orientations + nine lengths over-determine the six relative bar coordinates by three.
MJ_GEOM_QUATS = np.array([[-.180353, -.613728, .764173, .082762], [.243167, .726228, -.642295, -.030328], [-.358620, -.622310, .679296, .150597]])
MJ_TENDONS = np.array([.223362, .222672, .236712, .218046, .159673, .181029, 1.290653, 1.290078, 1.291460])


def _body_quat_from_obs(g):
    """invert what _get_obs does: the observation holds scipy's (x, y, z, w) of the r-cylinder's GEOM frame = body frame * quat(0,0,0,1)
    (flat xml:72): geom (x, y, z, w) = body (y, -x, w, -z)  ->  body (w, x, y, z) = (gz, -gy, gx, -gw)"""
    q = np.array([g[2], -g[1], g[0], -g[3]])
    return q / np.linalg.norm(q)


def _fit_relative_positions(O, m, quats, tendons, starts=24):
    from scipy.optimize import least_squares
    L = O.lib(); d = O.new_data(m)

    def res(x):
        q = np.zeros(21)
        for b in range(3):
            q[7 * b + 3:7 * b + 7] = quats[b]
        q[0:3] = (0, 0, 5.0); q[7:10] = x[:3] + (0, 0, 5.0); q[14:17] = x[3:] + (0, 0, 5.0)  # far above any floor: kinematics only
        O.arr(d.qpos)[:] = q
        L.tgo_forward(C.byref(m), C.byref(d))   # the oracle's own kinematics + tendon routine
        return O.arr(d.ten_length) - tendons
    rng = np.random.default_rng(0)
    best = None
    for _ in range(starts):
        r = least_squares(res, rng.normal(0, 0.4, 6), xtol=1e-15, ftol=1e-15, gtol=1e-15)
        if best is None or r.cost < best.cost:
            best = r
    return best


def test_mujoco_recorded_pose_fits_the_uneven_models_tendon_table_exactly(oracle):
    """The recorded quaternions and tendon lengths are one rigid configuration of the UNEVEN-ground XML's robot (sites 0.05 m off the axis
    in the permuted layout of uneven xml:76-81,122-158 -- the model these checkpoints were trained on, which is also why their action box
    is the stale [-0.45, -0.15]): solving the six relative coordinates by least squares on the oracle's own forward
    kinematics leaves 2.3e-7 m rms -- the truncation of the six-decimal record -- on nine equations.  That pins, against MuJoCo's own
    output at a non-trivial state: the site table and tendon connectivity of model_kind 1, site_z = 0.668, the geom-frame convention
    (body * 180 deg about z), scipy's scalar-last order and the tendon routine.  (Any wrong pairing, offset or convention misses by
    centimetres: see the next test.)"""
    O = oracle
    quats = [_body_quat_from_obs(g) for g in MJ_GEOM_QUATS]
    r = _fit_relative_positions(O, O.uneven_model(None), quats, MJ_TENDONS)
    assert np.sqrt(np.mean(r.fun ** 2)) < 1e-6 and np.abs(r.fun).max() < 2e-6, r.fun
    # and it is a sensible robot: bar centres 0.05 .. 0.5 m apart, no two bars through each other's axis
    p = [np.zeros(3), r.x[:3], r.x[3:]]
    dist = [np.linalg.norm(p[a] - p[b]) for a, b in ((0, 1), (0, 2), (1, 2))]
    assert 0.05 < min(dist) and max(dist) < 0.5, dist


def test_mujoco_recorded_pose_rejects_wrong_tables_and_conventions(oracle):
    """the same fit is not forgiving: with the FLAT model's table (sites at 0.0675 in the other layout -- the checkpoint predates it), a
    vector-first reading of the quaternions, or without the geom frame's half turn, nine tendon lengths cannot be matched to better than
    centimetres"""
    O = oracle
    good = [_body_quat_from_obs(g) for g in MJ_GEOM_QUATS]
    no_half_turn = [np.array([g[3], g[0], g[1], g[2]]) for g in MJ_GEOM_QUATS]          # geom frame taken for the body frame
    scalar_first = [_body_quat_from_obs(np.roll(g, -1)) for g in MJ_GEOM_QUATS]         # (w, x, y, z) reading of the record
    for model, quats in ((O.default_model(), good), (O.uneven_model(None), no_half_turn), (O.uneven_model(None), scalar_first)):
        r = _fit_relative_positions(O, model, quats, MJ_TENDONS, starts=8)
        assert np.sqrt(np.mean(r.fun ** 2)) > 3e-3, r.fun


# ------------------------------------------------------------------ all 35 checkpoints, full precision (tests/golden/mujoco_last_obs.json)
def _quat_z_to(axis):
    a = np.asarray(axis, float) / np.linalg.norm(axis)
    c = np.cross([0, 0, 1.0], a); w = 1.0 + a[2]
    q = np.concatenate([[w], c]) if w > 1e-9 else np.array([0.0, 1.0, 0.0, 0.0])
    return q / np.linalg.norm(q)


def _qmul(a, b):
    aw, ax, ay, az = a; bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])


def _fit_spins(O, m, caps, tendons, starts=8, want_qpos=False):
    """tr_env record: the six cap positions fix every bar's centre and axis; what is left is each bar's spin about its own axis -- three
    unknowns for nine tendon lengths"""
    from scipy.optimize import least_squares
    L = O.lib(); d = O.new_data(m)
    xs = [(caps[2 * b] + caps[2 * b + 1]) / 2 for b in range(3)]
    q0 = [_quat_z_to(caps[2 * b] - caps[2 * b + 1]) for b in range(3)]   # cap s_2b sits at +z of its bar (flat xml:75-76)

    def res(phi):
        q = np.zeros(21)
        for b in range(3):
            q[7 * b:7 * b + 3] = xs[b] + np.array([0, 0, 5.0])
            q[7 * b + 3:7 * b + 7] = _qmul(q0[b], np.array([np.cos(phi[b] / 2), 0, 0, np.sin(phi[b] / 2)]))
        O.arr(d.qpos)[:] = q
        L.tgo_forward(C.byref(m), C.byref(d))
        return O.arr(d.ten_length) - tendons
    rng = np.random.default_rng(0)
    best = np.inf
    for _ in range(starts):
        r = least_squares(res, rng.uniform(-np.pi, np.pi, 3), xtol=1e-15, ftol=1e-15, gtol=1e-15)
        best = min(best, float(np.sqrt(np.mean(r.fun ** 2))))
        if best < 1e-12:
            res(r.x)
            break
    return (best, O.arr(d.qpos).copy()) if want_qpos else best      # (qpos: of the last evaluation -- the solution when best < 1e-12)


def _with_site_offset(m, off):
    for i in range(6):
        x, y = m.site_local[i][0], m.site_local[i][1]
        r = np.hypot(x, y)
        m.site_local[i][0] = x / r * off; m.site_local[i][1] = y / r * off
    return m


def test_every_mujoco_recorded_observation_fits_a_site_table_to_machine_precision(oracle, golden_dir):
    """tests/golden/mujoco_last_obs.json (oracle/extract_last_obs.py): `_last_obs` and `_last_original_obs` of all 35 SB3 checkpoints the
    reference ships, at full double precision -- 70 observations written by real MuJoCo (the reset observation and a mid-gait one per
    checkpoint).  tensegrity_env (39): quaternions + tendon lengths, six unknowns for nine equations; tr_env (27 / 45 / 48): the cap positions
    fix centres and axes, three spins for nine equations.  Solved on the oracle's own kinematics + tendon routine:
      * every one of the 22 tensegrity_env observations and 34 of the tr_env ones fit the UNEVEN XML's table (sites 0.05 m off the axis, layout
        of uneven xml:76-81,122-158) to 1e-15 m rms -- machine precision on 9 equations;
      * the 12 observations of six `models_aiming` checkpoints (_tr, _t, _cw, _ccw) fit the FLAT XML's layout and connectivity (flat
        xml:82-87,127-164) with the sites 0.05 m off the axis, again to 1e-15: an intermediate version of the model -- today's flat XML
        moved the sites out to the 0.0675 m cap radius.  That pins the flat model's site layout and tendon connectivity against MuJoCo (the
        offset itself is a constant tests/test_model_constants.py reads off the XML);
      * models_forward/SAC_19800000 holds noisy observations (use_obs_noise: its cap pairs are 1.32 .. 1.38 m apart instead of 1.376) and is
        recognised as such.
    Besides: cap pairs are 2 x 0.688 m apart and the six relative positions sum to zero to 1e-12 (tr_env's observation layout, tr_env.py:536-552),
    cap velocities of a bar differ by something perpendicular to it (:586-604), the tracking tail is (waypt - centre, atan2) (:626-636)."""
    import json
    O = oracle
    recs = json.load(open(os.path.join(golden_dir, "mujoco_last_obs.json")))["records"]
    assert len(recs) == 35
    mu = O.uneven_model(None)
    flat05 = _with_site_offset(O.default_model(), 0.05)
    flat_now = O.default_model()
    n_uneven = n_flat05 = n_noisy = 0
    for r in recs:
        for key in ("_last_obs", "_last_original_obs"):
            o = np.array(r[key]); dim = r["obs_dim"]
            assert o.shape == (dim,) and np.isfinite(o).all()
            if dim == 39:
                quats = [_body_quat_from_obs(o[4 * b:4 * b + 4]) for b in range(3)]
                assert max(abs(np.linalg.norm(o[4 * b:4 * b + 4]) - 1) for b in range(3)) < 1e-12
                f = _fit_relative_positions(O, mu, quats, o[30:39], starts=6)
                assert np.sqrt(np.mean(f.fun ** 2)) < 1e-12, (r["file"], key, f.fun)
                n_uneven += 1
                continue
            caps = o[:18].reshape(6, 3)
            tend = o[18:27] if dim == 27 else o[36:45]
            lens = [np.linalg.norm(caps[2 * b] - caps[2 * b + 1]) for b in range(3)]
            if max(abs(x - 1.376) for x in lens) > 1e-9:        # observation noise on the cap positions (sigma 0.05 m)
                assert r["file"].endswith("models_forward/SAC_19800000.zip")
                n_noisy += 1
                continue
            assert np.abs(caps.sum(0)).max() < 1e-12
            if dim >= 45:
                vel = o[18:36].reshape(6, 3)
                for b in range(3):
                    assert abs(np.dot(vel[2 * b] - vel[2 * b + 1], caps[2 * b] - caps[2 * b + 1])) < 1e-10
            if dim == 48:
                assert abs(np.arctan2(o[46], o[45]) - o[47]) < 1e-12
            a = _fit_spins(O, mu, caps, tend)
            if a < 1e-12:
                n_uneven += 1
                continue
            b_ = _fit_spins(O, flat05, caps, tend)
            assert b_ < 1e-12, (r["file"], key, a, b_)
            assert "models_aiming" in r["file"]
            assert _fit_spins(O, flat_now, caps, tend, starts=3) > 1e-3      # (not today's 0.0675 m offset)
            n_flat05 += 1
    assert (n_uneven, n_flat05, n_noisy) == (22 + 34, 12, 2), (n_uneven, n_flat05, n_noisy)


def test_forward_policy_belongs_to_the_uneven_xml_robot(oracle, golden_dir):
    """Why best_models_pretrained/forward trips the 1500 N rule (tensegrity_env.py:371) after ~40 steps on the CURRENT flat XML although its
    training log shows ~1600-step episodes: it was trained on the robot of the uneven-ground XML (the inverse-kinematics test above pins
    that from its recorded observation; its action box is that model's [-0.45, -0.15]).  Rolled out deterministically in this restatement,
      * on the current flat XML (700 N/m passive tendons, affine actuators) every one of 16 rollouts trips within 150 steps -- cap-on-floor
        impacts of 2.3 .. 2.9 kN at 1 .. 1.4 mm depth and r-cylinder crossings, bars spinning at 13 .. 34 rad/s: the policy's commands throw
        the softer robot about;
      * on the uneven XML's robot (1e4 N/m tendons, gain-15000 filter actuators, 5 cm caps) standing on a flat floor, at most 2 of 16 trip
        within 400 steps and the robots travel ~2 m in 8 s.
    So the short episodes are a property of pairing that checkpoint with the newer XML, not a sign of inflated contact forces here; and the
    uneven model's tendon / actuator constants carry a MuJoCo-trained gait for 8 s."""
    import os
    O = oracle
    g = np.load(os.path.join(golden_dir, "policy_forward.npz"))
    W = [g[k].astype(np.float64) for k in ("latent_pi_0_weight", "latent_pi_2_weight", "mu_weight")]
    B = [g[k].astype(np.float64) for k in ("latent_pi_0_bias", "latent_pi_2_bias", "mu_bias")]
    lo, hi = float(g["low"]), float(g["high"])
    n = 16
    out = {}
    for name, kw, steps in (("flat", {}, 150), ("uneven-robot", dict(model_kind=1), 400)):
        ob = O.OracleBatch(O.make_cfg("tensegrity_env", "straight", **kw), n, seed=1234)
        obs = ob.reset()
        x0 = np.array([O.arr(ob.env(i).data.xpos)[:, :2].mean(axis=0) for i in range(n)])
        tripped = np.zeros(n, bool)
        for t in range(steps):
            h = obs
            for i in range(2):
                h = np.maximum(h @ W[i].T + B[i], 0)
            a = (lo + 0.5 * (np.tanh(h @ W[2].T + B[2]) + 1) * (hi - lo)).astype(np.float32)
            obs = ob.step(a, auto_reset=False)[0]
            for i in range(n):
                tripped[i] |= bool(np.abs(O.arr(ob.env(i).data.cfrc_ext)).max() > 1500)
        travel = np.linalg.norm(np.array([O.arr(ob.env(i).data.xpos)[:, :2].mean(axis=0) for i in range(n)]) - x0, axis=1)
        out[name] = (int(tripped.sum()), float(travel.mean()))
        ob.close()
    assert out["flat"][0] == n, out
    assert out["uneven-robot"][0] <= 2 and out["uneven-robot"][1] > 1.0, out
