"""-m gpu: closed-form known answers the reference's model constants imply (no oracle involved).

With the actuators and tendon springs switched off (raw tgr_config fields) the three bars are free bodies.  A bar laid flat on the floor
rests on its two caps; at rest every cap carries half the bar's weight, and MuJoCo's soft-contact law for the XML's
solref=(-5000,-500) / solimp=(0.9,0.95,0.001,0.5,2) (3prism_jonathan_steady_side.xml:30) fixes the penetration p in closed form:
    f(p) = D(imp) * K * imp * p,   K = 5000 / dmax^2,   D = imp / ((1 - imp) * invweight),   invweight = 1/m = 0.25,
    imp(p) = d0 + y(p / width) (dmax - d0),  y = 2 x^2 for x <= 1/2  (power-2 sigmoid, midpoint 1/2)
so  f(p) = m g / 2 = 19.62 N  gives p = 0.1078 mm.  cfrc_ext of a resting bar is then (torque 0, force (0, 0, m g)), which the env shows
through contact_cost = 5e-4 * sum(clip(cfrc_ext)^2) = 5e-4 * 3 * (4 * 9.81)^2 (tensegrity_env.py:256-268)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M_BAR, G, R_CAP, HALF = 4.0, 9.81, 0.0675, 0.688


def contact_force(p):
    d0, dmax, width, mid = 0.9, 0.95, 0.001, 0.5
    x = min(p / width, 1.0)
    y = x * x / mid if x <= mid else 1 - (1 - x) ** 2 / (1 - mid)
    imp = d0 + y * (dmax - d0)
    K = 5000.0 / dmax ** 2
    D = imp / ((1 - imp) * (1.0 / M_BAR))
    return D * K * imp * p


def rest_penetration(load):
    lo, hi = 0.0, 1e-3
    for _ in range(80):
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if contact_force(mid) < load else (lo, mid)
    return 0.5 * (lo + hi)


@pytest.mark.parametrize("prec,layout", [("f64", "row"), ("f32", "row"), ("f64", "block")])
def test_free_bars_lying_flat_rest_at_the_closed_form_penetration(prec, layout):
    from tensegrity_rl_amd import TensegrityVecEnv
    n = 16
    free = dict(gain=0.0, bias=[0.0, 0.0, 0.0], ten_stiffness=[0.0] * 9, ten_damping=[0.0] * 9)
    env = TensegrityVecEnv(n, "tensegrity_env-v0", precision=prec, auto_reset=False, kernel_layout=layout, use_contact_forces=1,
                           contact_force_range=(-1e4, 1e4), terminate_when_unhealthy=0, config_overrides=free)
    p_star = rest_penetration(M_BAR * G / 2)
    assert abs(p_star - 1.0776e-4) < 1e-7 and abs(contact_force(p_star) - M_BAR * G / 2) < 1e-9
    rng = np.random.default_rng(0)
    qpos = np.zeros((n, 21)); qvel = np.zeros((n, 18))
    for e in range(n):
        for b in range(3):
            yaw = rng.uniform(0, 2 * np.pi)
            # bar axis (local z) along a horizontal direction: rotate by 90 deg about the horizontal axis perpendicular to it
            ax = np.array([-np.sin(yaw), np.cos(yaw), 0.0])
            q = np.concatenate([[np.cos(np.pi / 4)], np.sin(np.pi / 4) * ax])
            qpos[e, 7 * b:7 * b + 3] = [3.0 * b + rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), R_CAP + 2e-4]  # dropped from 0.2 mm
            qpos[e, 7 * b + 3:7 * b + 7] = q
    env.set_state(qpos, qvel)
    zero = np.zeros((n, 6), dtype=np.float32)
    for _ in range(70):  # 1.4 s = 13 time constants of the over-damped contact (K imp / B = 9.5 1/s)
        obs, rew, term, trunc, tobs, info = env.step_tensors(zero)
    st = env.get_state()
    z = st["qpos"].reshape(n, 3, 7)[:, :, 2]
    tol = 2e-7 if prec == "f64" else 2e-6
    assert np.abs(z - (R_CAP - p_star)).max() < tol, (z.min(), z.max(), R_CAP - p_star)
    # (fp32: a free bar can roll about its own axis -- nothing but the 1e-4 rolling friction opposes it -- so rounding leaves a slow spin)
    assert np.abs(st["qvel"]).max() < (1e-6 if prec == "f64" else 1e-3)
    # cfrc_ext: force (0, 0, m g) and no torque on each bar -> contact_cost = 5e-4 * 3 * (m g)^2  (info["reward_ctrl"] = -contact_cost,
    # the reference's overwrite: tensegrity_env.py:404)
    cc = -info.cpu().numpy()[:, 1]
    want = 5e-4 * 3 * (M_BAR * G) ** 2
    assert np.abs(cc - want).max() < (1e-6 if prec == "f64" else 2e-3) * want, (cc.min(), cc.max(), want)
    env.close()


# ------------------------------------------------------------------ bar against bar, one sub-step, closed form (no oracle involved)
R_CYL, R_BCYL = 0.0381, 0.035


def _imp(p):
    d0, dmax, width, mid = 0.9, 0.95, 0.001, 0.5
    x = min(p / width, 1.0)
    y = x * x / mid if x <= mid else 1 - (1 - x) ** 2 / (1 - mid)
    return d0 + y * (dmax - d0)


def _quat_z_to(axis):
    """unit quaternion (w, x, y, z) turning the local z axis (the bar's axis) into `axis`"""
    a = np.asarray(axis, float) / np.linalg.norm(axis)
    c = np.cross([0, 0, 1.0], a); w = 1.0 + a[2]
    q = np.concatenate([[w], c]) if w > 1e-9 else np.array([0.0, 1.0, 0.0, 0.0])
    return q / np.linalg.norm(q)


def _free_env(prec, n=4):
    from tensegrity_rl_amd import TensegrityVecEnv
    free = dict(gain=0.0, bias=[0.0, 0.0, 0.0], ten_stiffness=[0.0] * 9, ten_damping=[0.0] * 9, gravity=0.0, frame_skip=1)
    return TensegrityVecEnv(n, "tensegrity_env-v0", precision=prec, auto_reset=False, terminate_when_unhealthy=0, config_overrides=free)


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("case", ["cap on a bar's side", "two bars crossing"])
def test_one_substep_of_a_central_bar_bar_contact_in_closed_form(prec, case):
    """Two free bars (no gravity, tendons and actuators off, frame_skip = 1) at rest, touching so that the contact point lies on the line
    through both centres of mass along the contact normal: no torque, no tangential motion, and mujoco's soft-contact problem for the one
    normal row is a scalar quadratic --
        aref = K imp p,  R = (1 - imp) / imp * (1/m + 1/m),  f = aref / (R + 2/m),  bar B gains +h f / m along the normal, bar A loses it
    (K = 5000 / 0.95^2, imp = solimp sigmoid at penetration p; flat xml:30).  Checked after ONE sub-step for
      * a cap sphere of bar B resting on the SIDE of bar A's r-cylinder above A's centre (the sphere-cylinder routine: "a bar end resting on
        another bar's side"), penetration 0.5 mm;
      * the r-cylinders of two perpendicular bars crossing above each other's centres (cylinder-cylinder side to side), 0.3 mm."""
    env = _free_env(prec)
    n = env.num_envs
    h = 1e-3
    K = 5000.0 / 0.95 ** 2
    if case == "cap on a bar's side":
        p = 5e-4
        qA, xA = _quat_z_to([1, 0, 0]), np.array([0.0, 0.0, 5.0])
        qB = _quat_z_to([0, 0, 1])
        cap = xA + np.array([0, 0, R_CYL + R_CAP - p])            # B's lower cap (-z end) centre, straight above A's centre
        xB = cap + np.array([0, 0, HALF])
    else:
        p = 3e-4
        qA, xA = _quat_z_to([1, 0, 0]), np.array([0.0, 0.0, 5.0])
        qB, xB = _quat_z_to([0, 1, 0]), np.array([0.0, 0.0, 5.0 + 2 * R_CYL - p])
    imp = _imp(p)
    f = K * imp * p / ((1 - imp) / imp * (2.0 / M_BAR) + 2.0 / M_BAR)
    qpos = np.zeros((n, 21)); qvel = np.zeros((n, 18))
    for e in range(n):
        qpos[e, 0:3], qpos[e, 3:7] = xA, qA
        qpos[e, 7:10], qpos[e, 10:14] = xB, qB
        qpos[e, 14:17], qpos[e, 17:21] = [0.0, 7.0, 5.0], [1, 0, 0, 0]      # the third bar: far away
    env.set_state(qpos, qvel)
    env.step_tensors(np.zeros((n, 6), dtype=np.float32))
    st = env.get_state()
    v = st["qvel"].reshape(n, 3, 6)
    want = h * f / M_BAR
    # fp32: the bar-bar routine works on fp32 coordinates RELATIVE to bar A's centre (the difference is formed in fp64): a cap 0.8 m from
    # that centre is placed to 6e-8 m, i.e. to 1e-4 of a 0.5 mm penetration
    tol = 1e-10 if prec == "f64" else 2e-4 * want
    assert abs(f - (4.7403 if case.startswith("cap") else 2.7466)) < 2e-3        # (N; the numbers the docstring's formula gives)
    assert np.abs(v[:, 1, 2] - want).max() < tol and np.abs(v[:, 0, 2] + want).max() < tol, (v[0, :2, 2], want)
    rest = v.copy(); rest[:, 0, 2] = 0; rest[:, 1, 2] = 0
    assert np.abs(rest).max() < tol                                               # nothing else moves: no torque, no friction force
    z = st["qpos"].reshape(n, 3, 7)[:, :2, 2]
    assert np.abs(z[:, 1] - (xB[2] + h * want)).max() < (1e-12 if prec == "f64" else 1e-6)
    assert env.stats()["dropped_contacts"] == 0
    env.close()


def test_a_deep_bar_crossing_carries_the_three_contacts_mujoco_would_keep():
    """r-cylinders crossing 4 mm deep at 0.4 m from both centres, where both bars also carry a b-cylinder (r = 0.035 inside r = 0.0381,
    flat xml:77): mujoco instantiates r-r (4 mm), r-b and b-r (0.9 mm each) -- b-b would need 6.2 mm.  All three share the normal and the
    point; the row kernels keep them on the pair's side-to-side lane (6) and the first two pool lanes (7, 9).  Contacts are unilateral: the deep one alone pushes the
    bars apart harder (relative normal acceleration 20 m/s^2) than the two shallow ones ask for (aref = K imp 0.9 mm = 4.7 m/s^2), so those
    stay force-free, and with the deep contact's rows in the quadratic zone of its cone the sub-step is the linear problem
    (M + J^T D J) a = J^T D aref, solved here with numpy from the geometry alone (six rows: normal, two tangents, torsion, two rolling;
    R = (1 - imp) / imp * (1/m + 1/m) for the normal and tangential rows, R mu0^2 / mu_j^2 for torsion and rolling: SURVEY.md App. B.5) and
    compared with the velocities after one sub-step (fp64): off-centre, so bars translate AND rotate, and friction rows carry force."""
    env = _free_env("f64")
    n = env.num_envs
    h, K = 1e-3, 5000.0 / 0.95 ** 2
    Ixx = 1.0 * (3 * R_CYL ** 2 + (2 * HALF) ** 2) / 12 + 2 * (0.4 * 0.5 * R_CAP ** 2 + 0.5 * HALF ** 2) + 2 * (1.0 * (3 * R_BCYL ** 2 + 0.45 ** 2) / 12 + 0.463 ** 2)
    Izz = 1.0 * R_CYL ** 2 / 2 + 2 * 0.4 * 0.5 * R_CAP ** 2 + 2 * 1.0 * R_BCYL ** 2 / 2
    s0 = 0.4
    pen = 4e-3
    xA, aA = np.array([0.0, 0.0, 5.0]), np.array([1.0, 0, 0])
    xB, aB = np.array([s0, -s0, 5.0 + 2 * R_CYL - pen]), np.array([0, 1.0, 0])      # B's axis passes over A's at A's s = 0.4, B's t = 0.4
    nrm = np.array([0, 0, 1.0])
    pa = xA + s0 * aA
    contacts = []   # (dist, position): r-r, r(A)-b(B), b(A)-r(B); position = pa + (r_A + dist / 2) n
    for ra, rb in ((R_CYL, R_CYL), (R_CYL, R_BCYL), (R_BCYL, R_CYL)):
        dist = (2 * R_CYL - pen) - (ra + rb)
        assert dist < 0
        contacts.append((dist, pa + (ra + 0.5 * dist) * nrm))
    assert (2 * R_CYL - pen) - 2 * R_BCYL > 0   # b-b does not touch
    # generalised coordinates: (v_A, w_A, v_B, w_B) in the world frame; M = diag(m, I_A, m, I_B) with I = Ixx 1 + (Izz - Ixx) a a^T
    inert = lambda a: Ixx * np.eye(3) + (Izz - Ixx) * np.outer(a, a)
    Mm = np.zeros((12, 12)); Mm[:3, :3] = M_BAR * np.eye(3); Mm[3:6, 3:6] = inert(aA); Mm[6:9, 6:9] = M_BAR * np.eye(3); Mm[9:, 9:] = inert(aB)
    t1, t2 = np.array([1.0, 0, 0]), np.array([0, 1.0, 0])
    mu = [1.0, 1.0, 0.005, 1e-4, 1e-4]
    skew = lambda r: np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])

    def rows_of(pos):
        rA_, rB_ = pos - xA, pos - xB
        Jp = np.hstack([-np.eye(3), skew(rA_), np.eye(3), -skew(rB_)])     # velocity of the contact point on B minus on A
        Jr = np.hstack([np.zeros((3, 3)), -np.eye(3), np.zeros((3, 3)), np.eye(3)])
        return [nrm @ Jp, t1 @ Jp, t2 @ Jp, nrm @ Jr, t1 @ Jr, t2 @ Jr]
    dist, pos = contacts[0]
    imp = _imp(-dist)
    R0 = (1 - imp) / imp * (0.25 + 0.25)
    R = [R0, R0, R0, R0 * mu[0] ** 2 / mu[2] ** 2, R0 * mu[0] ** 2 / mu[3] ** 2, R0 * mu[0] ** 2 / mu[4] ** 2]   # elliptic cone: R_j = R_1 mu_0^2 / mu_j^2
    H, g = Mm.copy(), np.zeros(12)
    for J, Rj, ar in zip(rows_of(pos), R, [-K * imp * dist, 0, 0, 0, 0, 0]):
        H += np.outer(J, J) / Rj; g += J * ar / Rj
    a = np.linalg.solve(H, g)
    jar = [J @ a - ar for J, ar in zip(rows_of(pos), [-K * imp * dist, 0, 0, 0, 0, 0])]
    assert mu[0] * jar[0] + np.sqrt(sum((j * f) ** 2 for j, f in zip(jar[1:], mu))) <= 0     # the deep contact sits in the quadratic zone
    for dist2, pos2 in contacts[1:]:                                                       # the shallow ones are satisfied: no force
        assert rows_of(pos2)[0] @ a - (-K * _imp(-dist2) * dist2) > 1.0
    qpos = np.zeros((n, 21)); qvel = np.zeros((n, 18))
    for e in range(n):
        qpos[e, 0:3], qpos[e, 3:7] = xA, _quat_z_to(aA)
        qpos[e, 7:10], qpos[e, 10:14] = xB, _quat_z_to(aB)
        qpos[e, 14:17], qpos[e, 17:21] = [0.0, 7.0, 5.0], [1, 0, 0, 0]
    env.set_state(qpos, qvel)
    trace = env.bind_substep_trace()
    env.step_tensors(np.zeros((n, 6), dtype=np.float32))
    st = env.get_state()
    assert ((trace.cpu().numpy()[:, 0] & 0xffff) == ((1 << 6) | (1 << 7) | (1 << 9))).all()      # pair (0, 1): side to side on its lane, the two others on the first two pool lanes
    v = st["qvel"].reshape(n, 3, 6)
    # qvel: world-frame linear velocity, BODY-frame angular velocity
    from scipy.spatial.transform import Rotation as Rot
    Ra = Rot.from_quat(np.roll(_quat_z_to(aA), -1)).as_matrix(); Rb = Rot.from_quat(np.roll(_quat_z_to(aB), -1)).as_matrix()
    want = np.concatenate([h * a[:3], Ra.T @ (h * a[3:6]), h * a[6:9], Rb.T @ (h * a[9:])])
    got = np.concatenate([v[0, 0], v[0, 1]])
    assert np.abs(got - want).max() < 1e-8, (got, want)
    assert abs(want[8] - 6.435e-3) < 1e-5 and abs(want[4]) > 5e-3 and env.stats()["dropped_contacts"] == 0   # translation and rotation
    env.close()


# ------------------------------------------------------------------ heightfield floor, closed form (no oracle involved): tests/hfield_known.py
@pytest.mark.parametrize("prec,layout", [("f64", "row"), ("f32", "row"), ("f64", "block")])
@pytest.mark.parametrize("case", ["groove", "ridge"])
def test_free_bars_on_a_zigzag_heightfield_rest_at_the_closed_form_height(prec, layout, case):
    """Uneven-ground model (MODEL = 1 kernels) on a synthetic zigzag terrain.  groove: a free bar along a V-groove rests with each cap on both
    walls -- two facets per cap (page 1 of the contact slots), each carrying m g / (4 cos theta): axis (r_cap - p) / cos theta above the groove
    line; the contact set is also what MuJoCo's one-contact-per-prism rule gives.  ridge: a free bar balanced across a ridge rests on the grid
    edge with one contact carrying m g (the documented rule for a bar's cylinder: deepest grid edge; MuJoCo would report two tilted contacts).
    Pins cell size (2 radius / (ncol - 1)), elevation scale, geom offset, facet normals and force sharing with no oracle in the loop."""
    import hfield_known as HK
    from tensegrity_rl_amd import TensegrityVecEnv
    n = 16
    free = dict(gain=0.0, ten_stiffness=[0.0] * 9, ten_damping=[0.0] * 9)
    env = TensegrityVecEnv(n, "tensegrity_env-v0", precision=prec, auto_reset=False, kernel_layout=layout, use_contact_forces=1,
                           contact_force_range=(-1e4, 1e4), terminate_when_unhealthy=0, config_overrides=free,
                           xml_file="3prism_jonathan_steady_side_uneven_ground.xml", terrain=HK.zigzag_heights())
    assert env.cfg.model_kind == 1 and env.cfg.cap_radius == HK.R_CAP and tuple(env.cfg.hf_size) == (50.0, 50.0, 1.0, 0.1)
    qpos, zr = (HK.groove_case if case == "groove" else HK.ridge_case)(n)
    env.set_state(qpos, np.zeros((n, 18)))
    zero = np.zeros((n, 6), dtype=np.float32)
    for _ in range(70):
        obs, rew, term, trunc, tobs, info = env.step_tensors(zero)
    st = env.get_state()
    q = st["qpos"].reshape(n, 3, 7)
    tol = 2e-7 if prec == "f64" else 3e-6
    assert np.abs(q[:, :, 2] - zr).max() < tol, (np.abs(q[:, :, 2] - zr).max(), zr[0])
    assert np.abs(q[:, :, :2] - qpos.reshape(n, 3, 7)[:, :, :2]).max() < (1e-6 if prec == "f64" else 1e-4)     # nothing pushes sideways
    assert np.abs(st["qvel"]).max() < (1e-6 if prec == "f64" else 2e-3)
    cc = -info.cpu().numpy()[:, 1]
    want = 5e-4 * 3 * (HK.M_BAR * HK.G) ** 2                      # cfrc_ext of a resting bar: (no torque, m g up), as on the flat floor
    assert np.abs(cc - want).max() < (1e-6 if prec == "f64" else 2e-3) * want, (cc.min(), cc.max(), want)
    assert env.stats()["dropped_contacts"] == 0
    env.close()


# ------------------------------------------------------------------ smooth dynamics against an independent numpy restatement: tests/smooth_numpy.py
@pytest.mark.parametrize("model,prec,layout", [("flat", "f64", "row"), ("flat", "f32", "row"), ("flat", "f64", "block"),
                                               ("uneven", "f64", "row"), ("uneven", "f32", "row"), ("uneven", "f64", "block")])
def test_an_airborne_control_period_matches_the_numpy_restatement(oracle, model, prec, layout):
    """One control period (20 sub-steps) of robots in the air -- no contacts, so the sub-step is tendon geometry, spring / damper / actuator
    forces (affine bias + force clamp on the flat model, filter activations on the uneven one), gyroscopic bias, implicitfast and the
    quaternion update -- against tests/smooth_numpy.py, which shares no code with oracle/ or the kernels (only the model constants, which
    tests/test_model_constants.py pins against the reference XMLs).  Actions beyond ctrlrange on purpose (the clip is part of the path)."""
    import smooth_numpy as SN
    from tensegrity_rl_amd import TensegrityVecEnv
    O = oracle
    m = O.default_model() if model == "flat" else O.uneven_model(None)
    c = SN.Consts(m)
    n = 16
    kw = dict(xml_file="3prism_jonathan_steady_side_uneven_ground.xml", terrain=np.zeros((100, 100), dtype=np.float32)) if model == "uneven" else {}
    env = TensegrityVecEnv(n, "tensegrity_env-v0", precision=prec, auto_reset=False, kernel_layout=layout, terminate_when_unhealthy=0, **kw)
    rng = np.random.default_rng(11)
    roll = O.rolling_qpos()
    qpos = np.zeros((n, 21)); qvel = np.zeros((n, 18)); act = np.zeros((n, 6))
    for e in range(n):
        q = (np.array(m.qpos0[:]) if e % 2 == 0 else roll[e % 6].copy()).reshape(3, 7)
        q[:, 2] += 5.0
        q[:, 3:7] += rng.normal(0, 0.02, (3, 4)); q[:, 3:7] /= np.linalg.norm(q[:, 3:7], axis=1, keepdims=True)
        qpos[e] = q.reshape(21)
        qvel[e] = np.concatenate([np.concatenate([rng.uniform(-0.5, 0.5, 3), rng.uniform(-2, 2, 3)]) for _ in range(3)])
        if c.filter:
            act[e] = rng.uniform(-0.3, 0.1, 6)
    env.set_state(qpos, qvel)
    if c.filter:
        aux = env.get_state()["aux"]; aux[:, 17:23] = act
        env.set_state(qpos, qvel, aux=aux)
    a = rng.uniform(-0.6, 0.3, (n, 6)).astype(np.float32)
    env.step_tensors(a)
    st = env.get_state()
    worst_q = worst_v = 0.0
    for e in range(n):
        q, v, ac = qpos[e].copy(), qvel[e].copy(), act[e].copy()
        for s in range(20):
            q, v, ac, _ = SN.substep(c, q, v, a[e].astype(np.float64), ac)
        g = st["qpos"][e].copy()
        for b in range(3):
            if np.dot(g[7 * b + 3:7 * b + 7], q[7 * b + 3:7 * b + 7]) < 0:
                g[7 * b + 3:7 * b + 7] *= -1
        worst_q = max(worst_q, np.abs(g - q).max())
        worst_v = max(worst_v, np.abs(st["qvel"][e] - v).max() / max(1.0, np.abs(v).max()))
        if c.filter:
            assert np.abs(st["aux"][e, 17:23] - ac).max() < (1e-12 if prec == "f64" else 1e-6)
    assert worst_q < (1e-10 if prec == "f64" else 2e-6), worst_q
    assert worst_v < (1e-8 if prec == "f64" else 1e-4), worst_v
    env.close()


# ------------------------------------------------------------------ kinematics and observation layout against MuJoCo-written observations
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_hip_observations_reproduce_the_mujoco_recorded_ones(oracle, golden_dir, prec):
    """tests/golden/mujoco_last_obs.json: observations real MuJoCo wrote (SB3 checkpoints of the reference, full precision).  The pose behind
    each record is recovered on the oracle's kinematics (tests/test_oracle_anchors.py: machine-precision fits), handed to an fp64 handle with
    everything that moves a bar switched off (no gravity, tendons, actuators; one sub-step per step), and the observation the HIP path then
    returns must be the recorded one in every entry that is kinematics: tensegrity_env's three geom quaternions (scipy's sign rule, the geom
    frame's half turn) and nine tendon lengths; tr_env's six cap positions about their centroid and nine tendon lengths.  Uneven-XML table
    for the records that belong to it, the flat XML's layout with the sites at 0.05 m for the six `models_aiming` checkpoints of that
    intermediate model version."""
    import json
    import os
    import test_oracle_anchors as A
    from tensegrity_rl_amd import TensegrityVecEnv
    O = oracle
    recs = json.load(open(os.path.join(golden_dir, "mujoco_last_obs.json")))["records"]
    mu = O.uneven_model(None)
    flat05 = A._with_site_offset(O.default_model(), 0.05)
    free = dict(gain=0.0, bias=[0.0, 0.0, 0.0], ten_stiffness=[0.0] * 9, ten_damping=[0.0] * 9, gravity=0.0, frame_skip=1)
    jobs = {}   # (env id, table) -> list of (qpos, record observation)
    for r, key in [(r, key) for r in recs for key in ("_last_original_obs", "_last_obs")]:   # (step t and step t + 1 of one rollout)
        o = np.array(r[key]); dim = r["obs_dim"]
        if dim == 39:
            quats = [A._body_quat_from_obs(o[4 * b:4 * b + 4]) for b in range(3)]
            f = A._fit_relative_positions(O, mu, quats, o[30:39], starts=6)
            assert np.sqrt(np.mean(f.fun ** 2)) < 1e-12
            q = np.zeros(21)
            for b in range(3):
                q[7 * b + 3:7 * b + 7] = quats[b]
            q[0:3] = (0, 0, 5.0); q[7:10] = f.x[:3] + (0, 0, 5.0); q[14:17] = f.x[3:] + (0, 0, 5.0)
            jobs.setdefault(("tensegrity_env-v0", "uneven", 39), []).append((q, o))
            continue
        caps = o[:18].reshape(6, 3); tend = o[18:27] if dim == 27 else o[36:45]
        if max(abs(np.linalg.norm(caps[2 * b] - caps[2 * b + 1]) - 1.376) for b in range(3)) > 1e-9:
            continue                                           # the noisy record
        a, q = A._fit_spins(O, mu, caps, tend, want_qpos=True)
        table = "uneven"
        if a > 1e-12:
            a, q = A._fit_spins(O, flat05, caps, tend, want_qpos=True); table = "flat05"
        assert a < 1e-12
        jobs.setdefault(("tr_env-v0", table, 27), []).append((q, np.concatenate([o[:18], tend])))
    assert sorted((k[1], len(v)) for k, v in jobs.items()) == [("flat05", 12), ("uneven", 22), ("uneven", 34)]
    for (env_id, table, dim), items in jobs.items():
        n = len(items)
        kw = dict(xml_file="3prism_jonathan_steady_side_uneven_ground.xml", terrain=np.zeros((100, 100), dtype=np.float32), config_overrides=free) \
            if table == "uneven" else dict(config_overrides=dict(free, site_offset=0.05))
        if env_id == "tr_env-v0":
            kw.update(use_cap_velocity=0)                       # the 27-dim observation: caps + tendon lengths
        env = TensegrityVecEnv(n, env_id, precision=prec, auto_reset=False, terminate_when_unhealthy=0, **kw)
        assert env.obs_dim == dim
        tol = 1e-12 if prec == "f64" else 2e-6                  # (fp32 handle: fp64 pose and kinematics, the observation itself is fp32)
        env.set_state(np.array([q for q, _ in items]), np.zeros((n, 18)))
        obs = env.step_tensors(np.zeros((n, 6), dtype=np.float32))[0].cpu().numpy()
        want = np.array([o for _, o in items])
        if dim == 39:
            assert np.abs(obs[:, :12] - want[:, :12]).max() < tol and np.abs(obs[:, 30:] - want[:, 30:]).max() < tol
        else:
            assert np.abs(obs - want).max() < tol, np.abs(obs - want).max()
        env.close()
