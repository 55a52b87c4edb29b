"""A third, independent restatement of ONE piece of the path: a contact-free sub-step of the 3-bar model in plain numpy, written from
MuJoCo's documented pipeline and the model documents (flat xml:28-39,125-165; uneven xml:29-41) without looking at oracle/tgr_physics.c or
the kernels -- so that the smooth dynamics (tendon geometry and Jacobians, spring / damper / affine-actuator / filter-actuator forces, force
clamp, gyroscopic bias, implicitfast, quaternion integration) of oracle AND HIP path are checked against something neither shares code with.
Constants come from the oracle's model table, which tests/test_model_constants.py pins number by number against the reference XMLs.

Conventions (MuJoCo free joint): qpos = (x world, quaternion w x y z), qvel = (linear velocity world, angular velocity BODY frame);
M = diag(m, m, m, Ixx, Iyy, Izz) per bar (body frame = principal frame, centre of mass at the origin)."""
import numpy as np


class Consts:
    def __init__(self, m):
        """m: oracle Model struct (ctypes)"""
        self.h = m.timestep; self.g = m.gravity; self.mass = m.mass; self.inertia = np.array(m.inertia[:])
        self.site_local = np.array([[m.site_local[i][k] for k in range(3)] for i in range(6)])
        self.ten_ends = np.array([[m.ten_ends[t][k] for k in range(4)] for t in range(9)], dtype=int)   # bar_a, site_a, bar_b, site_b
        self.k = np.array(m.ten_stiffness[:]); self.b = np.array(m.ten_damping[:]); self.L0 = np.array(m.ten_springlen[:])
        self.gain = m.gain; self.bias = (m.bias0, m.bias1, m.bias2); self.frange = (m.force_lo, m.force_hi); self.crange = (m.ctrl_lo, m.ctrl_hi)
        self.filter = bool(m.act_filter); self.tau = m.act_tau
        self.keep_clamped_deriv = not m.actuator_deriv_skip_clamped


def rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def qmul(a, b):
    aw, ax, ay, az = a; bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])


def tendons(c, qpos):
    """lengths [9] and Jacobian [9, 18] (dL / dqvel)"""
    X = qpos.reshape(3, 7)
    R = [rot(X[b, 3:7] / np.linalg.norm(X[b, 3:7])) for b in range(3)]
    L = np.zeros(9); J = np.zeros((9, 18))
    for t, (ba, sa, bb, sb) in enumerate(c.ten_ends):
        ra, rb = R[ba] @ c.site_local[sa], R[bb] @ c.site_local[sb]
        d = (X[bb, :3] + rb) - (X[ba, :3] + ra)
        L[t] = np.linalg.norm(d); u = d / L[t]
        # point velocity = v + R (w_body x s_local): dL/dt = u . (pdot_b - pdot_a)
        J[t, 6 * bb:6 * bb + 3] += u; J[t, 6 * bb + 3:6 * bb + 6] += np.cross(c.site_local[sb], R[bb].T @ u)
        J[t, 6 * ba:6 * ba + 3] -= u; J[t, 6 * ba + 3:6 * ba + 6] -= np.cross(c.site_local[sa], R[ba].T @ u)
    return L, J


def substep(c, qpos, qvel, ctrl, act):
    """one mj_step without contacts: returns (qpos, qvel, act, tendon force [9] incl. actuators)"""
    h = c.h
    L, J = tendons(c, qpos)
    Ld = J @ qvel
    f_t = -c.k * (L - c.L0) - c.b * Ld                        # passive spring-dampers (a tendon without stiffness / damping has k = b = 0)
    dfdv = -c.b.copy()                                        # d f_t / d Ldot
    u = np.clip(ctrl, *c.crange)
    if c.filter:                                              # dyntype=filter, gaintype=fixed, biastype=none: force = gain * act
        f_a = c.gain * act
        act_new = act + h * (u - act) / c.tau
        d_a = np.zeros(6)
    else:                                                     # dyntype=none, affine bias: gain ctrl + b0 + b1 L + b2 Ldot, clamped to forcerange
        raw = c.gain * u + c.bias[0] + c.bias[1] * L[:6] + c.bias[2] * Ld[:6]
        f_a = np.clip(raw, *c.frange)
        act_new = act
        inside = (raw > c.frange[0]) & (raw < c.frange[1])
        d_a = np.where(inside | c.keep_clamped_deriv, c.bias[2], 0.0)
    f_t[:6] += f_a; dfdv[:6] += d_a
    Mdiag = np.tile(np.concatenate([[c.mass] * 3, c.inertia]), 3)
    bias = np.zeros(18)
    for b in range(3):
        w = qvel[6 * b + 3:6 * b + 6]
        bias[6 * b + 2] = c.mass * c.g                        # -m g_vec, gravity = (0, 0, -g)
        bias[6 * b + 3:6 * b + 6] = np.cross(w, c.inertia * w)
    f = J.T @ f_t - bias
    D = J.T @ (dfdv[:, None] * J)                              # d f / d qvel (implicitfast: no derivative of the gyroscopic term)
    a = np.linalg.solve(np.diag(Mdiag) - h * D, f)
    v = qvel + h * a
    X = qpos.reshape(3, 7).copy()
    for b in range(3):
        X[b, :3] += h * v[6 * b:6 * b + 3]
        w = v[6 * b + 3:6 * b + 6]; n = np.linalg.norm(w)
        if n > 0:
            ang = h * n
            dq = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * w / n])
            q = qmul(X[b, 3:7], dq)
            X[b, 3:7] = q / np.linalg.norm(q)
    return X.reshape(21), v, act_new, f_t
