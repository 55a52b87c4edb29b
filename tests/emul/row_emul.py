"""ctypes driver of the host emulation of the row kernels (tests/emul/row_emul.cpp).  TEST INFRASTRUCTURE."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "librow_emul.so")
SRCS = [os.path.join(HERE, "row_emul.cpp"), os.path.join(ROOT, "tensegrity_rl_amd", "csrc", "tgr_row.h"),
        os.path.join(ROOT, "tensegrity_rl_amd", "csrc", "tgr_spmd.h")]
_lib = None


def build(force=False):
    if force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in SRCS):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-shared", "-fPIC", "-o", SO, SRCS[0]])
    return SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def params_from_oracle_model(m, tol, max_iter, ls_iter=50, ls_tol=0.01, cf=(-1.0, 1.0), rtol=0.0):
    """RParams of tgr_row.h from the oracle's tgo_model, the way fill_args (tgr_sim.hip) derives them from tgr_config"""
    dmax = m.solimp[1]
    fr = list(m.friction)
    p = [m.timestep, m.gravity, m.mass, m.inertia[0], m.inertia[2], m.cap_radius, m.bar_half, m.rcyl_radius, m.bcyl_radius, m.bcyl_lo, m.bcyl_hi,
         m.site_off, m.site_z] + list(m.ten_stiffness) + list(m.ten_damping) + list(m.ten_springlen) + \
        [m.gain, m.bias0, m.bias1, m.bias2, m.force_lo, m.force_hi, m.ctrl_lo, m.ctrl_hi, -m.solref[0] / dmax ** 2, -m.solref[1] / dmax,
         m.solimp[0], m.solimp[1], m.solimp[2], m.solimp[3], fr[0], fr[1] ** 2, fr[2] ** 2, fr[3] ** 2, m.impratio, m.invweight_tran,
         tol, rtol, ls_tol, cf[0], cf[1], max_iter, ls_iter, m.actuator_deriv_skip_clamped, 1.0 / max(m.act_tau, 1e-12),
         m.hf_size[0], m.hf_size[1], m.hf_size[2], m.hf_pos[0], m.hf_pos[1], m.hf_pos[2]]
    return np.ascontiguousarray(p, dtype=np.float64)


def step(params, qpos, qvel, ctrl, act=None, nsub=20, f64=True, model=0, heights=None, precision=None, trace=None, capd=None):
    """nsub sub-steps for n envs; returns (qpos, qvel, act, out[n, 32]) -- out: tenL[9], lag x[9], cfrc_absmax, cfrc_clip_sq, bar_contact,
    overflow, n_unconv, n_it, n_ls, n_own.  trace: optional uint32 [n, 24] array that receives the sub-step trace (tgr_row.h RMem::dbg); capd: optional per-env cap radius minus the
    model's (cap-size noise)"""
    n = qpos.shape[0]
    qpos = np.ascontiguousarray(qpos, dtype=np.float64).copy(); qvel = np.ascontiguousarray(qvel, dtype=np.float64).copy()
    ctrl = np.ascontiguousarray(ctrl, dtype=np.float64)
    act = np.zeros((n, 6)) if act is None else np.ascontiguousarray(act, dtype=np.float64).copy()
    out = np.zeros((n, 32))
    hf = None if heights is None else np.ascontiguousarray(heights, dtype=np.float32)
    P = C.POINTER(C.c_double)
    bad = lib().tgr_row_emul(int(f64) if precision is None else {'f32': 0, 'f64': 1, 'mixed': 2}[precision], int(model), params.ctypes.data_as(P), None if hf is None else hf.ctypes.data_as(C.c_void_p),
                             0 if hf is None else hf.shape[0], 0 if hf is None else hf.shape[1], n, nsub, qpos.ctypes.data_as(P),
                             qvel.ctypes.data_as(P), ctrl.ctypes.data_as(P), act.ctypes.data_as(P), out.ctypes.data_as(P),
                             None if trace is None else trace.ctypes.data_as(C.c_void_p),
                             None if capd is None else np.ascontiguousarray(capd, dtype=np.float64).ctypes.data_as(P))
    if bad != 0:
        raise RuntimeError(f"row emulation: {bad} replicated values differed across the lanes of a row (or unsupported model)")
    return qpos, qvel, act, out
