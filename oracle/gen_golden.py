"""Generate tests/golden/*.npz by running the REFERENCE env classes themselves.

Runs only in the build container (needs /root/reference); the GPU box and the
tests only ever see the resulting small fixtures.  TEST INFRASTRUCTURE.

How: `gym`, `gym.envs.mujoco`, `gym.spaces`, `gym.envs.registration` and `mujoco`
are not installed here (ordinary ModuleNotFoundError), so tiny stub modules are
put in sys.modules and the reference modules
    tensegrity_env/tensegrity_env/envs/tensegrity_env.py
    tr_env/tr_env/envs/tr_env.py
are imported unmodified.  The stub MujocoEnv exposes the attribute surface those
classes touch (data.qpos/qvel/ctrl/ten_length/cfrc_ext, data.geom(n).xpos/.xmat,
data.body(n).xpos, set_state, do_simulation, state_vector, get_body_com, dt, ...)
and is fed by the CPU physics restatement (oracle/tgr_physics.c).  The reference's
own step()/_get_obs()/reset_model() code therefore produces the golden
(obs, reward, terminated, info) and reset-pose vectors that pin
  * oracle/tgr_env.c (the C restatement of the Python layer), and
  * the HIP path's epilogue,
for identical physics inputs.  Physics itself stays "parity unpinned" (no MuJoCo).

Unseeded randomness of the reference (np.random.uniform / randint, fresh
default_rng()) is made reproducible by seeding the global RNG and patching
numpy.random.default_rng; the draws are replayed and stored with the fixture.
"""
import ctypes as C
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path = [p for p in sys.path if os.path.abspath(p or '.') != HERE]
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

REF = "/root/reference"
_real_default_rng = np.random.default_rng


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, shape, dtype


class _Named:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class FakeData:
    """numpy views over the oracle's tgo_data with MuJoCo-style accessors."""

    def __init__(self, d):
        self._d = d
        self.qpos = O.arr(d.qpos)
        self.qvel = O.arr(d.qvel)
        self.ctrl = O.arr(d.ctrl)
        self.ten_length = O.arr(d.ten_length)
        self.cfrc_ext = O.arr(d.cfrc_ext)
        self._xpos = O.arr(d.xpos)
        self._xmat = O.arr(d.xmat)
        self._cap = O.arr(d.cap_xpos)

    def geom(self, name):
        if name in ("r01", "r23", "r45"):
            b = ("r01", "r23", "r45").index(name)
            M = self._xmat[b].reshape(3, 3) @ np.diag([-1.0, -1.0, 1.0])  # geom quat "0 0 0 1" (xml:72)
            return _Named(xpos=self._xpos[b], xmat=M.reshape(9))
        k = int(name[1])
        return _Named(xpos=self._cap[k], xmat=self._xmat[k // 2].reshape(9))

    def body(self, name):
        b = ("r01_body", "r23_body", "r45_body").index(name)
        return _Named(xpos=self._xpos[b])


def terrain_heights():
    """normalised elevations of tests/golden/terrain_100.npy (grey level, (h - min) / (max - min), rows flipped: top row = +y)"""
    img = np.load(os.path.join(ROOT, "tests", "golden", "terrain_100.npy")).astype(np.float64)
    return np.ascontiguousarray(((img - img.min()) / (img.max() - img.min()))[::-1], dtype=np.float32)


class MujocoEnv:
    def __init__(self, model_path, frame_skip, observation_space=None, render_mode=None, **kw):
        self._L = O.lib()
        if "uneven" in os.path.basename(str(model_path)):   # run.py --env_xml 3prism_jonathan_steady_side_uneven_ground.xml
            self._m = O.uneven_model(terrain_heights())
        else:
            self._m = O.default_model()
        self._dd = O.new_data(self._m)
        self._L.tgo_forward(C.byref(self._m), C.byref(self._dd))
        self.model = _Named(nq=O.NQ, nv=O.NV, opt=_Named(timestep=self._m.timestep))
        self.data = FakeData(self._dd)
        self.init_qpos = np.array(self._m.qpos0)
        self.init_qvel = np.zeros(O.NV)
        self.frame_skip = frame_skip
        self.observation_space = observation_space
        self.action_space = Box(-0.45, 0.15, (6,), np.float32)
        self.render_mode = render_mode
        self.np_random = _real_default_rng(0)

    @property
    def dt(self):
        return self.model.opt.timestep * self.frame_skip

    def set_state(self, qpos, qvel):
        assert qpos.shape == (O.NQ,) and qvel.shape == (O.NV,)
        self.data.qpos[:] = qpos
        self.data.qvel[:] = qvel
        self._L.tgo_forward(C.byref(self._m), C.byref(self._dd))

    def do_simulation(self, ctrl, n_frames):
        if np.array(ctrl).shape != (6,):
            raise ValueError("Action dimension mismatch")
        self.data.ctrl[:] = ctrl
        self._L.tgo_step(C.byref(self._m), C.byref(self._dd), n_frames)

    def state_vector(self):
        return np.concatenate([self.data.qpos, self.data.qvel])

    def get_body_com(self, name):
        return self.data.body(name).xpos

    def reset(self, seed=None, options=None):
        self._L.tgo_reset_data(C.byref(self._m), C.byref(self._dd))
        return self.reset_model(), {}

    def render(self):
        pass


def install_stubs():
    gym = types.ModuleType("gym")
    utils = types.ModuleType("gym.utils")

    class EzPickle:
        def __init__(self, *a, **k):
            pass

    utils.EzPickle = EzPickle
    gym.utils = utils
    envs = types.ModuleType("gym.envs")
    mj = types.ModuleType("gym.envs.mujoco")
    mj.MujocoEnv = MujocoEnv
    reg = types.ModuleType("gym.envs.registration")
    reg.register = lambda **k: None
    spaces = types.ModuleType("gym.spaces")
    spaces.Box = Box
    envs.mujoco = mj
    envs.registration = reg
    gym.envs = envs
    gym.spaces = spaces
    for name, mod in [("gym", gym), ("gym.utils", utils), ("gym.envs", envs), ("gym.envs.mujoco", mj),
                      ("gym.envs.registration", reg), ("gym.spaces", spaces), ("mujoco", types.ModuleType("mujoco"))]:
        sys.modules[name] = mod
    sys.path.insert(0, os.path.join(REF, "tensegrity_env"))
    sys.path.insert(0, os.path.join(REF, "tr_env"))


INFO_SCALARS = ["reward_forward", "reward_ctrl", "reward_survive", "x_position", "y_position", "psi",
                "distance_from_origin", "x_velocity", "y_velocity", "forward_reward"]


def replay_draws(kind, seed, mean, std, lo, hi, action):
    """Re-derive what the reference's unseeded RNG calls return under our seeding."""
    st = np.random.get_state()
    np.random.seed(seed)
    pose = int(np.random.randint(0, 6)) if kind == "tr_env" else 0
    theta = float(np.random.uniform(low=0.0, high=2 * np.pi))
    u_way = [0.5, 0.5]
    if kind == "tr_env" and action in ("tracking", "aiming"):
        u_way = [float(np.random.uniform(0, 1)), float(np.random.uniform(0, 1))]
    np.random.set_state(st)
    t = _real_default_rng(seed).standard_normal(size=6) * std + mean
    t = np.clip(t, lo, hi)
    return pose, theta, t, u_way


def run_case(kind, action, seed, steps, direction=1, model="flat", **kwargs):
    if kind == "tensegrity_env":
        from tensegrity_env.envs.tensegrity_env import tensegrity_env as cls
        mean, std, lo, hi = -0.15, 0.1, -0.45, -0.15
    else:
        from tr_env.envs.tr_env import tr_env as cls
        mean, std, lo, hi = 0.15, 0.2, -0.45, 0.15
    np.random.default_rng = lambda *a, **k: _real_default_rng(seed) if not a and not k else _real_default_rng(*a, **k)
    xml = "3prism_jonathan_steady_side_uneven_ground.xml" if model == "uneven" else "unused.xml"
    env = cls(xml_file=xml, desired_action=action, desired_direction=direction, **kwargs)
    out = {"kind": kind, "action": action, "direction": direction, "model_kind": int(model == "uneven")}
    ep_resets = []
    rec = {k: [] for k in ["obs", "reward", "terminated", "qpos", "qvel", "actions", "ctrl", "act", "ring", "ring_len", "step_num"] + INFO_SCALARS}
    noise = float(kwargs.get("reset_noise_scale", 0.0))

    def ring_of(env):  # self._heading_buffer (a deque that is never cleared), oldest first, padded to the 32 slots of the C-ABI
        r = np.zeros(32); b = list(env._heading_buffer); r[:len(b)] = b
        return r, len(b)
    arng = _real_default_rng(1000 + seed)
    n_resets = 2
    for ep in range(n_resets):
        s = seed + 17 * ep
        np.random.default_rng = (lambda ss: (lambda *a, **k: _real_default_rng(ss) if not a and not k else _real_default_rng(*a, **k)))(s)
        np.random.seed(s)
        env.np_random = _real_default_rng(s + 5)  # what reset_noise_scale draws from (tensegrity_env.py:439-445)
        obs0, _ = env.reset()
        pose, theta, tendons, u_way = replay_draws(kind, s, mean, std, lo, hi, action)
        gq = _real_default_rng(s + 5)
        dq = gq.uniform(low=-noise, high=noise, size=21); dv = noise * gq.standard_normal(18)
        ring0, ring_len0 = ring_of(env)
        ep_resets.append(dict(pose=pose, theta=theta, tendons=tendons, u_way=u_way, obs=np.array(obs0, dtype=np.float64),
                              qpos=env.data.qpos.copy(), qvel=env.data.qvel.copy(), reset_psi=float(env._reset_psi), ctrl=env.data.ctrl.copy(),
                              act=O.arr(env._dd.act).copy(), dq=dq, dv=dv, ring=ring0, ring_len=ring_len0,
                              step_num=int(getattr(env, "_step_num", 0)),
                              xy_lag=np.array([np.mean([env.data.body(b).xpos[i] for b in ('r01_body', 'r23_body', 'r45_body')]) for i in range(2)]),
                              waypt=np.array(getattr(env, "_waypt", np.zeros(2)), dtype=np.float64).reshape(-1)[:2]
                              if np.size(getattr(env, "_waypt", np.zeros(0))) == 2 else np.zeros(2),
                              oripoint=np.array(getattr(env, "_oripoint", np.zeros(2)), dtype=np.float64)))
        for t in range(steps):
            a = arng.uniform(-0.45, 0.15, size=6).astype(np.float32)
            if t % 7 == 3:
                a = (a * 1.5).astype(np.float32)  # exercise ctrl clamping (values outside ctrlrange)
            obs, r, term, trunc, info = env.step(a)
            rec["obs"].append(np.array(obs, dtype=np.float64)); rec["reward"].append(float(r)); rec["terminated"].append(bool(term))
            rg, rl = ring_of(env)
            rec["ring"].append(rg); rec["ring_len"].append(rl); rec["step_num"].append(int(getattr(env, "_step_num", 0)))
            rec["qpos"].append(env.data.qpos.copy()); rec["qvel"].append(env.data.qvel.copy()); rec["actions"].append(a); rec["ctrl"].append(env.data.ctrl.copy()); rec["act"].append(O.arr(env._dd.act).copy())
            for k in INFO_SCALARS:
                rec[k].append(float(info[k]))
    for k, v in rec.items():
        out[k] = np.array(v)
    for k in ep_resets[0]:
        out["reset_" + k] = np.array([e[k] for e in ep_resets])
    out["steps_per_episode"] = steps
    return out


CASES = [
    ("tensegrity_env", "straight", 11, 12, 1, {}),
    ("tensegrity_env", "straight", 12, 8, -1, {"terminate_when_unhealthy": False}),
    ("tensegrity_env", "turn", 13, 30, 1, {}),
    ("tensegrity_env", "straight", 14, 6, 1, {"use_contact_forces": True}),
    ("tr_env", "straight", 21, 12, 1, {}),
    ("tr_env", "turn", 22, 8, -1, {}),
    ("tr_env", "straight", 23, 6, 1, {"use_cap_velocity": False}),
    ("tr_env", "tracking", 24, 8, 1, {}),
    ("tr_env", "aiming", 25, 8, 1, {}),
    ("tr_env", "vel_track", 26, 8, 1, {}),
    ("tensegrity_env", "straight", 31, 6, 1, {}, "uneven"),
    ("tr_env", "tracking", 32, 6, 1, {}, "uneven"),
    ("tensegrity_env", "straight", 41, 6, 1, {"reset_noise_scale": 0.05}),
    ("tr_env", "straight", 42, 6, 1, {"reset_noise_scale": 0.05}),
]


def helper_vectors():
    """pure helper known-answers (tr_env.py:648-683) evaluated by the reference class itself"""
    from tr_env.envs.tr_env import tr_env as cls
    env = cls(xml_file="unused.xml", desired_action="tracking")
    env._waypt = np.array([3.0, 0.0]); env._oripoint = np.array([0.0, 0.0])
    xs = np.array([[1.0, 0.1], [2.5, -0.2], [0.0, 0.0], [3.0, 0.0]])
    ditch = np.array([env._ditch_reward(x) for x in xs])
    ang_in = np.array([4.0, -4.0, 0.5, np.pi, -np.pi, 7.0])
    ang = np.array([env._angle_normalize(a) for a in ang_in])
    vt = np.array([env._vel_track_rew(np.array([.5, 0, 0]), np.array([.4, .1, .05]))])
    filt = env._action_filter(np.full(6, 0.1), np.zeros(6))
    return dict(ditch_xy=xs, ditch=ditch, ang_in=ang_in, ang=ang, vel_track=vt, filt=filt)


def noise_vectors():
    """tr_env.py:549-646 with use_obs_noise=True: the reference's own (clean, noisy) observation pairs together with the
    standard-normal draws it consumed (its unseeded default_rng() is patched to a seeded one and replayed here)."""
    from tr_env.envs.tr_env import tr_env as cls
    clean, noisy, zs, acts, capv = [], [], [], [], []
    for ci, (action, kw) in enumerate([("straight", {}), ("straight", {"use_cap_velocity": False}), ("tracking", {}),
                                       ("aiming", {}), ("vel_track", {})]):
        ss = 900 + ci
        np.random.default_rng = (lambda q: (lambda *a, **k: _real_default_rng(q) if not a and not k else _real_default_rng(*a, **k)))(ss)
        np.random.seed(ss)
        env = cls(xml_file="unused.xml", desired_action=action, use_obs_noise=True, **kw)
        env.reset()
        arng = _real_default_rng(ss)
        for t in range(3):
            obs, r, term, trunc, info = env.step(arng.uniform(-0.45, 0.15, size=6).astype(np.float32))
            g = _real_default_rng(ss)
            z = np.zeros(45)
            z[:18] = np.concatenate([g.standard_normal(size=3) for _ in range(6)])
            z[18:27] = g.standard_normal(size=9)
            if kw.get("use_cap_velocity", True):
                z[27:45] = np.concatenate([g.standard_normal(size=3) for _ in range(6)])
            c = np.zeros(48); n = np.zeros(48)
            c[:len(obs)] = info["real_observation"]; n[:len(obs)] = obs
            clean.append(c); noisy.append(n); zs.append(z); acts.append(action); capv.append(int(kw.get("use_cap_velocity", True)))
    return dict(clean=np.array(clean), noisy=np.array(noisy), z=np.array(zs), action=np.array(acts, dtype="U16"), use_cap_velocity=np.array(capv),
                tendon_stdev=0.02, cap_pos_stdev=0.05)


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not present; fixtures can only be regenerated in the build container")
    install_stubs()
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    for case in CASES:
        kind, action, seed, steps, direction, kw = case[:6]
        model = case[6] if len(case) > 6 else "flat"
        res = run_case(kind, action, seed, steps, direction, model=model, **kw)
        tag = "_".join(f"{k}-{int(v) if isinstance(v, bool) else v}" for k, v in kw.items())
        name = ("uneven_" if model == "uneven" else "") + f"{kind}_{action}_d{direction}" + (f"_{tag}" if tag else "") + ".npz"
        res["kwargs_keys"] = np.array(list(kw.keys()), dtype="U64")
        res["kwargs_vals"] = np.array([float(v) for v in kw.values()])
        np.savez_compressed(os.path.join(outdir, name), **res)
        print(name, "obs", res["obs"].shape, "reward[:3]", res["reward"][:3], "term any", res["terminated"].any())
    np.savez_compressed(os.path.join(outdir, "tr_env_helpers.npz"), **helper_vectors())
    np.savez_compressed(os.path.join(outdir, "tr_env_obs_noise.npz"), **noise_vectors())
    np.random.default_rng = _real_default_rng


if __name__ == "__main__":
    main()
