"""TensegrityVecEnv: N environments behind the Stable-Baselines3 VecEnv surface, stepped by the HIP
kernels through the C-ABI (include/tensegrity.h).

Mirrors what SB3 builds around the reference env (run.py:41-85 -> DummyVecEnv(Monitor(env))):
auto-reset with infos[i]["terminal_observation"], "TimeLimit.truncated" and Monitor's
info["episode"] = {"r","l","t"}.  PyTorch is only plumbing here: device buffers + the current stream.
"""
import ctypes as C
import os
import time

from collections.abc import Mapping

import numpy as np

from . import _native as N
from .spaces import Box

ENV_IDS = {"tensegrity_env-v0": N.ENV_TENSEGRITY, "tr_env-v0": N.ENV_TR, "tensegrity_env": N.ENV_TENSEGRITY, "tr_env": N.ENV_TR}

# constructor kwargs of the reference classes that map 1:1 onto tgr_config fields
_DIRECT = ["use_contact_forces", "use_cap_velocity", "terminate_when_unhealthy", "is_test", "desired_direction",
           "ctrl_cost_weight", "contact_cost_weight", "healthy_reward", "min_reset_heading", "max_reset_heading",
           "tendon_reset_mean", "tendon_reset_stdev", "tendon_max_length", "tendon_min_length", "ditch_reward_max",
           "ditch_reward_stdev", "waypt_reward_amplitude", "waypt_reward_stdev", "yaw_reward_weight", "use_obs_noise",
           "obs_noise_tendon_stdev", "obs_noise_cap_pos_stdev", "reset_noise_scale", "use_cap_size_noise"]
# accepted for signature compatibility, no effect on the built path (reference defaults)
_IGNORED = ["xml_file", "render_mode", "contact_with_self_penalty",
            "threshold_waypt", "width", "height", "camera_id", "camera_name"]


def heights_from_image(img):
    """Elevations of a heightfield image the way mujoco's compiler prepares them [ext]: grey level, normalised to [0, 1] by
    (h - min) / (max - min), rows flipped so that row 0 is the southern edge (y = -radius_y) -- the image's top row is north.
    (Row order and the triangulation are assumptions of this restatement: DESIGN.md, "uneven ground".)"""
    a = np.asarray(img, dtype=np.float64)
    if a.ndim == 3:
        a = a[..., :3].mean(axis=2)
    lo, hi = a.min(), a.max()
    a = (a - lo) / (hi - lo) if hi > lo else np.zeros_like(a)
    return np.ascontiguousarray(a[::-1], dtype=np.float32)


def _terrain_next_to(xml_file):
    """the <hfield file=...> image the model document refers to, when the caller points at a real checkout of the XML"""
    if not isinstance(xml_file, str) or not os.path.isfile(xml_file):
        return None
    try:
        import xml.etree.ElementTree as ET
        from PIL import Image
        node = ET.parse(xml_file).getroot().find("asset/hfield")
        path = os.path.join(os.path.dirname(xml_file), node.get("file"))
        return np.array(Image.open(path))
    except Exception:
        return None


def is_uneven_xml(xml_file):
    return isinstance(xml_file, str) and "uneven" in os.path.basename(xml_file)


def make_config(env_id="tensegrity_env-v0", desired_action="straight", precision="f32", auto_reset=True, seed=1234,
                first_env_id=0, max_episode_steps=5000, **kwargs):
    """tgr_config from the reference's constructor kwargs (tensegrity_env.py:160-181, tr_env.py:137-173)."""
    if env_id not in ENV_IDS:
        raise ValueError(f"unknown env id {env_id!r}")
    kind = ENV_IDS[env_id]
    if desired_action not in N.ACTIONS:
        raise ValueError(f"unknown desired_action {desired_action!r}")
    cfg = N.default_config(kind, desired_action)
    if kwargs.get("use_cap_size_noise") and kind != N.ENV_TR:
        # (the reference's tensegrity_env has no such kwarg -- TypeError there; silently doing nothing would be worse)
        raise TypeError("use_cap_size_noise is a tr_env option (tr_env.py:143,163,685-706)")
    # the reference picks the model with xml_file (run.py --env_xml); `model="uneven"` does the same without a path
    if kwargs.pop("model", "flat") == "uneven" or is_uneven_xml(kwargs.get("xml_file")):
        N.load().tgr_uneven_config(C.byref(cfg))
    for k in list(kwargs):
        if k in _DIRECT:
            setattr(cfg, k, type(getattr(cfg, k))(kwargs.pop(k)))
        elif k in _IGNORED:
            kwargs.pop(k)
    if "contact_force_range" in kwargs:
        lo, hi = kwargs.pop("contact_force_range")
        cfg.contact_force_lo, cfg.contact_force_hi = float(lo), float(hi)
        if desired_action == "turn":  # tensegrity_env.py:224-225
            cfg.contact_force_lo, cfg.contact_force_hi = -1000.0, 1000.0
    for src, dst in (("way_pts_range", "waypt_range"), ("way_pts_angle_range", "waypt_angle_range"), ("cap_size_noise_range", "cap_size_noise_range")):
        if src in kwargs:
            lo, hi = kwargs.pop(src)
            getattr(cfg, dst)[0], getattr(cfg, dst)[1] = float(lo), float(hi)
    if "reward_delay_seconds" in kwargs:
        cfg.reward_delay_steps = int(kwargs.pop("reward_delay_seconds") / (cfg.timestep * cfg.frame_skip))
    for k, v in (kwargs.pop("config_overrides", None) or {}).items():
        # raw tgr_config fields (model constants such as gain / bias / ten_stiffness): for known-answer tests of the physics
        cur = getattr(cfg, k)
        if hasattr(cur, "__len__"):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(cfg, k, type(cur)(v))
    if "kernel_layout" in kwargs:
        cfg.kernel_layout = N.KERNEL_LAYOUT[kwargs.pop("kernel_layout")]
    for k in ("actuator_deriv_skip_clamped", "solver_tolerance", "solver_rel_tolerance", "solver_iterations", "n_donors", "solver_warm_start",
              "reset_stream_envs", "donor_stream"):
        if k in kwargs:
            setattr(cfg, k, type(getattr(cfg, k))(kwargs.pop(k)))
    if kwargs:
        raise TypeError(f"unexpected keyword arguments: {sorted(kwargs)}")
    cfg.precision = N.PRECISION[precision]
    cfg.auto_reset = int(bool(auto_reset))
    cfg.seed = int(seed)
    cfg.first_env_id = int(first_env_id)
    cfg.max_episode_steps = int(max_episode_steps) if max_episode_steps else 2 ** 31 - 1
    return cfg


class _InfoFrame:
    """what one step brought home for info_mode="full": host copies (taken once per step, a few hundred KB) of the info block and of the
    noise-free observations -- immutable from then on, so a mapping handed out for this step keeps its values whatever steps follow"""
    __slots__ = ("info", "real_obs", "tr")

    def __init__(self, info, real_obs, tr):
        self.info, self.real_obs, self.tr = info, real_obs, tr


_SCALAR = {k: j for j, k in enumerate(N.INFO_KEYS)}
_TR_KEYS = N.INFO_KEYS + ["waypt", "oripoint", "tendon_length", "real_observation", "total_bar_contact"]
_TE_KEYS = N.INFO_KEYS + ["total_bar_contact"]


class LazyInfo(Mapping):
    """The reference's info dict of ONE env for ONE step (tensegrity_env.py:383-398, tr_env.py:496-511), materialised key by key on access from
    the step's frame.  A snapshot, not a view: the frame is never written again.  `extra` holds what VecEnv semantics add for an env that
    finished (terminal_observation, TimeLimit.truncated, episode)."""
    __slots__ = ("_f", "_i", "_extra")

    def __init__(self, frame, i, extra=None):
        self._f, self._i, self._extra = frame, i, extra

    def __getitem__(self, k):
        if self._extra is not None and k in self._extra:
            return self._extra[k]
        f, i = self._f, self._i
        j = _SCALAR.get(k)
        if j is not None:
            return np.float64(f.info[i, j])
        if k == "total_bar_contact":  # what run.py:149-155 computes from data.contact
            return float(f.info[i, N.INFO_BAR_CONTACT])
        if f.tr:  # tr_env.py:506-511
            if k == "waypt":
                return f.info[i, 10:12].astype(np.float64)
            if k == "oripoint":
                return f.info[i, 12:14].astype(np.float64)
            if k == "tendon_length":
                return f.info[i, N.INFO_TENDON_LENGTH:N.INFO_TENDON_LENGTH + 9].astype(np.float64)
            if k == "real_observation":
                return f.real_obs[i].astype(np.float64)
        raise KeyError(k)

    def __iter__(self):
        yield from (_TR_KEYS if self._f.tr else _TE_KEYS)
        if self._extra is not None:
            yield from self._extra

    def __len__(self):
        return len(_TR_KEYS if self._f.tr else _TE_KEYS) + (len(self._extra) if self._extra is not None else 0)

    def copy(self):
        return dict(self)

    def __repr__(self):
        return repr(dict(self))


class LazyInfoList(list):
    """infos of one step: a list whose entries come into being when they are looked at (SB3 indexes the finished envs and iterates for
    `episode`; run.py:139-161 reads env 0's keys) -- building 4096 dicts per step cost 4-8x the device step"""
    __slots__ = ("_frame", "_extras")

    def __init__(self, n, frame, extras):
        super().__init__([None] * n)
        self._frame, self._extras = frame, extras

    def _make(self, i):
        # (no frame: info_mode="minimal" -- an empty dict, or what VecEnv semantics add for an env that finished)
        x = LazyInfo(self._frame, i, self._extras.get(i)) if self._frame is not None else self._extras.get(i, None)
        if x is None:
            x = {}
        list.__setitem__(self, i, x)
        return x

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        x = list.__getitem__(self, i)
        return x if x is not None else self._make(i if i >= 0 else len(self) + i)

    def __iter__(self):
        for i in range(len(self)):
            x = list.__getitem__(self, i)
            yield x if x is not None else self._make(i)


class TensegrityVecEnv:
    """SB3 VecEnv duck type.  Not a subclass so that stable_baselines3 is optional; `to_sb3()` wraps it in a real
    VecEnv subclass when SB3 is importable."""

    metadata = {"render_modes": ["human", "rgb_array", "depth_array"], "render_fps": 50}

    def __init__(self, num_envs, env_id="tensegrity_env-v0", device=0, info_mode="minimal", terrain=None, **kwargs):
        """terrain: for the uneven-ground model, the heightfield -- an image array (uint8, as in terrain_image_100.png) or float
        elevations already normalised to [0, 1] with row 0 at y = -radius_y."""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("TensegrityVecEnv needs a HIP device (MI355X); there is no CPU fallback")
        self.torch = torch
        self.lib = N.load()
        self.num_envs = int(num_envs)
        self.env_id = env_id
        self.cfg = make_config(env_id, **kwargs)
        self.device_index = int(device)
        self.device = torch.device(f"cuda:{self.device_index}")
        self.handle = C.c_void_p()
        rc = self.lib.tgr_create(C.byref(self.cfg), self.num_envs, self.device_index, C.byref(self.handle))
        N.check(self.handle if self.handle else None, rc, "tgr_create")
        if self.cfg.model_kind == 1:
            if terrain is None:
                terrain = _terrain_next_to(kwargs.get("xml_file"))
            if terrain is None:
                raise ValueError("the uneven-ground model needs terrain= (the heightfield image or normalised elevations)")
            t_ = np.asarray(terrain)
            hts = heights_from_image(t_) if t_.dtype == np.uint8 or t_.ndim == 3 else np.ascontiguousarray(t_, dtype=np.float32)
            self.heights = hts
            N.check(self.handle, self.lib.tgr_set_hfield(self.handle, hts.ctypes.data, hts.shape[0], hts.shape[1]), "tgr_set_hfield")
        self.obs_dim = self.lib.tgr_obs_dim(self.handle)
        self.f64 = self.cfg.precision == 1
        dt = torch.float64 if self.f64 else torch.float32
        n, D = self.num_envs, self.obs_dim
        with torch.cuda.device(self.device):
            # every output of a step lives in ONE device allocation, so that the numpy-facing step_wait() brings a step home with a single
            # device-to-host copy into a pinned mirror (hard part 6): [obs | terminal_obs | info | reward | terminated | truncated]
            isz = 8 if self.f64 else 4
            # info["real_observation"] (tr_env.py:507): the noise-free observation; only needed (and bound) with use_obs_noise -- part of
            # the packed block, so that it comes home in the same copy
            want_real = bool(self.cfg.use_obs_noise and self.cfg.env_kind == N.ENV_TR)
            sizes = [n * D * isz, n * D * isz, n * N.INFO_K * isz, n * isz, n, n, (-2 * n) % 8, n * D * isz if want_real else 0]
            offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
            self._pack = torch.zeros(int(offs[-1]), dtype=torch.uint8, device=self.device)
            self._pack_h = torch.zeros(int(offs[-1]), dtype=torch.uint8).pin_memory()
            sec = lambda buf, i, dtype, shape: buf[int(offs[i]):int(offs[i + 1])].view(dtype).view(shape)
            self.obs = sec(self._pack, 0, dt, (n, D))
            self.terminal_obs = sec(self._pack, 1, dt, (n, D))
            self.info = sec(self._pack, 2, dt, (n, N.INFO_K))
            self.reward = sec(self._pack, 3, dt, (n,))
            self.terminated = sec(self._pack, 4, torch.uint8, (n,))
            self.truncated = sec(self._pack, 5, torch.uint8, (n,))
            hv = [sec(self._pack_h, i, d_, sh).numpy() for i, (d_, sh) in enumerate([(dt, (n, D)), (dt, (n, D)), (dt, (n, N.INFO_K)), (dt, (n,)),
                                                                                      (torch.uint8, (n,)), (torch.uint8, (n,))])]
            self._h_obs, self._h_tobs, self._h_info, self._h_rew, self._h_term, self._h_trunc = hv
            self._act_h = torch.zeros((n, 6), dtype=torch.float32).pin_memory()
            self._act_d = torch.zeros((n, 6), dtype=torch.float32, device=self.device)
            self.real_obs = sec(self._pack, 7, dt, (n, D)) if want_real else None
            self._h_real = sec(self._pack_h, 7, dt, (n, D)).numpy() if want_real else None
        if self.real_obs is not None:
            N.check(self.handle, self.lib.tgr_bind_real_observation(self.handle, self.real_obs.data_ptr()), "tgr_bind_real_observation")
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(D,), dtype=np.float64)
        self.action_space = Box(low=np.float32(self.cfg.ctrl_range[0]), high=np.float32(self.cfg.ctrl_range[1]), shape=(6,), dtype=np.float32)
        self.render_mode = None
        self.info_mode = info_mode
        self.reset_infos = [{} for _ in range(n)]
        self._actions = None
        self._t0 = time.time()
        self.dt = self.cfg.timestep * self.cfg.frame_skip
        self.frame_skip = self.cfg.frame_skip

    # ------------------------------------------------------------------ device-tensor API (zero host copies)
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def reset_tensors(self, mask=None):
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            if m.shape != (self.num_envs,):
                raise ValueError("mask shape")
        rc = self.lib.tgr_reset(self.handle, None if m is None else m.ctypes.data, self.obs.data_ptr(), self._stream())
        N.check(self.handle, rc, "tgr_reset")
        return self.obs

    def step_tensors(self, actions):
        """actions: float32 cuda tensor [N,6].  Returns views of the internal device buffers
        (obs, reward, terminated, truncated, terminal_obs, info)."""
        t = self.torch
        if not isinstance(actions, t.Tensor):
            actions = t.as_tensor(np.asarray(actions, dtype=np.float32), device=self.device)
        if tuple(actions.shape) != (self.num_envs, 6):
            raise ValueError("Action dimension mismatch")
        if actions.dtype != t.float32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=t.float32).contiguous()
        fn = self.lib.tgr_step_f64 if self.f64 else self.lib.tgr_step
        rc = fn(self.handle, actions.data_ptr(), self.obs.data_ptr(), self.reward.data_ptr(), self.terminated.data_ptr(),
                self.truncated.data_ptr(), self.terminal_obs.data_ptr(), self.info.data_ptr(), self._stream())
        N.check(self.handle, rc, "tgr_step")
        self._keep = actions
        return self.obs, self.reward, self.terminated, self.truncated, self.terminal_obs, self.info

    # ------------------------------------------------------------------ SB3 VecEnv surface (numpy)
    def reset(self):
        obs = self.reset_tensors().cpu().numpy().astype(np.float64)
        self.reset_infos = [{} for _ in range(self.num_envs)]
        return obs

    def step_async(self, actions):
        self._actions = actions

    def step_wait(self):
        """SB3's VecEnv.step_wait: numpy in, numpy out.  One pinned host-to-device copy of the actions, the step launches, ONE
        device-to-host copy of the packed outputs, one stream synchronisation; terminal observations and episode statistics are
        read from the pinned mirror for finished envs only."""
        t = self.torch
        a = self._actions
        if isinstance(a, t.Tensor):
            self.step_tensors(a)
        else:
            a = np.asarray(a, dtype=np.float32)
            if a.shape != (self.num_envs, 6):
                raise ValueError("Action dimension mismatch")
            np.copyto(self._act_h.numpy(), a)
            self._act_d.copy_(self._act_h, non_blocking=True)
            self.step_tensors(self._act_d)
        self._pack_h.copy_(self._pack, non_blocking=True)
        t.cuda.current_stream(self.device).synchronize()
        obs_h = self._h_obs.astype(np.float64)
        rew_h = self._h_rew.astype(np.float32)
        term_h = self._h_term.astype(bool)
        trunc_h = self._h_trunc.astype(bool)
        dones = term_h | trunc_h
        infos = self._infos(dones, term_h, trunc_h, obs_h)
        return obs_h, rew_h, dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def _infos(self, dones, term_h, trunc_h, obs_h):
        """info_mode="minimal": empty dicts except for finished envs (all SB3 itself reads); "full": the reference's info dict for every env,
        as lazy snapshot mappings (LazyInfo).  Reads the pinned host mirror of the step that has just been brought home."""
        n = self.num_envs
        ih = self._h_info
        full = self.info_mode == "full"
        if full:
            # one frame per step: copies of the info block and of the noise-free observations (tr_env.py:507: of this step -- for an env that
            # just finished, of its last step); the per-env mappings are made when somebody looks at them (LazyInfoList)
            tr = self.cfg.env_kind == N.ENV_TR
            rh = None
            if tr:
                if self._h_real is not None:
                    rh = self._h_real.copy()
                else:
                    rh = self._h_obs.copy()
                    fin = np.nonzero(dones)[0]
                    if fin.size and self.cfg.auto_reset:
                        rh[fin] = self._h_tobs[fin]
            frame = _InfoFrame(ih.copy(), rh, tr)
        else:
            frame = None
        extras = {}
        infos = LazyInfoList(n, frame, extras)
        idx = np.nonzero(dones)[0]
        if idx.size:
            th = self._h_tobs[idx].astype(np.float64)  # (a copy: the mirror is overwritten by the next step)
            ep = ih[idx, 14:16]
            now = round(time.time() - self._t0, 6)
            for j, i in enumerate(idx):
                d = extras.setdefault(int(i), {})
                d["terminal_observation"] = th[j]
                d["TimeLimit.truncated"] = bool(trunc_h[i] and not term_h[i])
                d["episode"] = {"r": float(ep[j, 0]), "l": int(ep[j, 1]), "t": now}
        return infos

    def close(self):
        if getattr(self, "handle", None) and self.handle:
            self.lib.tgr_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def seed(self, seed=None, reset=False):
        """re-key the reset RNG (SB3's VecEnv.seed).  Every lane's episode counter starts over.  LAZY: SB3 calls env.seed() and then
        env.reset(), and a full reset settles every env and every donor lane for 50 control periods -- so seed() itself only re-keys, and the
        reset() that follows is the ONE full reset (the first visible episodes are episode 0 of the new key; seed(s) twice gives the same
        episodes twice).  Until that reset the donor lanes still hold episodes settled under the old key: the library refuses to step
        (tgr_step returns an error) rather than hand them out.  reset=True does the full reset right here."""
        if seed is not None:
            N.check(self.handle, self.lib.tgr_set_seed(self.handle, int(seed)), "tgr_set_seed")
            if reset:
                self.reset_tensors()
        return [seed] * self.num_envs

    def get_attr(self, attr_name, indices=None):
        idx = range(self.num_envs) if indices is None else ([indices] if isinstance(indices, int) else indices)
        return [getattr(self, attr_name) for _ in idx]

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        idx = range(self.num_envs) if indices is None else ([indices] if isinstance(indices, int) else indices)
        return [getattr(self, method_name)(*args, **kwargs) for _ in idx]

    def env_is_wrapped(self, wrapper_class, indices=None):
        idx = range(self.num_envs) if indices is None else ([indices] if isinstance(indices, int) else indices)
        return [False for _ in idx]

    def get_images(self):
        return [None] * self.num_envs

    def render(self, mode=None):
        return None

    # ------------------------------------------------------------------ state exchange (parity tests, checkpoints)
    def get_state(self):
        n = self.num_envs
        qpos = np.zeros((n, N.NQ)); qvel = np.zeros((n, N.NV)); ctrl = np.zeros((n, N.NU)); warm = np.zeros((n, N.NV))
        aux = np.zeros((n, N.AUX_K))
        rc = self.lib.tgr_get_state(self.handle, qpos.ctypes.data, qvel.ctypes.data, ctrl.ctypes.data, warm.ctypes.data, aux.ctypes.data)
        N.check(self.handle, rc, "tgr_get_state")
        return {"qpos": qpos, "qvel": qvel, "ctrl": ctrl, "qacc_warmstart": warm, "aux": aux}

    def set_state(self, qpos=None, qvel=None, ctrl=None, qacc_warmstart=None, aux=None):
        def p(a, cols):
            if a is None:
                return None, None
            a = np.ascontiguousarray(a, dtype=np.float64)
            if a.shape != (self.num_envs, cols):
                raise ValueError(f"expected shape {(self.num_envs, cols)}, got {a.shape}")
            return a, a.ctypes.data
        keep = [p(qpos, N.NQ), p(qvel, N.NV), p(ctrl, N.NU), p(qacc_warmstart, N.NV), p(aux, N.AUX_K)]
        rc = self.lib.tgr_set_state(self.handle, *[k[1] for k in keep])
        N.check(self.handle, rc, "tgr_set_state")

    def bind_substep_trace(self, on=True):
        """diagnostic: a uint32 [N, 24] device tensor that every step fills with the per-sub-step contact / solver trace of each env
        (include/tensegrity.h tgr_bind_substep_trace); None unbinds"""
        t = self.torch
        self.trace = t.zeros((self.num_envs, N.TRACE_WORDS), dtype=t.int32, device=self.device) if on else None
        N.check(self.handle, self.lib.tgr_bind_substep_trace(self.handle, self.trace.data_ptr() if on else None), "tgr_bind_substep_trace")
        return self.trace

    def time_kernels(self, n_slots, every=1):
        """HIP events around every `every`-th launch of the dominant kernel from now on (a ring of n_slots; 0 = off): tgr_time_kernels_every"""
        N.check(self.handle, self.lib.tgr_time_kernels_every(self.handle, int(n_slots), int(every)), "tgr_time_kernels_every")

    def kernel_times_ms(self, n):
        out = (C.c_float * int(n))()
        got = self.lib.tgr_kernel_times_ms(self.handle, out, int(n))
        if got < 0:
            N.check(self.handle, got, "tgr_kernel_times_ms")
        return [float(out[i]) for i in range(got)]

    def stats(self):
        out = (C.c_uint64 * 8)()
        N.check(self.handle, self.lib.tgr_stats(self.handle, out), "tgr_stats")
        sub = max(int(out[5]), 1)
        return {"dropped_contacts": int(out[0]), "settle_kernel_resets": int(out[1]), "donor_resets": int(out[2]),
                "newton_iters_per_substep": int(out[3]) / sub, "linesearch_evals_per_substep": int(out[4]) / sub, "unconverged_lane_substeps": int(out[6]),
                "lane_substeps": int(out[7])}

    def synchronize(self):
        self.lib.tgr_synchronize(self.handle)

    def to_sb3(self):
        from .sb3 import as_sb3_vecenv
        return as_sb3_vecenv(self)
