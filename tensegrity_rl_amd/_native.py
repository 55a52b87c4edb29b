"""ctypes binding of libtensegrity.so (include/tensegrity.h).  No CPU fallback: if the HIP
library is missing or no GPU is present, construction fails loudly."""
import ctypes as C
import os

from . import build as _build

i32, f64, u64 = C.c_int32, C.c_double, C.c_uint64

ENV_TENSEGRITY, ENV_TR = 0, 1
ACTIONS = {"straight": 0, "turn": 1, "aiming": 2, "tracking": 3, "vel_track": 4}
PRECISION = {"f32": 0, "f64": 1, "float32": 0, "float64": 1}
KERNEL_LAYOUT = {"auto": 0, "block": 1, "row": 2}
INFO_K, AUX_K, NQ, NV, NU = 26, 56, 21, 18, 6
AUX_RING, RING_CAPACITY = 23, 32
TRACE_WORDS = 48
INFO_TENDON_LENGTH, INFO_BAR_CONTACT = 16, 25
INFO_KEYS = ["reward_forward", "reward_ctrl", "reward_survive", "x_position", "y_position", "psi",
             "distance_from_origin", "x_velocity", "y_velocity", "forward_reward"]
AUX_KEYS = ["xy_lag_x", "xy_lag_y", "psi_lag", "reset_psi", "ep_return", "step_num", "elapsed", "episode", "ring_len",
            "waypt_x", "waypt_y", "ori_x", "ori_y", "vcmd_x", "vcmd_y", "vcmd_w", "src"] + [f"act_{i}" for i in range(6)] + \
           [f"ring_{i}" for i in range(32)] + ["cap_radius"]


class Config(C.Structure):
    _fields_ = [
        ("env_kind", i32), ("desired_action", i32), ("desired_direction", f64), ("terminate_when_unhealthy", i32),
        ("use_contact_forces", i32), ("use_cap_velocity", i32), ("is_test", i32), ("ctrl_cost_weight", f64),
        ("contact_cost_weight", f64), ("healthy_reward", f64), ("contact_force_lo", f64), ("contact_force_hi", f64),
        ("min_reset_heading", f64), ("max_reset_heading", f64), ("tendon_reset_mean", f64), ("tendon_reset_stdev", f64),
        ("tendon_max_length", f64), ("tendon_min_length", f64), ("reward_delay_steps", i32), ("frame_skip", i32),
        ("max_episode_steps", i32), ("timestep", f64), ("gravity", f64), ("cap_radius", f64), ("solref", f64 * 2),
        ("solimp", f64 * 5), ("friction", f64 * 3), ("impratio", f64), ("gain", f64), ("bias", f64 * 3),
        ("force_range", f64 * 2), ("ctrl_range", f64 * 2), ("ten_stiffness", f64 * 9), ("ten_damping", f64 * 9),
        ("ten_springlength", f64 * 9), ("actuator_deriv_skip_clamped", i32), ("precision", i32), ("auto_reset", i32),
        ("n_donors", i32), ("seed", u64), ("first_env_id", u64), ("solver_tolerance", f64), ("solver_iterations", i32),
        ("waypt_range", f64 * 2), ("waypt_angle_range", f64 * 2), ("ditch_reward_max", f64), ("ditch_reward_stdev", f64),
        ("waypt_reward_amplitude", f64), ("waypt_reward_stdev", f64), ("yaw_reward_weight", f64), ("solver_warm_start", i32),
        ("use_obs_noise", i32), ("obs_noise_tendon_stdev", f64), ("obs_noise_cap_pos_stdev", f64),
        ("model_kind", i32), ("site_offset", f64), ("act_tau", f64), ("hf_size", f64 * 4), ("hf_pos", f64 * 3),
        ("reset_noise_scale", f64), ("kernel_layout", i32), ("solver_rel_tolerance", f64), ("reset_stream_envs", i32), ("donor_stream", i32), ("use_cap_size_noise", i32), ("cap_size_noise_range", f64 * 2),
    ]


SYMBOLS = ["tgr_default_config", "tgr_create", "tgr_destroy", "tgr_num_envs", "tgr_obs_dim", "tgr_precision", "tgr_reset",
           "tgr_step", "tgr_step_f64", "tgr_get_state", "tgr_set_state", "tgr_stats", "tgr_set_seed", "tgr_synchronize",
           "tgr_last_error", "tgr_version", "tgr_sizeof_config", "tgr_lane_key", "tgr_uneven_config", "tgr_set_hfield", "tgr_bind_real_observation",
           "tgr_bind_substep_trace", "tgr_time_kernels", "tgr_time_kernels_every", "tgr_kernel_times_ms"]

_lib = None


def load():
    """dlopen libtensegrity.so (building it first if the sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("TGR_LIB") or _build.lib_path()
    if not os.environ.get("TGR_LIB") and _build.needs_build():
        path = _build.build()
    try:
        # torch ships its own libamdhip64; the library must bind to the HIP runtime torch uses (same device state, same
        # streams), so torch goes first.  Loaded the other way round the process ends up with two runtimes and
        # hipGetDeviceCount fails inside the library (seen on MI355X when build() and smoke() ran in one process).
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp = C.c_void_p
    L.tgr_default_config.argtypes = [C.POINTER(Config), C.c_int, C.c_int]
    L.tgr_create.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.POINTER(vp)]
    L.tgr_destroy.argtypes = [vp]
    L.tgr_num_envs.argtypes = [vp]
    L.tgr_obs_dim.argtypes = [vp]
    L.tgr_precision.argtypes = [vp]
    L.tgr_reset.argtypes = [vp, vp, vp, vp]
    L.tgr_step.argtypes = [vp] * 9
    L.tgr_step_f64.argtypes = [vp] * 9
    L.tgr_get_state.argtypes = [vp] * 6
    L.tgr_set_state.argtypes = [vp] * 6
    L.tgr_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.tgr_set_seed.argtypes = [vp, C.c_uint64]
    L.tgr_synchronize.argtypes = [vp]
    L.tgr_last_error.argtypes = [vp]
    L.tgr_last_error.restype = C.c_char_p
    L.tgr_version.restype = C.c_char_p
    L.tgr_sizeof_config.restype = C.c_int
    L.tgr_lane_key.argtypes = [vp, C.c_int]
    L.tgr_uneven_config.argtypes = [C.POINTER(Config)]
    L.tgr_set_hfield.argtypes = [vp, vp, C.c_int, C.c_int]
    L.tgr_lane_key.restype = C.c_uint64
    L.tgr_bind_real_observation.argtypes = [vp, vp]
    L.tgr_bind_substep_trace.argtypes = [vp, vp]
    L.tgr_time_kernels.argtypes = [vp, C.c_int]
    L.tgr_time_kernels_every.argtypes = [vp, C.c_int, C.c_int]
    L.tgr_kernel_times_ms.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    if L.tgr_sizeof_config() != C.sizeof(Config):
        raise RuntimeError(f"tgr_config layout mismatch: C {L.tgr_sizeof_config()} vs ctypes {C.sizeof(Config)}")
    _lib = L
    return L


def default_config(env_kind, desired_action):
    c = Config()
    rc = load().tgr_default_config(C.byref(c), env_kind, ACTIONS[desired_action] if isinstance(desired_action, str) else desired_action)
    if rc != 0:
        raise RuntimeError("tgr_default_config failed")
    return c


class NativeError(RuntimeError):
    pass


def check(handle, rc, what):
    if rc != 0:
        msg = load().tgr_last_error(handle)
        raise NativeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
