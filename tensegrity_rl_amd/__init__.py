"""tensegrity_rl_amd -- MI355X-native batched tensegrity-robot RL environment.

Only what the hot path needs: csrc/ (HIP kernels + C-ABI), the ctypes binding, and the host-side
mirror of the reference's Gym Env / SB3 VecEnv surface."""
from . import build as _build  # noqa: F401
from ._native import load as load_native  # noqa: F401
from .envs import make, register_with_gym, tensegrity_env, tr_env  # noqa: F401
from .vec_env import TensegrityVecEnv, make_config  # noqa: F401

__all__ = ["TensegrityVecEnv", "make_config", "make", "register_with_gym", "tensegrity_env", "tr_env", "load_native"]
