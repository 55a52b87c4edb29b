"""Box space: the real gymnasium/gym class when importable, else a duck-typed stand-in with the same
attributes (shape, dtype, low, high, sample, contains) so SB3-style callers keep working."""
import numpy as np


def _real_box():
    for mod in ("gymnasium.spaces", "gym.spaces"):
        try:
            return __import__(mod, fromlist=["Box"]).Box
        except Exception:
            continue
    return None


class _Box:
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.low = np.full(self.shape, low, dtype=self.dtype) if np.isscalar(low) else np.asarray(low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype) if np.isscalar(high) else np.asarray(high, dtype=self.dtype)
        self._rng = np.random.default_rng(seed)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


Box = _real_box() or _Box
