"""Observation / action space objects.

The reference uses ``gym.spaces`` (main/envs/zone_envs/ZoneEnvBase.py:96-112,
main/envs/wrappers.py:144-153).  When gym is installed these ARE gym's classes, so
``isinstance(space, gym.spaces.Dict)`` in main/src/utils/format.py:23 holds; otherwise
minimal look-alikes with the attributes the reference consumers touch (``.spaces``,
``.shape``, ``.low``, ``.high``, ``.sample()``, ``.contains()``).
"""
import numpy as np

try:  # pragma: no cover - gym is absent from the build image
    from gym.spaces import Box, Dict   # noqa: F401
    HAVE_GYM = True
except Exception:  # ImportError or a broken install
    HAVE_GYM = False

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.shape(low)
            self.shape = tuple(shape)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return np.random.uniform(lo, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Dict:
        def __init__(self, spaces):
            self.spaces = dict(spaces)

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

        def contains(self, x):
            return (isinstance(x, dict) and x.keys() == self.spaces.keys()
                    and all(self.spaces[k].contains(x[k]) for k in x))

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()

        def __repr__(self):
            return "Dict(" + ", ".join(f"{k}:{v}" for k, v in self.spaces.items()) + ")"
