"""doodle_amd — MI355X-native render hot path of DOODLE's heliostat field.

``HelioField`` (optics core) and ``HelioEnv`` (Gym-style env) keep the reference's
Python surface; the render arithmetic is hand-written HIP for gfx950 behind the C ABI
of ``include/helio.h`` (``doodle_amd/libhelio.so``).
"""
from .field import HelioField  # noqa: F401

__all__ = ["HelioField", "HelioEnv"]


def __getattr__(name):
    if name == "HelioEnv":
        from .env import HelioEnv
        return HelioEnv
    raise AttributeError(name)
