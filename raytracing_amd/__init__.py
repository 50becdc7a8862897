"""raytracing_amd — the per-pixel path-tracing sample loop of MartinPuda/raytracing on MI355X.

Host side: mirrors of the reference's scene constructors (`hittables`, `materials`, `texture`,
`scenes`, `camera`), the flattener that turns a world into the C-ABI scene description
(`flatten`), and the ctypes binding of librtmi.so (`lib`). Device side: `csrc/`.
"""
from . import abi, camera, flatten, hittables, materials, scenes, texture  # noqa: F401

__all__ = ["abi", "camera", "flatten", "hittables", "materials", "scenes", "texture"]
