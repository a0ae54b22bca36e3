"""diffcloth_amd — MI355X-native (gfx950) differentiable cloth stepper with DiffCloth's step()/stepBackward() semantics.

The product is the C-ABI shared library `lib/libdiffcloth_hip.so` (include/diffcloth_hip.h) plus the C++ host
class mirroring the reference's `Simulation` and its `diffcloth_py` pybind11 surface (csrc/host/).
`diffcloth_amd.capi` is a ctypes view of the C-ABI used by the tests and by bench.py.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
