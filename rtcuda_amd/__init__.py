"""rtcuda_amd -- MI355X-native render path behind lashhw/rtcuda's render() interface.

The product is ``librtcuda_amd.so`` (hand-written HIP kernels for gfx950 + a C-ABI, declared in
``include/rtcuda_amd.h``).  This Python package is only the glue that tests and ``bench.py`` use to
reach that C-ABI: ``rtcuda_amd.api`` (ctypes binding) and ``rtcuda_amd.scenes`` (the caller-side
scene recipes as flat numpy arrays).  There is no CPU fallback: if the library is missing or no
GPU is present, the calls raise.
"""
from . import scenes  # noqa: F401  (numpy only; safe without a GPU)

__all__ = ["scenes", "api"]
